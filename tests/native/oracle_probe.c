/* oracle_probe.c -- runs the CPU oracle under AddressSanitizer / UBSan over pseudo-random
 * lines and configurations (CPU only; the GPU pool has no sanitizer support). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpc_oracle.h"

static unsigned long long s = 88172645463325252ull;
static unsigned rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }

int main(void)
{
  static mpc_o_config cfg;
  unsigned long long total = 0;
  for (int L = 32; L <= 128; L *= 2) {
    memset(&cfg, 0, sizeof(cfg));
    cfg.num_modules = 6;
    cfg.line_size = L;
    for (int k = 0; k <= 6; k++) cfg.enc_bits[k] = 3;
    cfg.modules[0].kind = MPC_O_ALLZERO;
    cfg.modules[1].kind = MPC_O_ALLWORDSAME;
    for (int m = 2; m < 6; m++) {
      mpc_o_module *mo = &cfg.modules[m];
      mo->kind = MPC_O_PREDCOMP;
      mo->pred_kind = m - 2;
      mo->root = (mo->pred_kind == MPC_O_PRED_CONSEC) ? 0 : (int)(rnd() % (unsigned)L);
      mo->consecutive_xor = rnd() & 1;
      mo->table_size = 8 * L - (int)(rnd() % 33);
      for (int j = 0; j < L; j++) {
        mo->base[j] = (int)(rnd() % (unsigned)L);
        mo->weight[j] = (float)(1u << (rnd() % 6)) / 8.0f;
        mo->diff[j] = (int)(rnd() % 600) - 300;
      }
      for (int j = 0; j < mo->table_size; j++) {
        mo->rows[j] = (int)(rnd() % 8);
        mo->cols[j] = (int)(rnd() % (unsigned)L);
      }
    }
    if (mpc_o_vpc_validate(&cfg) != 0) { printf("config rejected\n"); return 1; }
    mpc_o_vpc_stats st;
    memset(&st, 0, sizeof(st));
    st.hist_bins = 8 * L + 4;
    st.hist = calloc((size_t)(MPC_O_MAX_MODULES + 1) * st.hist_bins, sizeof(uint64_t));
    mpc_o_bdi_stats bst;
    memset(&bst, 0, sizeof(bst));
    mpc_o_fpc_stats fst;
    memset(&fst, 0, sizeof(fst));
    mpc_o_bpc_stats pst;
    memset(&pst, 0, sizeof(pst));
    uint8_t line[MPC_O_MAX_LINE];
    for (int i = 0; i < 3000; i++) {
      int mode = i % 5;
      for (int j = 0; j < L; j++)
        line[j] = mode == 0 ? (uint8_t)rnd() : mode == 1 ? 0 : mode == 2 ? (uint8_t)(j % 4 + 1)
                : mode == 3 ? (uint8_t)(rnd() % 3) : (uint8_t)(j * 3 + i);
      int sel;
      total += mpc_o_vpc_line(&cfg, line, &sel, &st);
      total += mpc_o_bdi_line(line, L, &sel, &bst);
      total += mpc_o_fpc_line(line, L, &fst);
      total += mpc_o_bpc_line(line, L, &pst);
    }
    if (fst.total_words != 3000ull * (unsigned)(L / 4) || pst.total_words != 3000ull * 33ull) { printf("word counts\n"); return 1; }
    free(st.hist);
  }
  printf("ok %llu\n", total);
  return 0;
}
