/*
 * mpc_hip.h -- C ABI of libmpc_hip.so: the MI355X (gfx950) implementation of
 * the per-cache-line compression-size evaluator of scalable-arch/CAL_22-MPC.
 *
 * The reference has no FFI; its boundary for this path is the C++ pair
 *   comp::Compressor::CompressLine(std::vector<uint8_t>&)   src/compressor/Compressor.h:28
 *   comp::Compressor::GetResult()                           src/compressor/Compressor.h:29
 * fed by trace::Loader::GetCacheline()                      src/loader/Loader.h:77
 * from the loop compressLines()                             src/main.cpp:208-248.
 * This header is what a binding for that path would call instead: one opaque
 * evaluator per comp::VPC / comp::BDI object, a batch call that replaces the
 * per-line loop, and a flat integer statistics vector that replaces
 * VPCResult / BDIResult (src/compressor/VPC.h:36-238, BDI.h:23-88).
 *
 * Conventions: plain C types only; every function returns 0 or a negative
 * errno-style code (never throws, never exits); the caller owns all buffers;
 * one handle is used from one thread at a time.  There is NO CPU fallback:
 * without a usable HIP device the create calls fail with MPC_E_NODEVICE.
 */
#ifndef MPC_HIP_H
#define MPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_ABI_VERSION 1

/* error codes */
#define MPC_OK            0
#define MPC_E_INVAL      -22  /* bad argument / unsupported configuration value */
#define MPC_E_NOENT       -2  /* configuration file cannot be opened */
#define MPC_E_PARSE      -74  /* configuration is not valid JSON / schema */
#define MPC_E_NODEVICE   -19  /* no HIP device / HIP runtime failure at create */
#define MPC_E_NOMEM      -12
#define MPC_E_HIP         -5  /* HIP call failed during a batch; see mpc_last_error */

typedef struct mpc_handle mpc_handle;

/* Which kernel a VPC configuration maps to (mpc_info.kernel_path). */
#define MPC_PATH_VPC_FAST     1  /* (truncated) plane- or byte-major scan, windowed tables, any root for OneBase/Diff/Weight: vpc_lane_kernel */
#define MPC_PATH_VPC_GENERIC  2  /* any table / root / scan order              */
#define MPC_PATH_BDI          3
#define MPC_PATH_FPC          4
#define MPC_PATH_BPC          5

typedef struct {
  int32_t abi_version;
  int32_t algorithm;        /* 0 = VPC, 1 = BDI, 2 = FPC, 3 = BPC */
  int32_t line_size;        /* bytes per line (L) */
  int32_t num_modules;      /* VPC: M; BDI: 0 */
  int32_t num_clusters;     /* VPC: M+1 (cluster -1 .. M-1); BDI: 9 states; FPC: 8 prefixes; BPC: 7 patterns */
  int32_t hist_bins;        /* VPC: bins per cluster in the stats vector */
  int32_t kernel_path;      /* MPC_PATH_* */
  int32_t device;           /* HIP device ordinal the handle is bound to */
  uint64_t stats_len;       /* number of uint64 in the statistics vector */
} mpc_info;

/* ---- construction (replaces `new comp::VPC(configPath)`, VPC.h:244-249, and
 *      `new comp::BDI(lineSize)`, BDI.h:94-98) ------------------------------
 * device < 0 selects the current HIP device.  The VPC JSON schema is the one
 * VPC::parseConfig reads (VPC.cpp:72-330).                                  */
int mpc_create_vpc(const char *config_json_path, int device, mpc_handle **out);
int mpc_create_vpc_from_string(const char *config_json_text, int device, mpc_handle **out);
int mpc_create_bdi(unsigned line_size, int device, mpc_handle **out);   /* any multiple of 8 up to 256 (BDI.cpp:8) */
/* Limits of mpc_create_vpc*: up to 32 modules (31 prediction modules), lineSize a multiple of 4 up to
 * 256, clusters x histogram bins x 4 B <= 150 KiB (the per-workgroup histogram lives in LDS); MAE / MSE
 * reproduce the reference's running doubles bit for bit when lineSize is a power of two and to 1e-12
 * otherwise ("Deliberate deviations").
 * `new comp::FPC(lineSize)` (FPC.h:91-97): frequent pattern compression of the line's 32-bit
 * words (FPC.cpp:7-88).  Per-line output: size in bits; `selected` is written as 0.
 * One definition where the reference has undefined behaviour: a zero run ends at the end of
 * the line (FPC.cpp:26 reads past it).  Parity is unpinned (no fixture).     */
int mpc_create_fpc(unsigned line_size, int device, mpc_handle **out);
/* `new comp::BPC(lineSize)` (BPC.h:93-99): bit-plane compression (BPC.cpp:20-185).  Per-line
 * output: size in bits (may exceed 8*L: the reference does not cap it); `selected` is 0.
 * Kept as in the source: the first word always costs 3+4 bits (`if (base = 0)`,
 * BPC.cpp:98).  Defined where the source is undefined: words are zero-extended to 64 bits
 * (BPC.cpp:42-44 copies 4 bytes into an uninitialised int64_t).  Parity is unpinned.       */
int mpc_create_bpc(unsigned line_size, int device, mpc_handle **out);
void mpc_destroy(mpc_handle *h);

int mpc_get_info(const mpc_handle *h, mpc_info *info);
/* Why a VPC configuration runs on the generic kernel (kernel_path == MPC_PATH_VPC_GENERIC, some hundred times slower
 * than the fast kernel): one sentence naming the module and the property; "" for every other handle.  The
 * `compressor` CLI prints it to stderr so that the slow path is never entered silently.                          */
const char *mpc_path_reason(const mpc_handle *h);
/* The form of the kernel the handle launches, for logs and tests: "unrolled" (a built-in instantiation of the module
 * sequence), "unrolled, general layout" (RootIndex 1..15 / whole-plane truncation), "unrolled, compiled at creation"
 * (the sequence had no built-in instantiation: hiprtc compiled it when the handle was created; "(from the cache)"
 * when a previous process had), "run-time loop", "generic".  No counterpart in the reference.                          */
const char *mpc_kernel_form(const mpc_handle *h);
/* Build check, needs no device: when the configuration's module sequence has no built-in unrolled instantiation and would
 * be compiled when a handle is created (hiprtc), compile it now for gfx950 and return the size of the
 * code object; 0 when nothing would be compiled (built in, or a layout that takes the run-time loop); negative MPC_E_*
 * with the parser's or the compiler's message in `log`.                                                             */
long long mpc_jit_compile_check(const char *config_json_text, char *log, size_t log_cap);
/* Last error text of this handle (or of the last failed create if h==NULL). */
const char *mpc_last_error(const mpc_handle *h);

/* ---- the hot path (replaces the CompressLine loop, main.cpp:237-243) -----
 * `lines` is n consecutive lines of line_size bytes in HOST memory.  They are
 * staged through pinned double buffers with hipMemcpyAsync and evaluated on
 * the device.  size_bits_out (n x uint16: CompressLine's return value) and
 * selected_out (n x int8: VPC cluster -1..M-1, or BDIState 0..8) may each be
 * NULL.  Statistics accumulate in the handle exactly as m_Stat does.  The
 * call returns when the results are in the output buffers.  Calls of up to
 * 512 lines (the per-line CompressLine of an unchanged reference driver) are
 * evaluated in place from a small pinned buffer: one launch, one stream
 * synchronisation, no staging slots.                                        */
int mpc_compress_batch(mpc_handle *h, const uint8_t *lines, uint64_t n_lines,
                       uint16_t *size_bits_out, int8_t *selected_out);

/* Same, for lines already resident in DEVICE memory (16-byte aligned) and
 * optional DEVICE output arrays; asynchronous on `hip_stream` (a hipStream_t
 * passed straight to the launch; NULL = HIP's default stream).  Statistics
 * accumulate on the device; call mpc_sync (or mpc_stats_get), which wait for the
 * whole device, before reading outputs.                                     */
int mpc_compress_batch_device(mpc_handle *h, const void *d_lines, uint64_t n_lines,
                              uint16_t *d_size_bits_out, int8_t *d_selected_out,
                              void *hip_stream);
int mpc_sync(mpc_handle *h);

/* ---- statistics (replaces VPCResult / BDIResult) -------------------------
 * Integer vector, identical on 1 GPU, N GPUs after a sum all-reduce, and the
 * CPU oracle.  VPC layout (K = M+1 clusters, index k = cluster+1,
 * B = hist_bins):
 *   [0] lines  [1] original_bits  [2] compressed_bits
 *   [3 + 6k + 0] count_k            [3 + 6k + 1] original_bits_k
 *   [3 + 6k + 2] compressed_bits_k  [3 + 6k + 3] residue_lines_k
 *   [3 + 6k + 4] sum_r_k            [3 + 6k + 5] sum_r2_k
 *   [3 + 6K + k*B + s] histogram_k[s]   (lines of cluster k with size s bits)
 * MAE_k = sum_r_k / (L * residue_lines_k), MSE_k likewise (VPC.h:62-76).
 * BDI layout: [0] lines [1] original_bits [2] compressed_bits [3..11] Counts.
 * FPC layout: [0] lines [1] original_bits (32 per word) [2] compressed_bits [3..10] Counts of
 *             Prefix0..7 (FPC.h:13-23); TotalWords = their sum.
 * BPC layout: [0] lines [1] original_bits [2] compressed_bits [3] TotalWords [4..10] Counts in
 *             BPCPattern order (BPC.h:12-21).
 */
int mpc_stats_len(const mpc_handle *h, uint64_t *len);
int mpc_stats_get(mpc_handle *h, uint64_t *vec, size_t n);      /* syncs */
int mpc_stats_merge(mpc_handle *h, const uint64_t *vec, size_t n); /* += */
int mpc_stats_set(mpc_handle *h, const uint64_t *vec, size_t n);   /* = (after an all-reduce) */

/* Device-side exchange (multi-GPU without a host round trip).  The handle's device
 * accumulators ("raw" statistics: VPC [sum_r(K)] [sum_r2(K)] [histogram(K x B)], BDI
 * [Counts(9)] [compressed_bits], FPC [Counts(8)] [compressed_bits], BPC [Counts(7)] [TotalWords] [compressed_bits]) are plain
 * uint64 sums, so ranks may all-reduce them
 * directly: mpc_stats_copy_raw_device enqueues an asynchronous device-to-device copy of
 * the raw_len words into d_dst on hip_stream (after everything already enqueued there),
 * and mpc_stats_from_raw turns such an array -- on the host, e.g. after the all-reduce --
 * into the statistics vector described above (merged-in host statistics not included). */
int mpc_stats_raw_len(const mpc_handle *h, uint64_t *raw_len);
int mpc_stats_copy_raw_device(mpc_handle *h, void *d_dst, void *hip_stream);
int mpc_stats_from_raw(const mpc_handle *h, const uint64_t *raw, size_t raw_len, uint64_t *vec, size_t n);
int mpc_stats_reset(mpc_handle *h);

/* ---- configuration check without a device --------------------------------
 * Parses and validates a VPC configuration exactly as mpc_create_vpc does and
 * writes a JSON description (line size, modules, id bits, which kernel path the
 * configuration maps to and why, whether the module sequence has an unrolled
 * instantiation or runs in the fast kernel's run-time module loop) into out[cap].  Touches no HIP API.  Returns
 * 0, or the negative code mpc_create_vpc would return ({"error": ...}).     */
int mpc_config_describe(const char *config_json_text, char *out, size_t cap);

/* ---- file streaming (replaces trace::LoaderNPY, LoaderNPY.cpp:14-54) -----
 * Reads a C-order uint8 [N, L] .npy file in chunks straight into the pinned
 * staging buffers and evaluates rows [first_row, first_row+n_rows) clipped to
 * the file; skip_last_row != 0 reproduces the reference driver, which never
 * compresses the final row (LoaderNPY.cpp:28-32 + main.cpp:240).
 * rows_done receives the number of rows evaluated.                          */
int mpc_compress_npy(mpc_handle *h, const char *npy_path, uint64_t first_row,
                     uint64_t n_rows, int skip_last_row, uint64_t *rows_done);
/* Header probe: shape of a 2-D uint8 .npy file. */
int mpc_npy_shape(const char *npy_path, uint64_t *n_rows, uint64_t *line_size);

/* ---- GPGPU-Sim ".log" traces (replaces trace::gpgpusim::LoaderGPGPU,
 *      LoaderGPGPU.cpp:26-55, 93-119, plus the driver's filter, main.cpp:222-224)
 * File: 1 byte key count (17), 17 x (6-byte key, 1-byte size), then requests of
 * 62 header bytes + req_size data bytes.  Every complete request is read; only
 * GLOBAL_ACC_R (0) and GLOBAL_ACC_W (4) requests are evaluated; an incomplete
 * trailing request is ignored.  The line size is the req_size of the first
 * request (GetCachelineSize, LoaderGPGPU.cpp:16-24) and must equal the handle's;
 * an evaluated request of another size is an error (MPC_E_INVAL).
 * requests_read / lines_done may be NULL.                                    */
int mpc_compress_gpgpusim_log(mpc_handle *h, const char *log_path, uint64_t *requests_read,
                              uint64_t *lines_done);
/* Header probe: req_size of the first request (0 for a trace without requests). */
int mpc_gpgpusim_log_line_size(const char *log_path, uint32_t *line_size);

/* ---- measurement helpers (bench.py; not part of the evaluator) -----------
 * Synthetic device-resident traces of SURVEY.md 8d, generated on the GPU:
 * kind 0 zeros, 1 random u32, 2 fp32 sine, 3 mixed int/fp, 4 pointer qwords.
 * first_line offsets the global line index so shards are independent.       */
int mpc_synth_fill(void *d_lines, uint64_t n_lines, unsigned line_size, int kind,
                   uint64_t first_line, uint64_t seed, void *hip_stream);
/* Pure streaming read of `bytes` bytes (sum-reduce to one word): the measured
 * HBM read ceiling on the same buffer.                                      */
int mpc_read_bandwidth_probe(const void *d_buf, uint64_t bytes, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MPC_HIP_H */
