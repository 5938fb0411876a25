"""The N>1 path on CPU: world_size-2 (and 3) `gloo` process groups; each rank
evaluates its contiguous shard (here with the CPU oracle standing in for the
device) and the SUM all-reduce of the integer statistics vector must equal the
single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, L, n_lines, out_dir):
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharded = importlib.import_module("cal_22-mpc_amd.sharded")
    configs = importlib.import_module("cal_22-mpc_amd.configs")
    traces = importlib.import_module("cal_22-mpc_amd.traces")
    from oracle import oracle as O
    b, e = sharded.shard_range(n_lines, rank, world)
    # every rank generates only its own shard of the global trace (counter-based generators)
    lines = traces.mixed(e - b, L, first_line=b)
    o = O.VpcOracle(configs.probe_config(L))
    o.compress(lines)
    total = sharded.all_reduce_stats(o.stats_vector())
    bo = O.BdiOracle(L)
    bo.compress(lines)
    btotal = sharded.all_reduce_stats(bo.stats_vector())
    np.save(os.path.join(out_dir, f"vpc_{rank}.npy"), total)
    np.save(os.path.join(out_dir, f"bdi_{rank}.npy"), btotal)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_stats_equal_single_process(world, tmp_path, oracle, configs, traces):
    import torch.multiprocessing as mp
    L, n = 64, 10007     # not divisible by the world size: ragged shards
    port = _free_port()
    mp.spawn(_worker, args=(world, port, L, n, str(tmp_path)), nprocs=world, join=True)
    whole = oracle.VpcOracle(configs.probe_config(L))
    whole.compress(traces.mixed(n, L))
    bw = oracle.BdiOracle(L)
    bw.compress(traces.mixed(n, L))
    for r in range(world):
        assert (np.load(tmp_path / f"vpc_{r}.npy") == whole.stats_vector()).all()
        assert (np.load(tmp_path / f"bdi_{r}.npy") == bw.stats_vector()).all()


def test_shard_ranges_cover_everything():
    sharded = pkg("sharded")
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [sharded.shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in cuts) - min(e - b for b, e in cuts) <= 1
    # without a process group the reduce is the identity
    v = np.arange(10, dtype=np.uint64)
    assert (sharded.all_reduce_stats(v) == v).all()
