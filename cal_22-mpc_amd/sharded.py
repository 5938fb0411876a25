"""Multi-GPU layer of the evaluator: contiguous line sharding and the single
collective of the path -- a SUM all-reduce of the integer statistics vector
(RCCL over xGMI with backend "nccl" on the GPU node, gloo on CPU in tests).

Lines are independent (reference ``CompressLine`` keeps no cross-line state,
``src/compressor/VPC.cpp:22-70``, ``BDI.cpp:6-74``) and every statistic is a
commutative integer sum, so the all-reduced vector is identical to a
single-device run and to the CPU path; ratios / MAE / MSE are derived after it.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(n_lines: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [begin, end) of rank `rank`: GPU g gets [g*N/G, (g+1)*N/G)."""
    return (rank * n_lines) // world, ((rank + 1) * n_lines) // world


def all_reduce_stats(vec: np.ndarray, device=None) -> np.ndarray:
    """SUM all-reduce of a uint64 statistics vector over the default process group.
    One message of a few tens of KB: latency-bound, one call per trace."""
    import torch
    import torch.distributed as dist

    v = np.ascontiguousarray(vec, dtype=np.uint64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return v.copy()
    t = torch.from_numpy(v.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().view(np.uint64)


def all_reduce_raw_on_device(evaluator, scratch, stream: int = 0):
    """Device-side form of the same collective: copy the evaluator's raw uint64 accumulators into
    ``scratch`` (an int64 torch tensor of ``evaluator.stats_raw_len()`` elements on the evaluator's
    GPU) on ``stream`` and SUM all-reduce it in place -- no host round trip, nothing waits.
    ``evaluator.stats_from_raw(scratch.cpu().numpy().view(np.uint64))`` gives the statistics vector
    when the host finally wants it.  ``stream`` must be torch's current stream."""
    import torch.distributed as dist

    evaluator.stats_copy_raw_device(scratch.data_ptr(), stream)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(scratch, op=dist.ReduceOp.SUM)
    return scratch


def evaluate_sharded(evaluator, npy_path: str, rank: int, world: int, device=None, skip_last_row: bool = True):
    """Each rank streams its contiguous share of the rows of `npy_path` through
    `evaluator` (a ``VPC`` / ``BDI`` of this package bound to the rank's GPU), then
    the statistics are all-reduced and installed in every rank's evaluator."""
    import ctypes as C
    from . import lib

    rows, cols = C.c_uint64(), C.c_uint64()
    rc = lib().mpc_npy_shape(npy_path.encode(), C.byref(rows), C.byref(cols))
    if rc != 0:
        raise RuntimeError(f"cannot read {npy_path}")
    usable = max(int(rows.value) - 1, 0) if skip_last_row else int(rows.value)
    b, e = shard_range(usable, rank, world)
    evaluator.reset()
    # skip_last_row is already applied through `usable`
    done = evaluator.compress_npy(npy_path, first_row=b, n_rows=e - b, skip_last_row=skip_last_row)
    assert done == e - b
    total = all_reduce_stats(evaluator.stats_vector(), device)
    evaluator.stats_set(total)
    return total
