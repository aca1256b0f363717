"""Multi-GPU = embarrassing batch sharding: utterances are independent everywhere on the synthesis path
(layer-norm is per position, attention per utterance), so a batch is cut into contiguous slices, one per rank / GPU,
weights are replicated, and NO collective touches the data path.  Results are gathered on the host of rank 0.

One process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm for the GPU processes, "gloo" in the CPU tests).
"""
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(B: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a batch of B utterances owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _gather_rows(x: torch.Tensor, counts, group, rank: int, dst: int = 0):
    """Host gather of per-rank row blocks x (n_r, ...) (CPU tensors, ideally pinned) to rank `dst` with tensor collectives only
    (no pickling): blocks are padded to the largest count, gathered, and cut back.  Returns the concatenation on dst, else None."""
    world = len(counts)
    nmax = max(counts)
    pad = x
    if x.shape[0] != nmax:
        pad = torch.zeros((nmax,) + tuple(x.shape[1:]), dtype=x.dtype)
        pad[: x.shape[0]] = x
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:n] for b, n in zip(bufs, counts)], dim=0)


def _all_ranks_ok(err: Optional[BaseException], group, world: int):
    """A rank whose decode failed must not leave the others blocked in the gathers: every rank contributes a flag, and all of them raise."""
    if world > 1:
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag[0]) and err is None:
            raise RuntimeError("synthesis failed on another rank (its decode reported a device-side failure); no results were gathered")
    if err is not None:
        raise err


def gather_to_host(Z: torch.Tensor, Zh: torch.Tensor, group, world: int, rank: int, engine=None):
    """SURVEY 8e's result gather for equal shards: this rank's device tensor Z -> its pinned host buffer Zh (one D2H copy), then
    rank 0's host over `group` (gloo).  Returns the (world * B, ...) host tensor on rank 0, Zh itself when world == 1, else None.
    With `engine`, the decode status of this rank is checked first and a failure on ANY rank raises on every rank before the gather."""
    Zh.copy_(Z, non_blocking=True)
    err = None
    try:
        if engine is not None:
            engine.synchronize()
        else:
            torch.cuda.synchronize()
    except Exception as e:                                             # noqa: BLE001 -- re-raised on every rank below
        err = e
    _all_ranks_ok(err, group, world)
    if world == 1:
        return Zh
    return _gather_rows(Zh, [Z.shape[0]] * world, group, rank)


def synthesize_sharded(L: np.ndarray, synth: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray, np.ndarray]],
                       group: Optional[dist.ProcessGroup] = None):
    """Every rank passes the SAME full character batch L (B, max_N) int32 (host); `synth` maps a host slice of it to
    host arrays (Y, Z, max_attentions) -- on a GPU rank: upload, Engine.synthesize, download into pinned memory.  Rank 0
    returns the full-batch (Y, Z, max_attentions) in the original utterance order; other ranks return None.
    The only communication is this final host gather: three tensor gathers over `group` (gloo on the GPU ranks too: results
    meet on the host), shards may be ragged (sizes differ by at most one; an empty shard sends nothing but padding)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    Bt = L.shape[0]
    bounds = [shard_bounds(Bt, world, r) for r in range(world)]
    lo, hi = bounds[rank]
    out, err = None, None
    try:
        out = synth(L[lo:hi]) if hi > lo else None
    except Exception as e:                                             # noqa: BLE001 -- re-raised on every rank by _all_ranks_ok
        err = e
    _all_ranks_ok(err, group, world)
    if world == 1:
        return out
    counts = [b[1] - b[0] for b in bounds]
    # shapes of one utterance's results are known from any non-empty shard; rank 0 always owns one when Bt >= 1
    shapes = [None, None, None]
    if out is not None:
        shapes = [(tuple(o.shape[1:]), str(o.dtype)) for o in out]
    meta = [shapes]
    dist.broadcast_object_list(meta, src=0, group=group)               # a few dozen bytes of shape metadata, not the data
    shapes = meta[0]
    res = []
    for k in range(3):
        shp, dt = shapes[k]
        x = torch.from_numpy(np.ascontiguousarray(out[k])) if out is not None else torch.zeros((0,) + shp, dtype=getattr(torch, dt))
        g = _gather_rows(x, counts, group, rank)
        res.append(None if g is None else g.numpy())
    return tuple(res) if rank == 0 else None


def gpu_synth(engine) -> Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """Adapter: host slice -> this rank's GPU -> host."""
    def run(Ls: np.ndarray):
        Ld = torch.from_numpy(np.ascontiguousarray(Ls, dtype=np.int32)).to(engine.device)
        Y, Z, mx = engine.synthesize(Ld, check=True)                   # waits; a decode that failed on the device is repeated once, one launch per layer
        outs = []
        for t in (Y, Z, mx):                                           # one D2H copy each, into pinned host memory
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            outs.append(h)
        engine.synchronize()                                           # waits for the copies AND raises if the decode's bounded in-kernel wait gave up
        return tuple(h.numpy() for h in outs)
    return run
