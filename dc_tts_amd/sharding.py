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


def synthesize_sharded(L: np.ndarray, synth: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray, np.ndarray]],
                       group: Optional[dist.ProcessGroup] = None):
    """Every rank passes the SAME full character batch L (B, max_N) int32 (host); `synth` maps a host slice of it to
    host arrays (Y, Z, max_attentions) -- on a GPU rank: upload, Engine.synthesize, download.  Rank 0 returns the
    full-batch (Y, Z, max_attentions) in the original utterance order; other ranks return None.
    The only communication is this final host gather (gather_object: slices may be ragged)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(L.shape[0], world, rank)
    out = synth(L[lo:hi]) if hi > lo else None
    if world == 1:
        return out
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((lo, hi, out), gathered, dst=0, group=group)
    if rank != 0:
        return None
    parts = sorted((g for g in gathered if g[2] is not None), key=lambda g: g[0])
    assert parts and parts[0][0] == 0 and all(a[1] == b[0] for a, b in zip(parts, parts[1:])) and parts[-1][1] == L.shape[0]
    return tuple(np.concatenate([p[2][k] for p in parts], axis=0) for k in range(3))


def gpu_synth(engine) -> Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray, np.ndarray]]:
    """Adapter: host slice -> this rank's GPU -> host."""
    def run(Ls: np.ndarray):
        Ld = torch.from_numpy(np.ascontiguousarray(Ls, dtype=np.int32)).to(engine.device)
        Y, Z, mx = engine.synthesize(Ld)
        return Y.cpu().numpy(), Z.cpu().numpy(), mx.cpu().numpy()
    return run
