"""Frame-range sharding of one long stream across ranks (one process per GPU) and the final gather.

The path shards naturally: frames are independent except for
  * the overlap-add neighbour (1 frame) and
  * mvdr/lcmv's covariance of the previous P frames,
so rank r recomputes `halo` warm-up frames in front of its range locally and drops their output;
no data-path collective is needed.  The only collective is the final gather of the output hops
(RCCL over xGMI on GPUs; gloo in the CPU tests).  gss, phasempf and the mcra node recurse over frames and shard
by stream only (halo = None).
"""
from __future__ import annotations

from dataclasses import dataclass


def halo_frames(params: dict):
    """Warm-up frames a shard must recompute in front of its first frame (None = not frame-shardable)."""
    algo = params["algo"]
    if algo in ("das", "phase"):
        return 1                                  # overlap-add neighbour only
    if algo in ("mvdr", "lcmv"):
        return int(params["past_windows"]) + 1    # covariance history of frame lo-1, plus that frame
    return None                                   # gss / phasempf / mcra / gsc: recursion over frames (gsc: over samples)


@dataclass
class Shard:
    lo: int      # first frame whose output this rank owns
    hi: int      # one past the last
    warm: int    # frames recomputed in front of lo (clipped at the stream start)

    @property
    def first_input_frame(self) -> int:
        return self.lo - self.warm

    @property
    def n_process(self) -> int:
        return self.hi - self.lo + self.warm


def plan(n_frames: int, world: int, rank: int, halo: int) -> Shard:
    """Contiguous, near-equal frame ranges; rank 0 starts from the true stream state (no warm-up)."""
    if halo is None:
        raise ValueError("this node recurses over frames: shard by stream, not by frame range")
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return Shard(lo, hi, min(halo, lo))


def gather_hops(y_local, n_frames: int, world: int, rank: int, hop: int = 512, dst: int = 0):
    """Collect per-rank output slabs [frames_r*hop] on `dst` in frame order; one collective.

    Works on any torch.distributed backend (nccl = RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    sizes = [plan(n_frames, world, r, 0).hi - plan(n_frames, world, r, 0).lo for r in range(world)]
    pad = max(sizes) * hop
    buf = torch.zeros(pad, dtype=y_local.dtype, device=y_local.device)
    buf[: y_local.numel()] = y_local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[: n * hop] for o, n in zip(out, sizes)])


def plan_dirs(n_dirs: int, world: int, rank: int):
    """Look-direction sharding: every rank sees the whole input and evaluates a contiguous,
    near-equal slice [lo, hi) of the direction list; no halo, no exchange until the final gather."""
    base, rem = divmod(n_dirs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_dirs(y_local, n_dirs: int, world: int, rank: int, dst: int = 0):
    """Collect per-rank direction slabs [dirs_r, samples] on `dst` in direction order; one collective."""
    import torch
    import torch.distributed as dist
    sizes = [plan_dirs(n_dirs, world, r)[1] - plan_dirs(n_dirs, world, r)[0] for r in range(world)]
    n = y_local.shape[-1]
    buf = torch.zeros((max(sizes), n), dtype=y_local.dtype, device=y_local.device)
    buf[: y_local.shape[0]] = y_local
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:k] for o, k in zip(out, sizes)])
