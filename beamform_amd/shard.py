"""Frame-range sharding of one long stream across ranks (one process per GPU) and the final gather.

The path shards naturally: frames are independent except for
  * the overlap-add neighbour (1 frame) and
  * mvdr/lcmv's covariance of the previous P frames,
so rank r recomputes `warm` frames in front of its range locally and drops their output;
no data-path collective is needed.  The only collective is the final gather of the output hops
(RCCL over xGMI on GPUs; gloo in the CPU tests).  gss, phasempf and the mcra node recurse over frames and shard
by stream only (halo = None).

What a rank must FEED for its owned hops to equal the single-stream result bit for bit
(util.h:272-277,292-308: frame t = [hop t-1 | hop t], output hop t = tail(frame t-1) + head(frame t), a cold node's
ring buffer holds one hop of zeros):

    hops  lo-warm-lead .. lo-warm-1 | lo-warm .. lo-1 | lo .. hi-1
          `lead` (1 hop, or 0 at    | `warm` frames   | owned
          the stream start): only   | recomputed,     |
          seeds the ring buffer so  | output dropped  |
          that frame lo-warm has    |                 |
          its true first half       |                 |

Without the lead hop frame lo-warm would start from the cold node's zero hop and (das/phase) the tail added into
hop lo, or (mvdr/lcmv) the oldest covariance column of frame lo-1, would be wrong.  `Shard.first_feed_frame`,
`n_feed` and `n_drop` include it; `run_shard` below drives the HIP path with exactly that.
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
    lo: int        # first frame (= output hop) this rank owns
    hi: int        # one past the last
    warm: int      # frames recomputed in front of lo (clipped at the stream start)
    lead: int = 0  # hops fed in front of the first recomputed frame to seed the ring buffer (0 at the stream start)

    @property
    def first_input_frame(self) -> int:
        """First frame that is recomputed correctly (its first half comes from the lead hop)."""
        return self.lo - self.warm

    @property
    def first_feed_frame(self) -> int:
        """First hop of the global stream a cold node on this rank is fed."""
        return self.lo - self.warm - self.lead

    @property
    def n_feed(self) -> int:
        """Hops fed to the cold node: lead + warm + owned."""
        return self.hi - self.first_feed_frame

    @property
    def n_drop(self) -> int:
        """Output hops in front of the owned range that are discarded."""
        return self.warm + self.lead

    @property
    def n_own(self) -> int:
        return self.hi - self.lo

    @property
    def n_process(self) -> int:  # round-1 name: recomputed + owned frames, WITHOUT the lead hop
        return self.hi - self.lo + self.warm


def plan(n_frames: int, world: int, rank: int, halo: int) -> Shard:
    """Contiguous, near-equal frame ranges; rank 0 starts from the true stream state (no warm-up, no lead)."""
    if halo is None:
        raise ValueError("this node recurses over frames: shard by stream, not by frame range")
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    warm = min(halo, lo)
    lead = 1 if (halo > 0 and lo - warm > 0) else 0
    return Shard(lo, hi, warm, lead)


def run_shard(bf, x_feed_ptr: int, sh: Shard, y_feed_ptr: int, stream: int = 0, reset: bool = True) -> int:
    """Drive the HIP path (beamform_amd.capi.Beamformer `bf`, one input stream) over this rank's part of the stream.

    x_feed_ptr: device pointer of the rank's slice of the global stream, hops [first_feed_frame, hi)
                (planar [M][n_feed*hop] or interleaved [n_feed*hop][M], per the handle's layout).
    y_feed_ptr: device buffer of n_feed*hop floats; the owned hops start at element n_drop*hop.
    The handle is put back to the reference's cold start first (the rank's node knows nothing about the frames
    before its slice): bf_reset_async clears the carried state ON `stream`, so the step orders against whatever the
    caller has in flight there; nothing synchronises with the host.
    Returns the element offset of the first owned output sample in y_feed.
    """
    if bf.n_streams != 1 or bf.n_dirs != 1:
        raise ValueError("frame-range sharding drives one input stream and one look direction per handle")
    if reset:
        bf.reset_async(stream)
    if sh.n_feed > 0:
        bf.process_device(x_feed_ptr, sh.n_feed, y_feed_ptr, 0, stream)
    return sh.n_drop * bf.H


def pieces(sh: Shard, n_pieces: int):
    """Cut a rank's fed hops into `n_pieces` consecutive runs: [(f0, f1, own0, n_own)] with f0:f1 the fed-hop range of the piece
    and own0, n_own the owned hops it completes (indices into the rank's owned range).  Every rank and the receiver use this."""
    per = -(-sh.n_feed // max(1, n_pieces))
    out = []
    for c in range(n_pieces):
        f0, f1 = c * per, min((c + 1) * per, sh.n_feed)
        if f1 <= f0:
            break
        o0 = max(f0, sh.n_drop)
        out.append((f0, f1, o0 - sh.n_drop, max(0, f1 - o0)))
    return out


class _StagedRecv:
    """irecv into a host buffer + the copy into its device slice once it has arrived (host_staged transfers)."""

    def __init__(self, work, host, dev_slice):
        self.work, self.host, self.dev_slice = work, host, dev_slice

    def wait(self):
        self.work.wait()
        self.dev_slice.copy_(self.host)


class _StagedSend:
    def __init__(self, work, host):
        self.work, self.host = work, host   # the host buffer lives as long as the transfer

    def wait(self):
        self.work.wait()


def run_shard_overlapped(bf, x_feed, y_feed, n_frames: int, world: int, rank: int, halo: int, n_pieces: int = 4, dst: int = 0,
                         out=None, stream: int = 0, host_staged: bool = False, self_loop: bool = False):
    """run_shard with the final gather overlapped: the rank's slice is walked in `n_pieces` pieces (state carries from piece
    to piece), and as soon as a piece is enqueued its owned hops are handed to an ASYNCHRONOUS point-to-point transfer to
    `dst`, which the backend orders behind the compute stream's work so far and runs on its own stream -- piece c travels
    while piece c+1 is computed (RCCL: the direct xGMI link of each peer; grouped send / recv is what ncclGather does underneath).

    x_feed [M, n_feed*hop] / y_feed [n_feed*hop]: torch tensors (planar) holding / receiving hops [first_feed_frame, hi).
    out (dst only): [n_frames*hop] tensor, filled in stream order.  Every rank cuts its slice by the same rule (`pieces`), so
    the receiver knows each sender's piece sizes without a handshake.
    host_staged: carry the pieces through host buffers (a backend without device-tensor point-to-point, e.g. gloo: the one-GPU
    test hook); the schedule -- who sends what in which round -- is the same.
    self_loop: `dst` moves its OWN pieces through the backend as well (one grouped isend + irecv to itself per piece instead of the
    local copy): on a one-rank group this is the whole point-to-point path of the N > 1 run executed on one GPU (bench.py
    BF_BENCH_FORCE_DIST=1).
    Returns the pending work handles: wait on them (or synchronise the device) before reading `out`."""
    import contextlib
    import torch
    import torch.distributed as dist
    if bf.n_streams != 1 or bf.n_dirs != 1:  # a piece writes [piece_frames * hop] at y_feed[f0 * hop:]: one output row only
        raise ValueError("frame-range sharding drives one input stream and one look direction per handle")
    H = bf.H
    sh = plan(n_frames, world, rank, halo)
    # the transfers and the copies below order against torch's CURRENT stream, the kernels against `stream`: make them the same
    ctx = contextlib.nullcontext()
    if x_feed.is_cuda and stream != torch.cuda.current_stream(x_feed.device).cuda_stream:
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=x_feed.device))
    with ctx:
        return _run_shard_overlapped(bf, x_feed, y_feed, n_frames, world, rank, halo, n_pieces, dst, out, stream, sh, H, dist, host_staged, self_loop)


def _run_shard_overlapped(bf, x_feed, y_feed, n_frames, world, rank, halo, n_pieces, dst, out, stream, sh, H, dist, host_staged, self_loop=False):
    import torch
    bf.reset_async(stream)
    peers = {r: pieces(plan(n_frames, world, r, halo), n_pieces) for r in range(world)} if rank == dst else None
    works = []
    mine_p = pieces(sh, n_pieces)
    rounds = max(len(ps) for ps in peers.values()) if rank == dst else len(mine_p)  # a peer may cut one piece more than dst
    for c in range(rounds):
        n_own = 0
        if c < len(mine_p):
            f0, f1, own0, n_own = mine_p[c]
            bf.process_device_strided(x_feed[:, f0 * H:].data_ptr(), f1 - f0, y_feed[f0 * H:].data_ptr(), sh.n_feed * H, stream)
            mine = y_feed[(sh.n_drop + own0) * H:(sh.n_drop + own0 + n_own) * H]
        if rank == dst:
            if n_own > 0 and self_loop and not host_staged:
                # to itself through the backend: send and receive in ONE group (an ungrouped send to oneself never meets its receive)
                sl = out[(sh.lo + own0) * H:(sh.lo + own0 + n_own) * H]
                works += dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, dst), dist.P2POp(dist.irecv, sl, dst)])
            elif n_own > 0:
                out[(sh.lo + own0) * H:(sh.lo + own0 + n_own) * H].copy_(mine, non_blocking=True)
            for r in range(world):
                if r == rank or c >= len(peers[r]):
                    continue
                _, _, o0, n = peers[r][c]
                if n > 0:
                    lo_r = plan(n_frames, world, r, halo).lo
                    sl = out[(lo_r + o0) * H:(lo_r + o0 + n) * H]
                    if host_staged:
                        hb = torch.empty(n * H, dtype=out.dtype)
                        works.append(_StagedRecv(dist.irecv(hb, src=r), hb, sl))
                    else:
                        works.append(dist.irecv(sl, src=r))
        elif n_own > 0:
            if host_staged:
                hb = mine.cpu()   # synchronises with the piece just enqueued
                works.append(_StagedSend(dist.isend(hb, dst=dst), hb))
            else:
                works.append(dist.isend(mine, dst=dst))
    return works


def run_sharded(bf, x_feed, n_frames: int, world: int, rank: int, halo: int, dst: int = 0, stream: int = 0,
                gather: bool = True):
    """One rank's whole job: plan -> cold node -> feed lead + warm + owned hops -> drop -> final gather.

    x_feed: torch tensor on the rank's GPU holding hops [first_feed_frame, hi) of the global stream
            (planar [M, n_feed*hop]).  Returns the full output [n_frames*hop] on `dst` (None elsewhere), or the
            owned slab when gather=False.
    """
    import torch
    sh = plan(n_frames, world, rank, halo)
    assert x_feed.is_contiguous() and x_feed.numel() == bf.M * sh.n_feed * bf.H, "x_feed must hold exactly the fed hops"
    y_feed = torch.empty(sh.n_feed * bf.H, dtype=torch.float32, device=x_feed.device)
    off = run_shard(bf, x_feed.data_ptr(), sh, y_feed.data_ptr(), stream)
    y_own = y_feed[off:]
    if not gather:
        return y_own
    return gather_hops(y_own, n_frames, world, rank, bf.H, dst)


def gather_hops(y_local, n_frames: int, world: int, rank: int, hop: int = 512, dst: int = 0, out=None):
    """Collect per-rank output slabs [frames_r*hop] on `dst` in frame order; one collective.

    Works on any torch.distributed backend (nccl = RCCL on GPUs, gloo on CPU).  `out` (dst only, optional):
    preallocated list of `world` receive buffers of max-slab size, reused across calls."""
    import torch
    import torch.distributed as dist
    sizes = [plan(n_frames, world, r, 0).n_own for r in range(world)]
    pad = max(sizes) * hop
    if y_local.numel() == pad:
        buf = y_local.contiguous()
    else:
        buf = torch.zeros(pad, dtype=y_local.dtype, device=y_local.device)
        buf[: y_local.numel()] = y_local
    if rank == dst and out is None:
        out = [torch.empty_like(buf) for _ in range(world)]
    dist.gather(buf, out if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    if all(n == sizes[0] for n in sizes):
        return torch.cat(out)
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
