"""world_size-2 gloo test of the N > 1 path: frame-range sharding with halo + lead hop + the final gather.
The compute stand-in on CPU is the oracle (test infrastructure); on GPUs shard.run_shard / bench.py drive the same
plan through the HIP path (tests/test_shard_gpu.py) and RCCL."""
import os
import socket

import numpy as np
import pytest

from beamform_amd import shard
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene


def test_plan_partitions_exactly():
    for F, W in [(10, 2), (11, 3), (65536, 8), (7, 8)]:
        cover = []
        for r in range(W):
            s = shard.plan(F, W, r, 11)
            assert s.warm == min(11, s.lo) and s.first_input_frame >= 0
            assert s.lead == (1 if s.first_input_frame > 0 else 0) and s.first_feed_frame >= 0
            assert s.n_feed == s.lead + s.warm + s.n_own and s.n_drop == s.lead + s.warm
            cover += list(range(s.lo, s.hi))
        assert cover == list(range(F))
    assert shard.halo_frames(make_params("das")) == 1
    assert shard.halo_frames(make_params("lcmv", interf=(10.0,))) == 11
    assert shard.halo_frames(make_params("gss")) is None
    with pytest.raises(ValueError):
        shard.plan(10, 2, 0, None)


def _worker(rank, world, port, algo, M, F, ret):
    import torch
    import torch.distributed as dist
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = make_params(algo, n_mics=M, theta=20.0)
    x = make_scene(M, F, seed=123)
    sh = shard.plan(F, world, rank, shard.halo_frames(p))
    # a cold node fed exactly what the plan says: lead hop (seeds the ring buffer) + warm frames + owned frames
    node = oracle.OracleNode(p)
    seg = np.ascontiguousarray(x[:, sh.first_feed_frame * 512: sh.hi * 512])
    assert seg.shape[1] == sh.n_feed * 512
    y, _ = node.process(seg)
    y_own = torch.from_numpy(y[sh.n_drop * 512:].copy())
    full = shard.gather_hops(y_own, F, world, rank)
    if rank == 0:
        ret.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,M,F", [("das", 4, 21), ("mvdr", 4, 40)])
def test_two_rank_sharding_reproduces_single_stream(algo, M, F):
    import torch.multiprocessing as mp
    import oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, algo, M, F, ret)) for r in range(2)]
    for pr in procs:
        pr.start()
    full = ret.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    y_ref, _ = oracle.OracleNode(make_params(algo, n_mics=M, theta=20.0)).process(make_scene(M, F, seed=123))
    lo1 = shard.plan(F, 2, 1, 0).lo
    assert np.array_equal(full[: lo1 * 512], y_ref[: lo1 * 512], equal_nan=True)
    # rank 1's first owned hop already has exact history (halo), so the rest is bit-identical too
    assert np.array_equal(full[lo1 * 512:], y_ref[lo1 * 512:], equal_nan=True)


def test_plan_dirs_partitions_exactly():
    for D, W in [(5, 2), (16, 8), (3, 4)]:
        cover = []
        for r in range(W):
            lo, hi = shard.plan_dirs(D, W, r)
            cover += list(range(lo, hi))
        assert cover == list(range(D))


def _dir_worker(rank, world, port, thetas, M, F, ret):
    import torch
    import torch.distributed as dist
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = make_scene(M, F, seed=321)
    lo, hi = shard.plan_dirs(len(thetas), world, rank)
    ys = [oracle.OracleNode(make_params("das", n_mics=M, theta=t)).process(x)[0] for t in thetas[lo:hi]]
    y_own = torch.from_numpy(np.stack(ys)) if ys else torch.zeros((0, F * 512))
    full = shard.gather_dirs(y_own, len(thetas), world, rank)
    if rank == 0:
        ret.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_direction_sharding():
    """the theta list shards across ranks, every rank reads the whole input, one gather."""
    import torch.multiprocessing as mp
    import oracle
    thetas, M, F = [-90.0, -20.0, 20.0, 75.0, 160.0], 4, 9
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_dir_worker, args=(r, 2, port, thetas, M, F, ret)) for r in range(2)]
    for pr in procs:
        pr.start()
    full = ret.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    x = make_scene(M, F, seed=321)
    for d, t in enumerate(thetas):
        assert np.array_equal(full[d], oracle.OracleNode(make_params("das", n_mics=M, theta=t)).process(x)[0])


class _OracleBf:
    """CPU stand-in with the slice of capi.Beamformer's interface that shard.run_shard_overlapped drives (raw pointers in,
    state carried from piece to piece): the oracle node is the compute (test infrastructure)."""

    def __init__(self, p):
        import oracle
        self._p, self._oracle = p, oracle
        self.H, self.M, self.n_streams, self.n_dirs = p["hop"], p["n_mics"], 1, 1
        self.node = oracle.OracleNode(p)

    def reset_async(self, stream=0):
        self.node = self._oracle.OracleNode(self._p)

    def process_device_strided(self, x_ptr, n_frames, y_ptr, mic_stride, stream=0):
        import ctypes
        n = n_frames * self.H
        x = np.stack([np.ctypeslib.as_array(ctypes.cast(x_ptr + 4 * m * mic_stride, ctypes.POINTER(ctypes.c_float)), (n,))
                      for m in range(self.M)])
        y, _ = self.node.process(np.ascontiguousarray(x))
        np.ctypeslib.as_array(ctypes.cast(y_ptr, ctypes.POINTER(ctypes.c_float)), (n,))[:] = y


def _overlap_worker(rank, world, port, algo, M, F, n_pieces, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = make_params(algo, n_mics=M, theta=20.0)
    halo = shard.halo_frames(p)
    sh = shard.plan(F, world, rank, halo)
    x = torch.from_numpy(np.ascontiguousarray(make_scene(M, F, seed=123)[:, sh.first_feed_frame * 512: sh.hi * 512]))
    y = torch.zeros(sh.n_feed * 512)
    out = torch.full((F * 512,), float("nan")) if rank == 0 else None
    works = shard.run_shard_overlapped(_OracleBf(p), x, y, F, world, rank, halo, n_pieces=n_pieces, out=out)
    for w in works:
        w.wait()
    if rank == 0:
        ret.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,M,F,world,n_pieces", [("das", 4, 23, 2, 3), ("mvdr", 4, 41, 3, 4), ("das", 3, 9, 2, 8)])
def test_overlapped_gather_reproduces_single_stream(algo, M, F, world, n_pieces):
    """The chunked walk + per-piece point-to-point transfers (shard.run_shard_overlapped) assemble the single-stream output,
    including pieces that lie entirely inside a rank's warm-up frames and more pieces than frames."""
    import torch.multiprocessing as mp
    import oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, algo, M, F, n_pieces, ret)) for r in range(world)]
    for pr in procs:
        pr.start()
    full = ret.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    y_ref, _ = oracle.OracleNode(make_params(algo, n_mics=M, theta=20.0)).process(make_scene(M, F, seed=123))
    assert np.array_equal(full, y_ref, equal_nan=True)


def test_pieces_cover_the_owned_range():
    for F, W, halo, K in [(100, 3, 11, 4), (9, 2, 1, 8), (65536, 8, 1, 4), (40, 4, 11, 1)]:
        for r in range(W):
            sh = shard.plan(F, W, r, halo)
            ps = shard.pieces(sh, K)
            assert ps[0][0] == 0 and ps[-1][1] == sh.n_feed and all(a[1] == b[0] for a, b in zip(ps, ps[1:]))
            own = [h for _, _, o0, n in ps for h in range(o0, o0 + n)]
            assert own == list(range(sh.n_own))


def test_c_shard_plan_matches_python():
    """include/bfcore.h bf_shard_plan / bf_shard_halo (what a C++ embedding calls) == beamform_amd.shard.plan / halo_frames."""
    import ctypes as C
    from beamform_amd import capi

    class CShard(C.Structure):
        _fields_ = [("lo", C.c_longlong), ("hi", C.c_longlong), ("warm", C.c_int), ("lead", C.c_int)]

    L = capi.load()
    L.bf_shard_plan.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(CShard)]
    for F, W, halo in [(10, 2, 1), (11, 3, 11), (65536, 8, 1), (262144, 8, 11), (7, 8, 0), (5, 1, 3)]:
        for r in range(W):
            cs = CShard()
            assert L.bf_shard_plan(F, W, r, halo, C.byref(cs)) == 0
            ps = shard.plan(F, W, r, halo)
            assert (cs.lo, cs.hi, cs.warm, cs.lead) == (ps.lo, ps.hi, ps.warm, ps.lead)
    assert L.bf_shard_plan(10, 2, 2, 1, C.byref(CShard())) != 0
    for algo in ("das", "phase", "mvdr", "lcmv", "gss", "phasempf", "mcra", "gsc"):
        p = make_params(algo, interf=(10.0,) if algo in ("lcmv", "gss") else ())
        cfg = capi.config_from_params(p)
        h = L.bf_shard_halo(C.byref(cfg))
        assert h == (shard.halo_frames(p) if shard.halo_frames(p) is not None else -1)
