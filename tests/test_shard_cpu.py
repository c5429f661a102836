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
