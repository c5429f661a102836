#!/usr/bin/env python3
"""Cost of one sharded step (cold handle + lead + warm + owned frames) against a plain step of the same length, on one GPU:
what the weak-scaling line of bench.py --gpus N pays per rank beyond the kernel itself."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beamform_amd import shard
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
from beamform_amd.synth import stream_noise
M, F, world = 8, 65536, 4
p = make_params("das", n_mics=M)
halo = shard.halo_frames(p)
dev = torch.device("cuda", 0)
for rank in (0, 1):
    sh = shard.plan(world * F, world, rank, halo)
    x = stream_noise(1234, M, sh.first_feed_frame * 512, sh.hi * 512, device=dev)
    y = torch.empty(sh.n_feed * 512, device=dev)
    bf = Beamformer(p)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(50):
        shard.run_shard(bf, x.data_ptr(), sh, y.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        shard.run_shard(bf, x.data_ptr(), sh, y.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        bf.process_device(x.data_ptr(), sh.n_feed, y.data_ptr(), 0, s)
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / 200
    print(f"rank {rank}: n_feed {sh.n_feed} run_shard {dt*1e3:.4f} ms per step, plain process_device {dt2*1e3:.4f} ms")
