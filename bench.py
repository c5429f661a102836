#!/usr/bin/env python3
"""bench.py -- STFT frames/s of the beamforming hot path on MI355X.

A "step" is one pass of the hot path (window -> FFTs -> per-bin weight-and-sum
-> IFFT -> overlap-add) over one batch of synthetic multichannel audio that is
already resident in HBM.  Workload at N=1: BASELINE.json configs[1]
(das, 8 mics, 1024-pt FFT / hop 512, 65536-frame batch).  With --gpus N the
launcher starts one rank per GPU (torch.distributed over RCCL); every rank
processes its own 65536-frame shard (weak scaling: independent frame ranges, no
data-path collective).  The timed region is the K steps of the hot path; after
it the per-rank output slabs of the last step are collected on rank 0 with one
RCCL gather ("final gather"), timed separately and reported as
config.final_gather_ms / value_including_final_gather (--gather step puts a
gather inside every step instead; --gather none skips it).

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for definitions.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HOP = 512
NFFT = 1024
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def algorithmic_bytes_per_frame(n_mics: int) -> int:
    """M*H*4 B of new input + H*4 B of output per frame."""
    return n_mics * HOP * 4 + HOP * 4


def cpu_baseline(algo: str, n_mics: int, frames: int):
    """The oracle (CPU restatement of the reference FFTW/Eigen path), one thread, bounded sample."""
    import numpy as np
    import oracle
    from beamform_amd.params import make_params
    from beamform_amd.synth import make_scene
    p = make_params(algo, n_mics=n_mics)
    reps = max(1, frames // 512)
    x = make_scene(n_mics, 512, seed=11)  # 512 frames, fed `reps` times as one continuing stream
    F = reps * (x.shape[1] // HOP)
    node = oracle.OracleNode(p)
    node.process(np.ascontiguousarray(x[:, : 64 * HOP]))  # warm caches / page in
    t0 = time.perf_counter()
    for _ in range(reps):
        node.process(x)
    dt = time.perf_counter() - t0
    return {
        "value": F / dt, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{algo} {n_mics}-mic hop512 fft1024, {F} frames of the seeded synthetic scene, "
                  f"{dt:.1f} s on 1 of {os.cpu_count()} host cores (oracle/bf_oracle.cpp; FFTW/Eigen/JACK/ROS absent "
                  "from the image, so the reference binary itself cannot be timed)",
    }


def _cpu_worker(job):
    """One process = one reference node on its own 512-frame chunks (frames of different streams are independent)."""
    algo, n_mics, reps, seed = job
    import numpy as np
    import oracle
    from beamform_amd.params import make_params
    from beamform_amd.synth import make_scene
    node = oracle.OracleNode(make_params(algo, n_mics=n_mics))
    x = make_scene(n_mics, 512, seed=seed)
    node.process(np.ascontiguousarray(x[:, : 32 * HOP]))
    t0 = time.perf_counter()
    for _ in range(reps):
        node.process(x)
    return reps * 512, time.perf_counter() - t0


def cpu_baseline_all_cores(algo: str, n_mics: int, frames_per_core: int):
    """The same oracle, one process per host core, each on its own stream; must run BEFORE this process touches the GPU
    (the pool forks)."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    reps = max(1, frames_per_core // 512)
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(algo, n_mics, reps, 100 + i % 7) for i in range(cores)], chunksize=1)
    wall = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": frames / busy, "unit": "frames/s", "cores": cores,
            "sample": f"{cores} processes x {reps * 512} frames each ({algo} {n_mics}-mic), slowest worker {busy:.1f} s, "
                      f"{wall:.1f} s wall incl. pool start-up and scene synthesis"}


def load_traffic(tag: str):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", f"traffic_{tag}.json")
    if os.path.exists(path):
        try:
            with open(path) as f:
                return json.load(f).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=200.0,
                    help="keep the GPU busy with untimed steps for this long before the W warmup steps: the clock/power\n"
                         "controller needs ~50 launches (25 ms) after an idle period before kernel durations are steady\n"
                         "(profiles/README.md r01_g); 0 disables")
    ap.add_argument("--algo", default="das", choices=["das", "mvdr", "lcmv", "gss", "phase", "phasempf", "mcra", "gsc"])
    ap.add_argument("--mics", type=int, default=8)
    ap.add_argument("--frames", type=int, default=65536, help="frames per GPU per step")
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--layout", default="planar", choices=["planar", "interleaved"])
    ap.add_argument("--gather", default="final", choices=["final", "step", "none"])
    ap.add_argument("--cpu-frames", type=int, default=81920,
                    help="frames in the CPU-baseline sample (0 = skip); the default is ~11 s of single-core work")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-all-cores-frames", type=int, default=2048,
                    help="frames per host core in the all-cores CPU baseline (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary mvdr measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from beamform_amd.capi import BF_INTERLEAVED, BF_PLANAR, Beamformer
    from beamform_amd.params import make_params

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # CPU baselines first (rank 0, N = 1 only): the all-cores pool forks, which must happen before this process
    # initialises the GPU; the untimed settle phase below brings the clocks back up afterwards
    cpu_line = None
    if world == 1 and not args.no_cpu and args.cpu_frames > 0:
        cpu_line = cpu_baseline(args.algo, args.mics, args.cpu_frames)
        if args.cpu_all_cores_frames > 0:
            try:
                cpu_line["all_cores"] = cpu_baseline_all_cores(args.algo, args.mics, args.cpu_all_cores_frames)
            except Exception as e:  # the single-core figure is the contract; this one is extra
                cpu_line["all_cores"] = {"error": str(e)}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    # test hook for single-GPU boxes: BF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo for the
    # (host-staged) gather, so the launcher / barrier / gather code path can be exercised without 2 GPUs
    one_dev = os.environ.get("BF_BENCH_ONE_DEVICE", "0") == "1"
    if one_dev:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():  # launcher narrowed the visible devices per rank (HIP_VISIBLE_DEVICES)
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def gather_to_rank0(t, out_list):
        if one_dev:
            host = t.cpu()
            dist.gather(host, [torch.empty_like(host) for _ in range(world)] if rank == 0 else None, dst=0)
        else:
            dist.gather(t, out_list, dst=0)

    def all_max(vals):
        t = torch.tensor(vals, device="cpu" if one_dev else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    M, F, S = args.mics, args.frames, args.streams
    interf = (-60.0, 90.0, 150.0) if args.algo in ("lcmv", "gss") else ()
    p = make_params(args.algo, n_mics=M, interf=interf)
    layout = BF_PLANAR if args.layout == "planar" else BF_INTERLEAVED
    bf = Beamformer(p, device=local_rank, n_streams=S, layout=layout)

    # synthetic input, resident in HBM before the timed region: uniform noise in [-0.5, 0.5)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    shape = (S, M, F * HOP) if layout == BF_PLANAR else (S, F * HOP, M)
    x = torch.rand(shape, device=dev, generator=g, dtype=torch.float32) - 0.5
    y = torch.empty((S, F * HOP), device=dev, dtype=torch.float32)
    gathered = None
    if world > 1 and args.gather != "none" and rank == 0 and not one_dev:
        gathered = [torch.empty_like(y) for _ in range(world)]
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream

    def step():
        bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, sptr)
        if world > 1 and args.gather == "step":
            gather_to_rank0(y, gathered)

    settle_launches = 0
    if args.settle_ms > 0:  # untimed: let DVFS settle (a cold chip runs the first ~50 launches up to 45 % slower)
        ts = time.perf_counter()
        while (time.perf_counter() - ts) * 1e3 < args.settle_ms:
            for _ in range(8):
                step()
            torch.cuda.synchronize(dev)
            settle_launches += 8
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gather_dt = 0.0
    if world > 1 and args.gather == "final":  # output delivery, outside the hot path: timed on its own
        gather_to_rank0(y, gathered)           # first call also builds the RCCL channels
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        g0 = time.perf_counter()
        gather_to_rank0(y, gathered)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        gather_dt = time.perf_counter() - g0
    if world > 1:
        dt, gather_dt = all_max([dt, gather_dt])

    # dominant-kernel duration: HIP events on the launch stream, one pair per launch
    k_iters = max(5, min(args.steps, 50))
    ms_call, ms_kernel = bf.time_device(x.data_ptr(), F, y.data_ptr(), k_iters, sptr)
    torch.cuda.synchronize(dev)

    # secondary line of the BASELINE metric ("DAS+MVDR"): mvdr 8-mic on the same input, rank 0, few steps
    extra = None
    if rank == 0 and args.algo == "das" and not args.no_extra and S == 1 and layout == BF_PLANAR:
        try:
            pm = make_params("mvdr", n_mics=M)
            bm = Beamformer(pm, device=local_rank)
            for _ in range(2):
                bm.process_device(x.data_ptr(), F, y.data_ptr(), 0, sptr)
            torch.cuda.synchronize(dev)
            ms_m, _ = bm.time_device(x.data_ptr(), F, y.data_ptr(), 5, sptr)
            extra = {"mvdr_frames_per_s": F / (ms_m * 1e-3), "mvdr_ms_per_step": ms_m,
                     # same algorithmic floor as das (spectra need never leave the chip in principle)
                     "mvdr_frac_of_hbm_roofline_algorithmic_bytes":
                         algorithmic_bytes_per_frame(M) * F / (ms_m * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "mvdr_workload": f"mvdr {M}-mic 1024-pt, {F}-frame batch, fp64 bin pipeline (compute-bound: "
                                      "per-bin covariance + Cholesky solve), launch-file parameters"}
            bm.close()
        except Exception as e:  # the headline must not die on the secondary measurement
            extra = {"mvdr_error": str(e)}

    if rank == 0:
        frames_total = world * S * F * args.steps
        value = frames_total / dt
        bpf = algorithmic_bytes_per_frame(M)
        units_per_launch = S * F
        # fused das: its one kernel; the other nodes run a chain of kernels (stft -> per-bin -> istft): the chain's duration
        k_ms = ms_kernel if ms_kernel > 0 else ms_call
        achieved = bpf * units_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        tag = f"{args.algo}{M}"
        out = {
            "metric": "stft_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.algo == "das" else "f64", "data": "synthetic",
            "config": {"workload": f"{args.algo} {M}-mic 1024-pt (hop 512), {F}-frame batch per GPU, {S} stream(s), "
                                   f"{args.layout} input resident in HBM", "frames_per_gpu": F, "mics": M, "fft": NFFT,
                       "hop": HOP, "streams": S, "layout": args.layout, "gather": args.gather if world > 1 else "n/a",
                       "final_gather_ms": gather_dt * 1e3 if world > 1 and args.gather == "final" else None,
                       "value_including_final_gather": (frames_total / (dt + gather_dt)) if gather_dt > 0 else None,
                       "parallelism": f"frame-sharded x{world}", "settle_launches_before_warmup": settle_launches},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # PMC traffic exists for the headline workload only (profiles/traffic_das8.json)
                         "traffic": load_traffic(tag) if (F == 65536 and S == 1 and args.layout == "planar") else None,
                         "kernel": "das_fused_kernel" if args.algo == "das" else "bin pipeline (stft + per-bin kernel + istft)",
                         "kernel_ms": k_ms, "call_ms": ms_call, "algorithmic_bytes_per_frame": bpf,
                         "frames_per_launch": units_per_launch, "frac_of_measured_copy_ceiling_6290": achieved / 6290.0},
        }
        out["cpu_baseline"] = cpu_line  # None at N > 1 (the contract asks for it on rank 0 at N = 1 only)
        if extra:
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
