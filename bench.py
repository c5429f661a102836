#!/usr/bin/env python3
"""bench.py -- STFT frames/s of the beamforming hot path on MI355X.

A "step" is one pass of the hot path (window -> FFTs -> per-bin weight-and-sum
-> IFFT -> overlap-add) over one batch of synthetic multichannel audio that is
already resident in HBM.

N = 1 (default): BASELINE.json configs[1] -- das, 8 mics, 1024-pt FFT / hop 512, one
65 536-frame batch, computed in DOUBLE like the reference (das.cpp:16-24: std::complex<double>, FFTW double plans):
das_f64_pair_kernel (das_f64_w64.hip).  `extra` carries the fused fp32 kernel on the same batch (`das_f32`, with its own roofline
block; --das-f32 makes it the headline) and one line per other BASELINE config (mvdr 8-mic, phasempf 256 x 256,
lcmv 16-mic K = 3 per-GPU shard).

N > 1 (`--gpus N`, one rank per GPU over torch.distributed / RCCL): ONE global stream is cut into
contiguous frame ranges by beamform_amd.shard.plan; every rank holds only its slice (its owned hops
plus the lead hop and the warm-up frames in front of them), starts from a COLD handle each step and drops the
warm-up output -- no data-path collective.  By default the stream has N x 65 536 frames (weak scaling: 65 536
owned frames per GPU); `--strong` keeps the total at --total-frames for every N.  `value` times the K compute
steps; `value_including_final_gather` times K steps that each end with the one RCCL gather of the owned slabs
onto rank 0 (north_star: "RCCL over xGMI only for the final gather").

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for definitions.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HOP = 512
NFFT = 1024
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
FP64_VECTOR_PEAK_TF = 78.6  # AMD's public MI355X FP64 vector figure (not in the guide)
GATHER_TIMEOUT_RC = 3       # exit code of every rank when the gather watchdog fires (the compute-only line is printed first)


def algorithmic_bytes_per_frame(n_mics: int) -> int:
    """M*H*4 B of new input + H*4 B of output per frame."""
    return n_mics * HOP * 4 + HOP * 4


def model_flops_per_frame(algo: str, M: int, K: int = 0, P: int = 10, in_band_bins: int = 339) -> float:
    """Algorithmic fp64 flop model: packed-real FFTs + the per-bin work in its Hermitian /
    Cholesky form.  8 flops per complex multiply-add.  Used only for the '% of FP64 vector peak' figures."""
    fft = 5.0 * NFFT * math.log2(NFFT)                       # 51 200 flops per complex FFT-1024
    flops = (M / 2.0 + 0.5) * fft                            # two real mics per forward FFT, two frames per inverse
    if algo in ("mvdr", "lcmv"):
        rhs = (K + 1) + 1                                    # constraint columns + the frame's x
        cmac = (2 * M * (M + 1) / 2                          # slide the covariance: one update + one downdate (lower triangle)
                + M ** 3 / 6.0                               # Cholesky of R o whiteR
                + rhs * M * M / 2.0                          # forward substitution of [C | x]
                + (K + 1) * (K + 2) / 2.0 * M + (K + 1) * M  # Gram U_C^H U_C and U_C^H u_x
                + (K + 1) ** 3 / 3.0)                        # the small (K+1)^2 solve
        flops += in_band_bins * cmac * 8.0
    elif algo == "das":
        flops += 514 * M * 8.0
    return flops


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:           # cgroup v2
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    usable = aff if quota is None else max(1, min(aff, int(math.floor(quota + 1e-9)) or 1))
    return {"os_cpu_count": os.cpu_count(), "affinity": aff, "cgroup_quota_cpus": quota, "usable": usable}


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
    node.process(np.ascontiguousarray(x[:, : 64 * HOP]))  # warm caches / page in / build the FFT plan
    t0 = time.perf_counter()
    for _ in range(reps):
        node.process(x)
    dt = time.perf_counter() - t0
    hc = host_cores()
    return {
        "value": F / dt, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{algo} {n_mics}-mic hop512 fft1024, {F} frames of the seeded synthetic scene, "
                  f"{dt:.1f} s on 1 host core (affinity {hc['affinity']}, cgroup quota {hc['cgroup_quota_cpus']}, "
                  f"os.cpu_count {hc['os_cpu_count']}); oracle/bf_oracle.cpp -O2 with a per-node FFT plan (radix-2, tables built "
                  "once like an FFTW plan); FFTW/Eigen/JACK/ROS are absent from the image, so the reference binary itself "
                  "cannot be timed",
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
    """The same oracle, one process per USABLE host core (affinity mask capped by the cgroup quota), each on its own
    stream; must run BEFORE this process touches the GPU (the pool forks)."""
    import multiprocessing as mp
    hc = host_cores()
    cores = hc["usable"]
    reps = max(1, frames_per_core // 512)
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(algo, n_mics, reps, 100 + i % 7) for i in range(cores)], chunksize=1)
    wall = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": frames / busy, "unit": "frames/s", "cores": cores, "host": hc,
            "sample": f"{cores} processes x {reps * 512} frames each ({algo} {n_mics}-mic), slowest worker {busy:.1f} s, "
                      f"{wall:.1f} s wall incl. pool start-up and scene synthesis"}


def git_head():
    """The tree this line was measured on: BF_GIT_HEAD (the GPU box has no .git: the job script exports it) or `git rev-parse`."""
    h = os.environ.get("BF_GIT_HEAD")
    if h:
        return h
    try:
        return subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def norm_kernel_name(n: str) -> str:
    """A kernel name as bf_trace_end spells it, from rocprofv3's spelling (possibly truncated): no 'void ', no parameter list, no
    anonymous namespaces, no 'bf::'."""
    n = n.strip()
    if n.startswith("void "):
        n = n[5:]
    n = n.replace("(anonymous namespace)::", "")
    depth = 0
    for i, c in enumerate(n):
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace("bf::", "").strip()


def load_traffic(tag: str, launched=None):
    """(HBM-side bytes per step from a committed rocprofv3 --pmc summary under profiles/, note).  The figure is handed out only when
    the kernels named INSIDE the file are the kernels this run just launched (`launched`: names from bf_trace_end); otherwise
    (None, why) -- a counter figure printed beside a kernel it was not measured on is worse than none."""
    path = os.path.join(ROOT, "profiles", f"traffic_{tag}.json")
    if not os.path.exists(path):
        return None, None
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception as e:
        return None, f"profiles/traffic_{tag}.json unreadable: {e}"
    names = d.get("kernel_names")
    if names is None:  # files written before the names were stored separately: the keys of "kernels" (rocprofv3's spelling, cut at 90)
        names = [k for k in d.get("kernels", {})] or ([d["kernel"]] if "kernel" in d else [])
    had = sorted({norm_kernel_name(k) for k in names})
    if launched is not None:
        now = sorted({norm_kernel_name(k) for k in launched})
        # (a file that names one kernel by a substring -- the round-1 format -- matches when that substring names a launched kernel)
        ok = (had == now) or (len(had) == 1 and "kernel" in d and any(had[0] in k for k in now) and len(now) == 1)
        if not ok:
            return None, (f"profiles/traffic_{tag}.json (git {d.get('git_head', '?')}) was taken on {had}; this run launched {now}")
    return d.get("hbm_bytes_per_launch"), None


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children through torch.distributed.run
    (this parent has not touched the GPU) and hand back their exit code."""
    # --standalone: the launcher picks a free rendezvous port itself (no bind-close-reuse race on a shared box)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    if rc != 0:  # the ranks' own output is above; say in one line that no result line follows
        print(json.dumps({"error": f"torch.distributed.run exited with {rc}: a rank failed, see its traceback above", "n_gpus": n}), flush=True)
    return rc


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
    ap.add_argument("--frames", type=int, default=65536, help="frames per GPU per step (per stream)")
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--layout", default="planar", choices=["planar", "interleaved"])
    ap.add_argument("--das-f64", action="store_true", help="das in double, the reference's arithmetic (BF_DAS_F64): the default")
    ap.add_argument("--das-f32", action="store_true", help="das through the fused fp32 kernel (BF_DAS_FUSED_F32) as the headline")
    ap.add_argument("--mixed", action="store_true", help="bf_config.precision = BF_PRECISION_MIXED (z48 spectra for mvdr / lcmv, fp32 backward transform); default: the reference's doubles")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: keep the global stream at --total-frames for every N (strong scaling)")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="--strong: frames of the global stream (default 262144 = BASELINE config 5's batch)")
    ap.add_argument("--independent", action="store_true",
                    help="N > 1: round-1 behaviour, every rank an independent random batch (no shard plan)")
    ap.add_argument("--gather", default="overlap", choices=["final", "overlap", "none"],
                    help="final: K steps that each end with one gather of the owned slabs onto rank 0 (serial with the compute);\n"
                         "overlap (default): additionally K steps whose slices are walked in --pieces pieces, each piece's owned hops sent\n"
                         "to rank 0 point-to-point while the next piece computes (shard.run_shard_overlapped)")
    ap.add_argument("--gather-timeout-s", type=float, default=180.0,
                    help="N > 1: if the gather timings (which follow the compute timing) have not finished after this long, rank 0 prints\n"
                         "the line with the compute figures and gather_error set, and every rank exits with code 3: a collective that\n"
                         "hangs on hardware this build never saw must not cost the run its compute figures, and must not pass as a success")
    ap.add_argument("--pieces", type=int, default=4)
    ap.add_argument("--dump-gathered", default="", help="rank 0 writes the stream its final gather assembled (float32 .npy): for tests")
    ap.add_argument("--cpu-frames", type=int, default=196608,
                    help="frames in the CPU-baseline sample (0 = skip); the default is ~12 s of single-core work (its oracle, with the FFT plan cached, does ~16 k frames per second)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-all-cores-frames", type=int, default=40960,
                    help="frames per host core in the all-cores CPU baseline (0 = skip): ~2.5 s per worker at the oracle's ~16 k frames/s")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary lines (other BASELINE configs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))          # nothing has touched the GPU in this process
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BF_BENCH_FORCE_DIST=1 with --gpus 1: run the multi-rank code path (init_process_group("nccl"), shard plan, cold handle per step,
    # final gather, overlapped gather with every piece going out and coming back through RCCL point-to-point) on a one-rank group, so that
    # the RCCL calls have executed on hardware before a real N > 1 run depends on them.  The figures of such a run are not a result.
    force_dist = world == 1 and os.environ.get("BF_BENCH_FORCE_DIST", "0") == "1"
    use_dist = world > 1 or force_dist

    import torch
    import torch.distributed as dist

    from beamform_amd import shard
    from beamform_amd.capi import BF_DAS_F64, BF_DAS_FUSED_F32, BF_INTERLEAVED, BF_PLANAR, BF_PRECISION_MIXED, BF_PRECISION_REFERENCE, Beamformer
    from beamform_amd.params import make_params
    from beamform_amd.synth import stream_noise

    # CPU baselines first (rank 0, N = 1 only): the all-cores pool forks, which must happen before this process
    # initialises the GPU; the untimed settle phase below brings the clocks back up afterwards
    cpu_line = None
    if world == 1 and not force_dist and not args.no_cpu and args.cpu_frames > 0:
        cpu_line = cpu_baseline(args.algo, args.mics, args.cpu_frames)
        if args.cpu_all_cores_frames > 0:
            try:
                cpu_line["all_cores"] = cpu_baseline_all_cores(args.algo, args.mics, args.cpu_all_cores_frames)
            except Exception as e:  # the single-core figure is the contract; this one is extra
                cpu_line["all_cores"] = {"error": str(e)}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    # test hook for single-GPU boxes: BF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo for the
    # (host-staged) gather, so the launcher / barrier / shard-plan / gather code path can be exercised without 2 GPUs
    one_dev = os.environ.get("BF_BENCH_ONE_DEVICE", "0") == "1"
    narrowed = any(k in os.environ for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"))
    if one_dev:
        local_rank = 0
    elif world > torch.cuda.device_count() and not narrowed:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) visible: two ranks on one GPU would halve each "
                         "other's numbers silently (BF_BENCH_ONE_DEVICE=1 is the test hook for that)")
    elif local_rank >= torch.cuda.device_count():  # launcher narrowed the visible devices per rank (HIP_VISIBLE_DEVICES)
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29671")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n_ranks_seen = 1
    if use_dist:  # every rank really joined the group: a sum of ones
        ones = torch.ones(1, device="cpu" if one_dev else dev, dtype=torch.float64)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())
        if n_ranks_seen != world:
            raise SystemExit(f"all_reduce of ones over the group gives {n_ranks_seen}, WORLD_SIZE is {world}")

    def all_max(vals):
        t = torch.tensor(vals, device="cpu" if one_dev else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    M, F, S = args.mics, args.frames, args.streams
    interf = (-60.0, 90.0, 150.0) if args.algo in ("lcmv", "gss") else ()
    p = make_params(args.algo, n_mics=M, interf=interf)
    layout = BF_PLANAR if args.layout == "planar" else BF_INTERLEAVED
    das_impl = BF_DAS_FUSED_F32 if (args.das_f32 and not args.das_f64) else BF_DAS_F64   # only das looks at it
    precision = BF_PRECISION_MIXED if args.mixed else BF_PRECISION_REFERENCE
    bf = Beamformer(p, device=local_rank, n_streams=S, layout=layout, das_impl=das_impl, precision=precision)
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream

    def identical_weight_rows(b):
        """Groups of microphones (other than the reference microphone) whose weight rows are identical: they share a forward transform in
        das_f64_pair_kernel.  The reference drops z (util.h:82-92), so microphones 1 and 7 of its aira16 array (beamform_config.yaml:21,27)
        -- the first 8 of which are this benchmark's geometry since round 1 -- have the same delay for every look direction."""
        import numpy as np
        w = b.weights()[:, :, 0]
        groups, seen = [], set()
        for m1 in range(1, w.shape[1]):
            if m1 in seen:
                continue
            g = [m1] + [m2 for m2 in range(m1 + 1, w.shape[1]) if np.array_equal(w[:, m1], w[:, m2])]
            if len(g) > 1:
                groups.append(g)
                seen.update(g)
        return groups
    dup_rows = identical_weight_rows(bf) if args.algo == "das" else []

    # ---- what this rank processes ------------------------------------------------------------------------------
    sharded = use_dist and not args.independent
    sh = None
    if sharded:
        halo = shard.halo_frames(p)
        if halo is None or S != 1 or layout != BF_PLANAR:
            raise SystemExit(f"{args.algo} recurses over frames (or streams/layout given): shards by stream only -- use --independent")
        F_total = (args.total_frames or 262144) if args.strong else world * F
        sh = shard.plan(F_total, world, rank, halo)
        # the rank's slice of the ONE global stream: lead hop + warm frames + owned frames (counter-based noise, so the
        # halo a rank re-reads is bit-identical to what its neighbour owns)
        x = stream_noise(1234, M, sh.first_feed_frame * HOP, sh.hi * HOP, device=dev)
        n_feed, n_own = sh.n_feed, sh.n_own
        y = torch.empty(n_feed * HOP, device=dev, dtype=torch.float32)
        own_max = max(shard.plan(F_total, world, r, halo).n_own for r in range(world))
        y_own = y[sh.n_drop * HOP: sh.n_drop * HOP + n_own * HOP]
        gathered = None
        if rank == 0 and args.gather != "none" and not one_dev:
            gathered = [torch.empty(own_max * HOP, device=dev, dtype=torch.float32) for _ in range(world)]
        frames_per_step_all_ranks = F_total
    else:
        # synthetic input, resident in HBM before the timed region: uniform noise in [-0.5, 0.5)
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        shape = (S, M, F * HOP) if layout == BF_PLANAR else (S, F * HOP, M)
        x = torch.rand(shape, device=dev, generator=g, dtype=torch.float32) - 0.5
        y = torch.empty((S, F * HOP), device=dev, dtype=torch.float32)
        n_feed = n_own = F
        frames_per_step_all_ranks = world * S * F

    last_gather = {"parts": None}   # rank 0: what the most recent final gather received, one tensor of own_max hops per rank

    def gather_owned():
        if one_dev:
            host = torch.zeros(own_max * HOP, dtype=torch.float32)
            host[: n_own * HOP] = y_own.cpu()
            parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            last_gather["parts"] = parts
        else:
            shard.gather_hops(y_own, F_total, world, rank, HOP, 0, out=gathered)
            last_gather["parts"] = gathered

    def step(with_gather=False):
        if sharded:
            shard.run_shard(bf, x.data_ptr(), sh, y.data_ptr(), sptr)   # cold handle, lead + warm + owned hops
            if with_gather:
                gather_owned()
        else:
            bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, sptr)

    def timed(n_steps, with_gather=False):
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(with_gather)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        return all_max([dt])[0] if use_dist else dt

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
    from beamform_amd.capi import launch_trace
    with launch_trace() as tr0:   # which kernels a step launches (untimed): `traffic` is printed only beside the kernels it was taken on
        step()
    launched_headline = tr0.kernels
    # the K timed steps, with every launch of the dominant kernel bracketed by its own HIP event pair on the launch stream: the
    # roofline duration comes from the very launches the step time covers (so kernel_ms <= ms_per_step by construction)
    bf.kernel_timing_begin()
    dt = timed(args.steps)
    ms_kernel, n_timed_launches = bf.kernel_timing_end()
    ms_call = dt / args.steps * 1e3
    def make_line(dt_g, dt_o, gather_error, extra):
        """The one JSON line (rank 0).  Called at the end -- or by the gather watchdog with what has been measured so far."""
        frames_total = frames_per_step_all_ranks * args.steps
        value = frames_total / dt
        bpf = algorithmic_bytes_per_frame(M)
        units_per_launch = S * n_feed
        # fused das: its one kernel; the other nodes run a chain of kernels (stft -> per-bin -> istft): the chain's duration
        k_ms = ms_kernel if ms_kernel > 0 else ms_call
        achieved = bpf * units_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        tag = f"{args.algo}{M}" + ("_f64" if das_impl == BF_DAS_F64 else "")
        traffic, traffic_note = (load_traffic(tag, launched_headline) if (n_feed == 65536 and S == 1 and args.layout == "planar") else (None, None))
        if sharded:
            shl = shard.plan(F_total, world, world - 1, halo)
            wl = (f"{args.algo} {M}-mic 1024-pt (hop 512), ONE {F_total}-frame stream frame-sharded x{world} by shard.plan "
                  f"(last rank: {shl.n_own} owned + {shl.warm} warm-up frames + {shl.lead} lead hop), cold handle per step, "
                  f"planar slices resident in HBM")
        else:
            wl = (f"{args.algo} {M}-mic 1024-pt (hop 512), {F}-frame batch per GPU, {S} stream(s), {args.layout} input resident in HBM")
        out = {
            "metric": "stft_frames_per_sec", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (sharded and args.strong) else "weak", "vs_baseline": None,
            "dtype": "f32" if (args.algo == "das" and das_impl == BF_DAS_FUSED_F32) else "f64", "data": "synthetic",
            "value_including_final_gather": (frames_total / dt_g) if dt_g else None,
            "ms_per_step_including_final_gather": (dt_g / args.steps * 1e3) if dt_g else None,
            "value_including_overlapped_gather": (frames_total / dt_o) if dt_o else None,
            "ms_per_step_including_overlapped_gather": (dt_o / args.steps * 1e3) if dt_o else None,
            "config": {"workload": wl, "frames_per_gpu": n_own, "frames_fed_per_gpu": n_feed,
                       "global_stream_frames": frames_per_step_all_ranks if sharded else None, "mics": M, "fft": NFFT,
                       "hop": HOP, "streams": S, "layout": args.layout,
                       "gather": (args.gather if sharded else "n/a"),
                       "final_gather_bytes_into_rank0": ((world - 1) * own_max * HOP * 4) if sharded else None,
                       "parallelism": (f"frame-sharded x{world} (shard.plan: halo recomputed locally, no data-path collective)"
                                       if sharded else f"independent batches x{world}"),
                       "input": "uniform noise in [-0.5, 0.5) (counter-based global stream)" if sharded else "uniform noise in [-0.5, 0.5)",
                       "settle_launches_before_warmup": settle_launches, "git_head": git_head(),
                       "geometry": (f"the first {M} microphones of the reference's aira16 array (beamform_config.yaml:20-35), z dropped as util.h:82-92 does"
                                    if M <= 16 else "explicit"),
                       "microphones_with_identical_weight_rows": dup_rows,
                       "forward_transforms_per_frame_pair": ((M - 1 - (1 if dup_rows else 0)) if (args.algo == "das" and das_impl == BF_DAS_F64 and args.layout == "planar") else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # PMC traffic exists for the profiled workloads only (profiles/traffic_<tag>.json), and only while the
                         # file's kernels are the ones this run launched
                         "traffic": traffic,
                         "kernel": ("das_fused_kernel" if das_impl == BF_DAS_FUSED_F32 else ("das_f64_pair_kernel" if args.layout == "planar" else "das_f64_ring_kernel")) if args.algo == "das"
                                   else "bin pipeline (stft + per-bin kernel + istft)",
                         "kernels_launched_per_step": sorted({norm_kernel_name(k) for k in launched_headline}),
                         "kernel_ms": k_ms, "call_ms": ms_call, "kernel_launches_timed": n_timed_launches,
                         "kernel_ms_source": "one HIP event pair per launch on the launch stream, inside the K timed steps",
                         "algorithmic_bytes_per_frame": bpf,
                         "frames_per_launch": units_per_launch, "frac_of_measured_copy_ceiling_6290": achieved / 6290.0},
        }
        if traffic_note:
            out["roofline"]["traffic_stale"] = traffic_note
        out["n_ranks_seen"] = n_ranks_seen
        if force_dist:
            out["forced_dist"] = ("BF_BENCH_FORCE_DIST=1: the N > 1 code path on a one-rank RCCL group (every piece of the overlapped gather sent "
                                  "to and received from rank 0 itself); an execution check, not a scaling result")
        if gather_error:
            out["gather_error"] = gather_error
        out["cpu_baseline"] = cpu_line  # None at N > 1 (the contract asks for it on rank 0 at N = 1 only)
        if extra:
            out["extra"] = extra
        return out

    dt_g = dt_o = None
    gather_state = {"dt_g": None, "dt_o": None, "error": None, "phase": "idle", "armed": False}
    watchdog = None
    if sharded and args.gather in ("final", "overlap") and args.gather_timeout_s > 0:
        import threading

        def on_gather_timeout():
            if not gather_state["armed"]:
                return
            gather_state["error"] = (f"gather phase '{gather_state['phase']}' did not finish within {args.gather_timeout_s:.0f} s: "
                                     "compute figures only")
            if rank == 0:
                print(json.dumps(make_line(gather_state["dt_g"], gather_state["dt_o"], gather_state["error"], None)), flush=True)
            else:
                time.sleep(2.0)  # grace: the launcher tears every rank down as soon as one exits non-zero; rank 0 prints first
            # the hung collective cannot be cancelled from here (every rank's own watchdog does the same), and a run whose gather never
            # finished is a FAILED run: the compute-only line is on stdout, the exit code says so (GATHER_TIMEOUT_RC)
            os._exit(GATHER_TIMEOUT_RC)

        watchdog = threading.Timer(args.gather_timeout_s, on_gather_timeout)
        watchdog.daemon = True
    if sharded and args.gather in ("final", "overlap"):
        if watchdog is not None:
            gather_state["armed"], gather_state["phase"] = True, "final"
            watchdog.start()
        step(True)                                  # first gather also builds the RCCL channels
        dt_g = timed(args.steps, with_gather=True)  # K steps, each ending with the final gather onto rank 0
        gather_state["dt_g"] = dt_g
    if sharded and args.gather == "overlap":
        gather_state["phase"] = "overlap"
        out_full = torch.empty(F_total * HOP, device=dev, dtype=torch.float32) if rank == 0 else None
        def step_overlapped():
            for w in shard.run_shard_overlapped(bf, x, y, F_total, world, rank, halo, n_pieces=args.pieces, out=out_full, stream=sptr,
                                                host_staged=one_dev, self_loop=force_dist):
                w.wait()                            # stream-level wait: the next step's kernels queue behind it
        step_overlapped()
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_overlapped()
        torch.cuda.synchronize(dev)
        dist.barrier()
        dt_o = all_max([time.perf_counter() - t0])[0]
        gather_state["dt_o"] = dt_o
    gather_state["armed"] = False
    if watchdog is not None:
        watchdog.cancel()
    torch.cuda.synchronize(dev)
    if sharded and rank == 0 and args.dump_gathered and last_gather["parts"] is not None:
        import numpy as np
        owns = [shard.plan(F_total, world, r, halo).n_own for r in range(world)]
        np.save(args.dump_gathered, np.concatenate([last_gather["parts"][r][: owns[r] * HOP].cpu().numpy() for r in range(world)]))

    # ---- secondary lines: the other BASELINE configs, rank 0 at N = 1 -------------------------------------------
    def distinct_mics(M_):
        from beamform_amd.params import AIRA16_XY
        out = []
        for xy in AIRA16_XY:
            if xy not in out:
                out.append(xy)
        return out[:M_] if len(out) >= M_ else None

    def node_line(algo, M_, F_, S_, interf_=(), das_impl_=BF_DAS_FUSED_F32, iters=5, xin=None, note="", layout_=BF_PLANAR,
                  with_traffic=True, roofline_kernel=None, traffic_tag=None, precision_=BF_PRECISION_REFERENCE, mics_=None):
        pm = make_params(algo, n_mics=M_, interf=interf_, **({"mics": mics_} if mics_ else {}))
        bm = Beamformer(pm, device=local_rank, n_streams=S_, das_impl=das_impl_, layout=layout_, precision=precision_)
        if xin is None:
            gg = torch.Generator(device=dev).manual_seed(4321)
            xin = torch.rand((S_, M_, F_ * HOP), device=dev, generator=gg, dtype=torch.float32) - 0.5
        yo = torch.empty((S_, F_ * HOP), device=dev, dtype=torch.float32)
        # untimed until the clocks have settled under THIS node's load (the first ~15 launches after a change of kernel mix run up
        # to 10 % slower: tools/time_scene.py), like the headline's settle phase
        ts = time.perf_counter()
        n_settle = 0
        while n_settle < 2 or (time.perf_counter() - ts) < 0.12:
            bm.process_device(xin.data_ptr(), F_, yo.data_ptr(), 0, sptr)
            torch.cuda.synchronize(dev)
            n_settle += 1
        with launch_trace() as trn:
            bm.process_device(xin.data_ptr(), F_, yo.data_ptr(), 0, sptr)
        torch.cuda.synchronize(dev)
        # the median of three batches of `iters` launches (a secondary line; one batch now and then reads 10 % high -- both mvdr lines of one
        # un-profiled run of round 6 did, + 0.23 ms each, with the profiled run on the same box at the usual figure)
        ms, ms_k = sorted(bm.time_device(xin.data_ptr(), F_, yo.data_ptr(), iters, sptr) for _ in range(3))[1]
        bm.close()
        fr = S_ * F_
        bpf = algorithmic_bytes_per_frame(M_)
        line = {"workload": f"{algo} {M_}-mic 1024-pt, {S_} stream(s) x {F_} frames" + (f", {len(interf_)} interferers" if interf_ else "")
                            + ("" if not note else "; " + note),
                "ms_per_step": ms, "frames_per_s": fr / (ms * 1e-3), "timing": f"median of 3 batches of {iters} launches",
                "frac_of_hbm_roofline_algorithmic_bytes": bpf * fr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_frame": bpf,
                "arithmetic": ("fp32 (BF_DAS_FUSED_F32)" if (algo == "das" and das_impl_ == BF_DAS_FUSED_F32) else
                               "f64 throughout, like the reference (BF_PRECISION_REFERENCE)" if precision_ == BF_PRECISION_REFERENCE else
                               "f64 per-bin stage; z48 spectra (36-bit mantissa) in HBM for mvdr / lcmv, fp32 backward transform (BF_PRECISION_MIXED)")}
        if algo in ("mvdr", "lcmv") or das_impl_ == BF_DAS_F64:
            fl = model_flops_per_frame(algo, M_, len(interf_), pm["past_windows"])
            line["model_flops_per_frame"] = fl
            line["frac_of_fp64_vector_peak"] = fl * fr / (ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TF
        if roofline_kernel and ms_k > 0:  # a one-kernel node: the same block the headline carries (event pair per launch, on the launch stream)
            ach = bpf * fr / (ms_k * 1e-3) / 1e9
            tr_b, tr_note = load_traffic(traffic_tag, trn.kernels) if traffic_tag else (None, None)
            line["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "traffic": tr_b, "kernel": roofline_kernel,
                                "kernel_ms": ms_k, "call_ms": ms, "kernel_launches_timed": iters,
                                "algorithmic_bytes_per_frame": bpf, "frames_per_launch": fr,
                                "frac_of_measured_copy_ceiling_6290": ach / 6290.0}
            if tr_note:
                line["roofline"]["traffic_stale"] = tr_note
        line["kernels"] = sorted({norm_kernel_name(k) for k in trn.kernels})
        if layout_ != BF_PLANAR or not with_traffic:  # the committed counter files are for the noise input, planar
            return line
        tag = traffic_tag or {"das": "das8_f64", "mvdr": "mvdr8", "phasempf": "phasempf8", "phase": "phase8", "lcmv": "lcmv16"}.get(algo)
        if tag and precision_ == BF_PRECISION_MIXED:
            tag += "_mixed"
        tr, note = load_traffic(tag, trn.kernels) if tag else (None, None)
        if tr is not None:
            line["traffic"] = tr
        if note:
            line["traffic"] = None
            line["traffic_stale"] = note
        return line

    extra = None
    if rank == 0 and world == 1 and not force_dist and args.algo == "das" and not args.no_extra and S == 1 and layout == BF_PLANAR:
        extra = {}
        noise = ("input = uniform noise in [-0.5, 0.5): every in-band bin passes the magnitude gate, so every bin-frame "
                 "takes the covariance solve (the worst case; real scenes close part of the gates)")
        jobs = [
            ("mvdr", lambda: node_line("mvdr", M, F, 1, xin=x, note="BASELINE config 3; the library default: c128 spectra in HBM, fp64 backward transform; " + noise)),
            ("mvdr_mixed", lambda: node_line("mvdr", M, F, 1, xin=x, precision_=BF_PRECISION_MIXED,
                                             note="BASELINE config 3 with the BF_PRECISION_MIXED opt-in: z48 spectra, fp32 backward transform; " + noise)),
            ("das_f32" if das_impl == BF_DAS_F64 else "das_f64",
             (lambda: node_line("das", M, F, 1, das_impl_=BF_DAS_FUSED_F32, xin=x, iters=20, roofline_kernel="das_fused_kernel", traffic_tag="das8",
                                note="the headline batch through the fused fp32 kernel (das_fused_kernel): meets north_star's 1e-5 (1.5e-7 observed) but "
                                     "computes in single precision where the reference computes in double"))
             if das_impl == BF_DAS_F64 else
             (lambda: node_line("das", M, F, 1, das_impl_=BF_DAS_F64, xin=x, iters=20, roofline_kernel="das_f64_pair_kernel", traffic_tag="das8_f64",
                                note="same precision as the reference: das_f64_pair_kernel, one launch"))),
            ("das_f64_distinct_rows", lambda: node_line("das", M, F, 1, das_impl_=BF_DAS_F64, xin=x, iters=20, roofline_kernel="das_f64_pair_kernel",
                                                        with_traffic=False, mics_=distinct_mics(M),
                                                        note="the headline batch on a geometry WITHOUT coinciding microphones (aira16's microphone 7 "
                                                             "replaced by its microphone 8): every microphone but the reference one gets its own forward "
                                                             "transform: 4.0 transforms per frame where the headline geometry (microphones 1 and 7 share "
                                                             "x, y and so their weight row) needs 3.5")),
            ("das_interleaved", lambda: node_line("das", M, F, 1, xin=x.reshape(1, F * HOP, M), layout_=BF_INTERLEAVED, iters=20,
                                                  note="the headline workload with [sample][mic] input (same bytes read as interleaved samples), fused fp32 kernel")),
            ("das_f64_interleaved", lambda: node_line("das", M, F, 1, das_impl_=BF_DAS_F64, xin=x.reshape(1, F * HOP, M), layout_=BF_INTERLEAVED,
                                                      iters=20, with_traffic=False,
                                                      note="the headline workload in double on [sample][mic] input: das_f64_ring_kernel (the headline "
                                                           "kernel's body; the wavefront that draws a frame pair transposes the pair's two new hops through "
                                                           "its exchange plane into the block's ring of planar hop slots and reads them from there)")),
            ("phasempf", lambda: node_line("phasempf", 8, 256, 256, note="BASELINE config 4: 256 streams x 256 frames, recursion per stream")),
            ("phase", lambda: node_line("phase", M, F, 1, xin=x,
                                        note="uniform noise of this level stays below the node's mag_threshold (0.05): every bin takes the cheap branch "
                                             "without the phase-difference test; see phase_gate_open for the other extreme")),
            ("phase_gate_open", lambda: node_line("phase", M, F, 1, xin=x * 8.0,
                                                  note="the same noise 8x louder: every bin passes mag_threshold and runs the 8 atan2 + 28 wrapped differences")),
            ("lcmv16", lambda: node_line("lcmv", 16, 32768, 1, (-60.0, 90.0, 150.0), iters=3,
                                         note="BASELINE config 5, one GPU's shard of the 262144-frame stream; the library default (c128 spectra, fp64 backward transform); " + noise)),
            ("lcmv16_mixed", lambda: node_line("lcmv", 16, 32768, 1, (-60.0, 90.0, 150.0), iters=3, precision_=BF_PRECISION_MIXED,
                                               note="BASELINE config 5 shard with the BF_PRECISION_MIXED opt-in; " + noise)),
            ("gss", lambda: node_line("gss", 8, 256, 256, (-60.0, 90.0), with_traffic=False,
                                      note="gss 8-mic, 2 interferers, 256 streams x 256 frames (the demixing matrices recurse over the frames of a stream); " + noise)),
            ("gsc", lambda: node_line("gsc", 8, 64, 256, iters=3, with_traffic=False,
                                      note="gsc 8-mic (SURVEY 8(f) row 1), 256 streams x 64 frames: STFT + per-microphone alignment + fp64 ISTFT, then the "
                                           "sample-serial float NLMS with the 128 taps of the 7 blocking branches over a wavefront's lanes; " + noise)),
            ("lcmv8", lambda: node_line("lcmv", M, F, 1, (-60.0, 90.0), xin=x, with_traffic=False,
                                        note="lcmv on the headline array (8 microphones, 2 interferers): mvdr_fast_kernel<8, 3>; " + noise)),
        ]
        def scene_input(M_, F_):
            # SURVEY 8(d)'s seeded scene (directional band-limited target at 20 deg + 3 interferers + sensor noise, last 10 % silent),
            # 2048 frames tiled to the batch length: part of the magnitude gates are closed, as in real recordings
            from beamform_amd.synth import make_scene
            base = torch.from_numpy(make_scene(M_, 2048, seed=77)).to(dev)
            return base.repeat(1, F_ // 2048).reshape(1, M_, F_ * HOP).contiguous()

        scene_note = ("input = the seeded synthetic scene of SURVEY 8(d) (beamform_amd.synth.make_scene, 2048 frames tiled): target + 3 "
                      "interferers + sensor noise, 10 % near-silent frames; gates partly closed")
        jobs += [
            ("mvdr_scene", lambda: node_line("mvdr", M, F, 1, xin=scene_input(M, F), with_traffic=False,
                                             note="BASELINE config 3 on a realistic scene; " + scene_note)),
            ("phase_scene", lambda: node_line("phase", M, F, 1, xin=scene_input(M, F), with_traffic=False, note=scene_note)),
            ("lcmv16_scene", lambda: node_line("lcmv", 16, 32768, 1, (-60.0, 90.0, 150.0), iters=3, xin=scene_input(16, 32768), with_traffic=False,
                                               note="BASELINE config 5 shard on a realistic scene; " + scene_note)),
        ]

        def other_period_line(hop_, algo_="das"):
            # JACK periods other than 512 frames (rosjack.cpp:131; fft_win = 2 * period): one fused fp32 kernel on LDS-staged radix-4
            # transforms (das_fused_gen.hip; period 1024: das_fused.hip's wavefront-per-frame kernel; below 512: das_fused.hip's group mode); same number of SAMPLES as the headline batch
            pm = make_params(algo_, n_mics=M, hop=hop_)
            F_ = F * HOP // hop_
            bm = Beamformer(pm, device=local_rank, das_impl=BF_DAS_FUSED_F32)   # (only das looks at it: the fp32 opt-in's kernels)
            xin = x.reshape(1, M, F * HOP)
            yo = torch.empty((1, F_ * hop_), device=dev, dtype=torch.float32)
            ts = time.perf_counter()
            n_settle = 0
            while n_settle < 2 or (time.perf_counter() - ts) < 0.12:
                bm.process_device(xin.data_ptr(), F_, yo.data_ptr(), 0, sptr)
                torch.cuda.synchronize(dev)
                n_settle += 1
            ms, _ = bm.time_device(xin.data_ptr(), F_, yo.data_ptr(), 5, sptr)
            bm.close()
            if algo_ != "das":  # the fp64 chain of that node at this period: STFT and backward transform in registers, the history-free nodes in one launch
                return {"workload": f"{algo_} {M}-mic, JACK period {hop_} (FFT {2 * hop_}), {F_} frames = the headline batch's samples; fp64 bin pipeline on "
                                    "register-resident transforms (stft_istft.hip stft_small / stft_split / istft_small / istft_split kernels; phase: "
                                    "mask_kernels.hip stft_bins_small / stft_bins_split kernel)",
                        "ms_per_step": ms, "frames_per_s": F_ / (ms * 1e-3), "samples_per_s": F_ * hop_ / (ms * 1e-3)}
            how = ("one register-resident 2048-point transform per frame on a full wavefront (das_fused.hip das_fused_wave2048_kernel)" if hop_ == 1024 else
                   "1024 / N frames interleaved into one pass of the register-resident 1024-point machinery (das_fused.hip, the period-512 kernel in group mode)" if hop_ < 512 else
                   "LDS-staged radix-4 transforms (das_fused_gen.hip)")
            return {"workload": f"{algo_} {M}-mic, JACK period {hop_} (FFT {2 * hop_}), {F_} frames = the headline batch's samples; fused fp32 kernel on " + how,
                    "ms_per_step": ms, "frames_per_s": F_ / (ms * 1e-3), "samples_per_s": F_ * hop_ / (ms * 1e-3)}

        def dirs_line(D_):
            # D_ look directions of the headline batch in one call (a controller scanning candidate angles, scripts/energy2theta.py:62-101):
            # one set of forward transforms per frame serves all of them (das_fused_dirs_kernel)
            pm = make_params("das", n_mics=M)
            bm = Beamformer(pm, device=local_rank, n_dirs=D_, das_impl=BF_DAS_FUSED_F32)
            bm.set_thetas([-180.0 + 360.0 * d / D_ for d in range(D_)])
            yo = torch.empty((D_, F * HOP), device=dev, dtype=torch.float32)
            ts = time.perf_counter()
            n_settle = 0
            while n_settle < 2 or (time.perf_counter() - ts) < 0.12:
                bm.process_device(x.data_ptr(), F, yo.data_ptr(), 0, sptr)
                torch.cuda.synchronize(dev)
                n_settle += 1
            ms, _ = bm.time_device(x.data_ptr(), F, yo.data_ptr(), 5, sptr)
            bm.close()
            return {"workload": f"das {M}-mic 1024-pt, {F} frames, {D_} look directions from one set of forward transforms (das_fused_dirs_kernel)",
                    "ms_per_step": ms, "frames_per_s": F / (ms * 1e-3), "beam_frames_per_s": F * D_ / (ms * 1e-3),
                    "ratio_to_one_direction_fp32": (ms / one_dir_f32_ms()) if one_dir_f32_ms() else None}

        def one_dir_f32_ms():  # the fused fp32 kernel on one direction of the same batch: this run's own figure
            if das_impl == BF_DAS_FUSED_F32:
                return dt / args.steps * 1e3
            return extra.get("das_f32", {}).get("ms_per_step")

        jobs.append(("das_dirs16", lambda: dirs_line(16)))
        jobs.append(("das_period256", lambda: other_period_line(256)))
        jobs.append(("das_period1024", lambda: other_period_line(1024)))
        jobs.append(("phase_period256", lambda: other_period_line(256, "phase")))
        jobs.append(("mvdr_period256", lambda: other_period_line(256, "mvdr")))

        def resample_line():
            # the output stage's sample-rate converter on the batch the headline step just produced (rosjack.cpp:311-338)
            from beamform_amd.capi import Resampler
            rs = Resampler(48000, 16000)
            n_in = F * HOP
            cap = rs.out_count(n_in)
            yo = torch.empty(cap, device=dev, dtype=torch.float32)
            ysrc = y.reshape(-1)[:n_in]
            for _ in range(2):
                rs.reset()
                rs.process_device(ysrc.data_ptr(), n_in, yo.data_ptr(), cap, sptr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
            for _ in range(5):
                rs.reset()
                rs.process_device(ysrc.data_ptr(), n_in, yo.data_ptr(), cap, sptr)
            e1.record(torch.cuda.current_stream(dev))
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / 5
            rs.close()
            return {"workload": f"sinc sample-rate converter 48 kHz -> 16 kHz of one {F}-hop output batch ({2 * rs_lat(48000, 16000)} fp64 taps per output sample)",
                    "ms_per_step": ms, "hops_per_s": F / (ms * 1e-3), "GBps_in_plus_out": (n_in + cap) * 4 / (ms * 1e-3) / 1e9}

        def rs_lat(a_, b_):
            return int(round(2464 / 128 * max(1.0, a_ / b_))) + 1

        def hop_line():
            # the drop-in path itself: one jack_callback per call (host buffers in, host buffer out), 512-frame period
            import numpy as np
            out = {"workload": "bf_process_hop, one 512-frame period per call from host buffers (the reference's jack_callback); "
                               "median wall time per call; the JACK real-time budget at 48 kHz is 10 667 us"}
            for algo_ in ("das", "mvdr", "phasempf"):
                bh = Beamformer(make_params(algo_, n_mics=M), device=local_rank)
                seg = np.random.default_rng(3).standard_normal((M, HOP)).astype(np.float32) * 0.1
                for _ in range(30):
                    bh.process_hop(seg)
                ts = []
                for _ in range(200):
                    t0 = time.perf_counter()
                    bh.process_hop(seg)
                    ts.append(time.perf_counter() - t0)
                bh.close()
                ts.sort()
                out[f"{algo_}_us_per_callback"] = ts[len(ts) // 2] * 1e6
            return out

        jobs.append(("resample_48k_16k", resample_line))
        jobs.append(("streaming_callback", hop_line))
        for name, job in jobs:
            try:
                extra[name] = job()
            except Exception as e:  # the headline must not die on a secondary measurement
                extra[name] = {"error": str(e)}
        # round-1 field names kept for continuity (mvdr = the library default: the reference's arithmetic)
        if "ms_per_step" in extra.get("mvdr", {}):
            extra["mvdr_ms_per_step"] = extra["mvdr"]["ms_per_step"]
            extra["mvdr_frames_per_s"] = extra["mvdr"]["frames_per_s"]
            extra["mvdr_frac_of_hbm_roofline_algorithmic_bytes"] = extra["mvdr"]["frac_of_hbm_roofline_algorithmic_bytes"]
            extra["mvdr_frac_of_fp64_vector_peak"] = extra["mvdr"]["frac_of_fp64_vector_peak"]

        # one figure per secondary line, LAST in the line: the round driver's record keeps the tail of it (tools/bench_tables.py reads it there)
        extra["ms"] = {k: round(v["ms_per_step"], 4) for k, v in extra.items() if isinstance(v, dict) and isinstance(v.get("ms_per_step"), float)}

    if rank == 0:
        print(json.dumps(make_line(dt_g, dt_o, gather_state["error"], extra)), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
