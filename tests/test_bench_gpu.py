"""bench.py contract on the GPU box: one JSON line with the agreed fields, at N=1 and through the N>1 launcher path.

The N>1 case uses the BF_BENCH_ONE_DEVICE=1 hook (every rank on cuda:0, gloo for the host-staged gather) because the
test box has one GPU; it exercises torch.distributed.run, the barriers, the max-over-ranks timing and the gather code."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--frames", "4096", "--steps", "3", "--warmup", "1", "--settle-ms", "0", "--no-extra"]


def _last_json(out: str) -> dict:
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _check_line(d, n, dtype="f64"):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "stft_frames_per_sec" and d["unit"] == "frames/s" and d["n_gpus"] == n
    assert d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    # the headline computes in double, as the reference does (das.cpp:16-24); --das-f32 selects the fused fp32 kernel
    assert d["vs_baseline"] is None and d["dtype"] == dtype and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["kernel"] == ("das_f64_pair_kernel" if dtype == "f64" else "das_fused_kernel")
    assert 0 < r["kernel_ms"] <= d["ms_per_step"] and r["kernel_launches_timed"] == 3   # event pairs inside the timed steps
    assert abs(d["value"] - n * 4096 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_bench_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-frames", "256", "--cpu-all-cores-frames", "512"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    _check_line(d, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "frames/s"
    usable = c["all_cores"]["host"]["usable"]
    assert c["all_cores"]["cores"] == usable <= len(os.sched_getaffinity(0))
    assert usable == 1 or c["all_cores"]["value"] > c["value"]


def test_bench_fp32_headline_on_request():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--no-cpu", "--das-f32"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    _check_line(_last_json(out.stdout), 1, dtype="f32")


def test_bench_two_ranks_through_the_launcher():
    env = dict(os.environ, BF_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL + ["--no-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    _check_line(d, 2)
    # the two ranks ran ONE 8192-frame stream through shard.plan (halo 1 + lead hop on rank 1), gather inside the second timing
    assert d["config"]["global_stream_frames"] == 8192 and d["config"]["frames_per_gpu"] == 4096
    assert d["value_including_final_gather"] is not None and 0 < d["value_including_final_gather"] <= d["value"] * 1.5
    # the default is --gather overlap: pieces sent to rank 0 while the next piece computes (host-staged over gloo on this one-GPU box)
    assert d["config"]["gather"] == "overlap" and d["value_including_overlapped_gather"] is not None
    assert 0 < d["value_including_overlapped_gather"] <= d["value"] * 1.5
    assert d["n_ranks_seen"] == 2 and "gather_error" not in d
    assert d["cpu_baseline"] is None  # rank 0 at N = 1 only


def test_bench_spawns_its_own_ranks_when_no_launcher_is_present():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must not silently run one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BF_BENCH_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--strong", "--total-frames", "6000"] + SMALL
                         + ["--no-cpu"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_stream_frames"] == 6000
    assert abs(d["value"] - 6000 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_bench_gather_watchdog_prints_the_compute_line():
    """A gather that does not finish (here: a time-out far below what two ranks need to build their channels) must not cost the
    run its compute figures -- and must not pass as a success: rank 0 prints the line with the compute figures and gather_error,
    then every rank exits with bench.GATHER_TIMEOUT_RC, so the launcher's return code is non-zero."""
    env = dict(os.environ, BF_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29655", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "65536", "--steps", "40", "--warmup", "1",
           "--settle-ms", "0", "--no-extra", "--no-cpu", "--gather-timeout-s", "0.05"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode != 0, "a gather that never finished must fail the run"
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "did not finish" in d["gather_error"]
    assert d["value_including_overlapped_gather"] is None


def test_bench_runs_the_rccl_code_path_on_one_rank():
    """BF_BENCH_FORCE_DIST=1: init_process_group("nccl") for real, the shard plan, the final gather (dist.gather of device tensors) and
    the overlapped gather with every piece sent to and received from rank 0 itself (grouped isend + irecv over RCCL) -- the calls a
    real N > 1 run makes, executed once on this one-GPU box.  The output of the overlapped walk must be the stream's output."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29672")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "8192"] + SMALL[2:] + ["--no-cpu"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 1 and "forced_dist" in d and d["n_ranks_seen"] == 1 and "gather_error" not in d
    assert d["config"]["gather"] == "overlap" and d["config"]["global_stream_frames"] == 8192
    assert d["value_including_final_gather"] and d["value_including_overlapped_gather"]


def test_self_loop_gather_reproduces_the_stream():
    """shard.run_shard_overlapped(self_loop=True) on a one-rank RCCL group: the pieces that went out and came back through
    ncclSend / ncclRecv are the output of the unsharded batch, bit for bit on the deterministic fp32 path."""
    code = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29673", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from beamform_amd import shard
from beamform_amd.capi import BF_DAS_FUSED_F32, Beamformer
from beamform_amd.params import make_params
p = make_params("das", n_mics=8)
F, H = 3001, 512
x = torch.rand(8, F * H, device="cuda") - 0.5
y = torch.empty(F * H, device="cuda"); out = torch.full((F * H,), float("nan"), device="cuda"); ref = torch.empty(F * H, device="cuda")
bf = Beamformer(p, das_impl=BF_DAS_FUSED_F32)
for w in shard.run_shard_overlapped(bf, x, y, F, 1, 0, shard.halo_frames(p), n_pieces=5, out=out, stream=torch.cuda.current_stream().cuda_stream, self_loop=True):
    w.wait()
torch.cuda.synchronize()
Beamformer(p, das_impl=BF_DAS_FUSED_F32).process_device(x.data_ptr(), F, ref.data_ptr())
torch.cuda.synchronize()
assert torch.equal(out, ref), float((out - ref).abs().max())
dist.destroy_process_group()
print("self-loop ok")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "self-loop ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_traffic_files_name_the_kernels_that_run():
    """bench.py prints a `traffic` figure from profiles/traffic_<tag>.json only when the kernels named in that file are the kernels the
    node launches today (bf_trace_begin / bf_trace_end): a counter figure beside a kernel it was not taken on is reported as
    traffic: null + traffic_stale.  Here: every committed traffic file must match the kernels of its configuration (re-take the file with
    tools/gpu_profile_all.sh after changing a default kernel), and a deliberately wrong launch list must be refused."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from beamform_amd.capi import BF_DAS_F64, BF_DAS_FUSED_F32, Beamformer, launch_trace
    from beamform_amd.params import make_params
    cases = {"das8_f64": ("das", 8, 65536, 1, (), BF_DAS_F64, 0), "das8": ("das", 8, 65536, 1, (), BF_DAS_FUSED_F32, 0),
             "mvdr8": ("mvdr", 8, 65536, 1, (), 1, 0), "phase8": ("phase", 8, 65536, 1, (), 1, 0),
             "phasempf8": ("phasempf", 8, 256, 256, (), 1, 0), "lcmv16": ("lcmv", 16, 32768, 1, (-60.0, 90.0, 150.0), 1, 0),
             "mvdr8_mixed": ("mvdr", 8, 65536, 1, (), 1, 1), "lcmv16_mixed": ("lcmv", 16, 32768, 1, (-60.0, 90.0, 150.0), 1, 1)}   # last: bf_config.precision
    for tag, (algo, M, F, S, interf, impl, prec) in cases.items():
        bf = Beamformer(make_params(algo, n_mics=M, interf=interf), n_streams=S, das_impl=impl, precision=prec)
        x = torch.rand((S, M, F * 512), device="cuda") - 0.5
        y = torch.empty((S, F * 512), device="cuda")
        bf.process_device(x.data_ptr(), F, y.data_ptr())
        with launch_trace() as t:
            bf.process_device(x.data_ptr(), F, y.data_ptr())
        torch.cuda.synchronize()
        bf.close()
        del x, y
        assert t.kernels, tag
        val, note = bench.load_traffic(tag, t.kernels)
        assert note is None and val and val > 0, (tag, note)
        val2, note2 = bench.load_traffic(tag, t.kernels + ["bf::some_other_kernel"])
        assert val2 is None and "was taken on" in note2


def test_config5_command_line_with_8_ranks_on_one_gpu(tmp_path):
    """BASELINE config 5 as the round driver would launch it on an 8-GPU node -- `bench.py --gpus 8 --strong --algo lcmv --mics 16` --
    rehearsed with every rank on cuda:0 (BF_BENCH_ONE_DEVICE=1: gloo, host-staged gather): 8 ranks joined, ONE 262 144-frame stream cut
    by shard.plan (P + 1 = 11 recomputed frames + a lead hop per rank), 7 x 64 MiB into rank 0, and the stream rank 0 assembled equals
    the unsharded stream on sampled windows (window starts at shard edges included).  Scaling itself stays unmeasured on hardware."""
    import numpy as np
    import torch
    from beamform_amd.capi import Beamformer
    from beamform_amd.params import make_params
    from beamform_amd.synth import stream_noise
    dump = str(tmp_path / "gathered.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BF_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29677")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--strong", "--algo", "lcmv", "--mics", "16",
                          "--steps", "2", "--warmup", "1", "--settle-ms", "0", "--no-extra", "--no-cpu", "--gather", "final",
                          "--gather-timeout-s", "600", "--dump-gathered", dump],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["scaling"] == "strong" and "gather_error" not in d
    assert d["config"]["global_stream_frames"] == 262144 and d["config"]["frames_per_gpu"] == 32768
    assert d["config"]["final_gather_bytes_into_rank0"] == 7 * 64 * 1024 * 1024
    assert d["value_including_final_gather"] is not None and d["dtype"] == "f64"
    got = np.load(dump)
    assert got.shape == (262144 * 512,)
    # the unsharded stream on this GPU (8 GiB of input, 34 GB of c128 spectra: resident), same counter-based noise
    F, M, H = 262144, 16, 512
    p = make_params("lcmv", n_mics=M, interf=(-60.0, 90.0, 150.0))
    x = stream_noise(1234, M, 0, F * H, device="cuda")
    y = torch.empty(F * H, device="cuda")
    Beamformer(p).process_device(x.data_ptr(), F, y.data_ptr())
    torch.cuda.synchronize()
    rng = np.random.default_rng(3)
    starts = [0, 11, 32768 - 5, 32768, 32768 + 11, 5 * 32768 - 1, 7 * 32768, F - 40] + [int(v) for v in rng.integers(12, F - 40, 8)]
    for t0 in starts:
        ref = y[t0 * H:(t0 + 40) * H].cpu().numpy()
        seg = got[t0 * H:(t0 + 40) * H]
        ok = np.isfinite(ref)   # the cold start's first frames invert a zero covariance (mvdr.cpp:228-232): NaN on both sides
        assert (np.isfinite(seg) == ok).all(), t0
        assert np.linalg.norm(seg[ok] - ref[ok]) <= 1e-5 * np.linalg.norm(ref[ok]), t0
