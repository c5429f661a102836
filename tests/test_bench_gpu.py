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
