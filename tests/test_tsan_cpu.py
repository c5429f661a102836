"""ThreadSanitizer run of libbfcore's HOST control plane (tests/host_tsan): one thread hammers bf_set_theta /
bf_set_interference / bf_get_weights / checkpoints while another drives bf_process_hop -- the reference's ROS-callback
thread against its JACK thread (das.cpp:94-99 vs :72-92, lcmv.cpp:258-309 vs :142-162).  g++ -fsanitize=thread build
of capi.cpp + config.cpp + pipeline.hip against a host stand-in for the HIP runtime; kernels are stubs that only read
the tables a launch was given and check that every batch saw ONE consistent {column count, steering table}."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_control_plane_is_race_free_under_tsan():
    d = os.path.join(ROOT, "tests", "host_tsan")
    subprocess.check_call(["make", "-C", d, "-s"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([os.path.join(d, "tsan_harness"), "300"], capture_output=True, text=True, timeout=600, env=env)
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count("0 inconsistent batches") == 4, r.stdout
