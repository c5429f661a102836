"""The N > 1 job on the HIP path: one long stream cut by shard.plan (lead hop + warm frames + owned frames), every
logical rank a COLD handle, outputs concatenated -- run sequentially on the one GPU of the test box.

  * small seeded scenes: concatenation == unsharded HIP run == oracle (das bit-identical, mvdr/lcmv 1e-5);
  * BASELINE config 5 whole (lcmv 16-mic, K = 3, 262 144 frames, 8 shards): concatenation vs the unsharded HIP run at
    full size, and random windows of the unsharded run vs the oracle.
"""
import numpy as np
import pytest

from beamform_amd import shard
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene, stream_noise
from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_TIME = 1e-5  # north_star tolerance, relative L2


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def run_logical_shards(p, x_full, F, world):
    """x_full: torch [M, F*512] on the GPU.  Every logical rank gets its own cold Beamformer and only its slice."""
    torch = _torch()
    from beamform_amd.capi import Beamformer
    halo = shard.halo_frames(p)
    parts = []
    for r in range(world):
        sh = shard.plan(F, world, r, halo)
        bf = Beamformer(p)
        x_feed = x_full[:, sh.first_feed_frame * 512: sh.hi * 512].contiguous()   # what the rank would hold
        parts.append(shard.run_sharded(bf, x_feed, F, world, r, halo, gather=False).clone())
        torch.cuda.synchronize()
        bf.close()
    return torch.cat(parts)


@pytest.mark.parametrize("algo,M,interf,F,world", [
    ("das", 8, (), 203, 8), ("das", 3, (), 64, 5), ("phase", 4, (), 90, 4),
    ("mvdr", 8, (), 160, 8), ("mvdr", 4, (), 37, 2), ("lcmv", 8, (-60.0, 90.0), 150, 4), ("lcmv", 16, (-60.0, 90.0, 150.0), 120, 3)])
def test_logical_shards_equal_unsharded_and_oracle(algo, M, interf, F, world):
    import oracle
    torch = _torch()
    from beamform_amd.capi import Beamformer
    p = make_params(algo, n_mics=M, interf=interf, theta=20.0)
    x = make_scene(M, F, seed=700 + M + world)
    y_ref, _ = oracle.OracleNode(p).process(x)
    xd = torch.from_numpy(x).cuda()
    bf = Beamformer(p)
    yd = torch.empty(F * 512, dtype=torch.float32, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr())
    torch.cuda.synchronize()
    y_one = yd.cpu().numpy()
    y_cat = run_logical_shards(p, xd, F, world).cpu().numpy()
    assert y_cat.shape == y_one.shape
    ok = np.isfinite(y_ref)
    assert (np.isfinite(y_one) == ok).all()
    if algo in ("das", "phase"):
        # the fused DAS overlap-add and the phase mask are independent of how the stream is cut
        assert np.array_equal(y_cat, y_one)
    else:
        # a shard rebuilds its covariance from the recomputed frames (same terms, other summation start):
        # frames whose reference output is NaN (frame 0 of the stream: zero covariance) exist only in shard 0
        assert (np.isfinite(y_cat) == ok).all()
        assert rel_l2(y_cat[ok], y_one[ok]) < TOL_TIME
    assert rel_l2(y_cat[ok], y_ref[ok]) < TOL_TIME


def test_plan_without_lead_is_wrong():
    """The lead hop is not optional: feeding a cold handle from first_input_frame (the round-1 Shard fields) changes
    the first owned hop."""
    torch = _torch()
    from beamform_amd.capi import Beamformer
    M, F = 4, 40
    p = make_params("das", n_mics=M, theta=20.0)
    x = make_scene(M, F, seed=3)
    xd = torch.from_numpy(x).cuda()
    bf = Beamformer(p)
    yd = torch.empty(F * 512, dtype=torch.float32, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr())
    sh = shard.plan(F, 2, 1, shard.halo_frames(p))
    assert sh.lead == 1
    seg = xd[:, sh.first_input_frame * 512: sh.hi * 512].contiguous()            # no lead hop
    b2 = Beamformer(p)
    y2 = torch.empty(seg.shape[1], dtype=torch.float32, device="cuda")
    b2.process_device(seg.data_ptr(), sh.n_process, y2.data_ptr())
    torch.cuda.synchronize()
    first_owned = slice(sh.lo * 512, (sh.lo + 1) * 512)
    assert not torch.equal(y2[sh.warm * 512: (sh.warm + 1) * 512], yd[first_owned])
    assert torch.equal(y2[(sh.warm + 1) * 512:], yd[(sh.lo + 1) * 512:])          # everything after it is fine


def test_config5_whole_8_logical_shards():
    """BASELINE config 5: lcmv 16-mic, 3 interferers, one 262 144-frame stream, 8 shards of 32 768 owned frames
    (+ 11 warm + 1 lead), each on a cold handle; vs the unsharded HIP run and oracle windows of it."""
    torch = _torch()
    from beamform_amd.capi import Beamformer
    from test_pipeline_gpu import windows_vs_oracle
    M, F, world = 16, 262144, 8
    p = make_params("lcmv", n_mics=M, interf=(-60.0, 90.0, 150.0), theta=20.0)
    xd = stream_noise(55, M, 0, F * 512, device="cuda")                           # 8 GiB, one global stream
    bf = Beamformer(p)
    yd = torch.empty(F * 512, dtype=torch.float32, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr())
    torch.cuda.synchronize()
    bf.close()
    y_cat = run_logical_shards(p, xd, F, world)
    fin = torch.isfinite(yd)
    assert torch.equal(torch.isfinite(y_cat), fin)
    assert int((~fin).sum()) <= 2 * 512                                           # only the stream's first frame(s)
    # per-shard relative L2 against the unsharded run
    for r in range(world):
        sh = shard.plan(F, world, r, shard.halo_frames(p))
        a = y_cat[sh.lo * 512: sh.hi * 512].double()
        b = yd[sh.lo * 512: sh.hi * 512].double()
        m = fin[sh.lo * 512: sh.hi * 512]
        err = float(torch.linalg.norm((a - b)[m]) / torch.linalg.norm(b[m]))
        assert err < TOL_TIME, (r, err)
        # the first owned hop of every shard is where a wrong halo would show
        h = slice(0, 512)
        if r > 0:
            e0 = float(torch.linalg.norm(a[h] - b[h]) / torch.linalg.norm(b[h]))
            assert e0 < TOL_TIME, (r, e0)
    # windows of the unsharded run against the oracle, two of them straddling shard boundaries
    y = yd.cpu().numpy()
    rng = np.random.default_rng(8)
    starts = [int(t) for t in rng.integers(20, F - 10, size=3)] + [32768 - 3, 5 * 32768 - 2]
    import oracle
    warm, span = 14, 6
    for t in starts:
        seg = xd[:, (t - warm) * 512: (t + span) * 512].cpu().numpy()
        y_ref, _ = oracle.OracleNode(p).process(np.ascontiguousarray(seg))
        ref = y_ref[(warm + 1) * 512:]
        for got in (y[(t + 1) * 512: (t + span) * 512], y_cat[(t + 1) * 512: (t + span) * 512].cpu().numpy()):
            assert np.isfinite(got).all() and rel_l2(got, ref) < TOL_TIME
