"""The reference's closed loop (output window -> energy2theta -> /theta -> next callback) around the HIP node: the angles it
publishes follow the same loop around the oracle node."""
import numpy as np
import pytest

from beamform_amd import controllers
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("algo,ctl", [("das", "energy"), ("phase", "energy"), ("das", "diff")])
def test_closed_loop_follows_the_oracle_loop(algo, ctl):
    import oracle
    import torch
    from beamform_amd.capi import Beamformer
    assert torch.cuda.is_available()
    M, F = 4, 130
    p = make_params(algo, n_mics=M, theta=-15.0)
    x = make_scene(M, F, seed=47, theta_s=20.0, silent_frac=0.0)
    mk = (lambda: controllers.Energy2Theta(initial_angle=-15.0, num_win=30)) if ctl == "energy" else \
         (lambda: controllers.Energy2ThetaDiff(initial_angle=-15.0, num_win=30))
    y_ref, pub_ref = controllers.follow(oracle.OracleNode(p), x, mk())
    y, pub = controllers.follow(Beamformer(p), x, mk())
    assert len(pub) == len(pub_ref) > 90 and [k for k, _ in pub] == [k for k, _ in pub_ref]
    th, th_ref = np.array([t for _, t in pub]), np.array([t for _, t in pub_ref])
    assert np.ptp(th_ref) > 1e-3                                    # the controller really steers
    assert np.abs(th - th_ref).max() < 1e-3                         # degrees
    assert rel_l2(y, y_ref) < 1e-5


def test_window_energy_on_the_gpu_is_the_scripts_energy():
    """bf_stream_rms == get_energy_from_list (energy2theta.py:23-27) of the same window."""
    import torch
    from beamform_amd.capi import Beamformer
    M, F = 4, 6
    p = make_params("das", n_mics=M, theta=10.0)
    x = make_scene(M, F, seed=5, silent_frac=0.0)
    bf = Beamformer(p)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty(F * 512, dtype=torch.float32, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr())
    rms = bf.stream_rms(yd.data_ptr(), F)[0, 0]
    assert abs(rms - controllers.window_rms(yd.cpu().numpy())) < 1e-12
