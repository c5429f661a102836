"""Parameter sets of the reference nodes, as plain dicts.

The values are the *launch-file* column of SURVEY.md App. B (the effective
defaults a user of balkce/beamform runs with), cited per key.  Both the product
binding (beamform_amd.capi) and the test oracle binding (oracle/) consume these
dicts; nothing here computes anything on the hot path.
"""
from __future__ import annotations

import copy

ALGOS = ("das", "mvdr", "lcmv", "gss", "phase", "phasempf", "mcra", "gsc")
ALGO_ID = {name: i for i, name in enumerate(ALGOS)}

#: beamform/beamform_config.yaml:20-35 ("aira16"), z dropped as util.h:82-92 does.
AIRA16_XY = [
    (0.158, 0.115), (0.158, -0.115), (-0.045, 0.000), (-0.050, -0.188),
    (-0.195, 0.000), (-0.057, 0.186), (0.180, 0.000), (0.158, -0.115),
    (0.056, -0.171), (-0.050, -0.188), (-0.128, -0.098), (-0.195, 0.000),
    (-0.132, 0.098), (-0.057, 0.186), (0.056, 0.171), (0.158, 0.115),
]
#: beamform/beamform_config.yaml:15-17 ("aira3", the uncommented default)
AIRA3_XY = [(0.000, 0.000), (0.000, -0.180), (-0.156, -0.090)]

#: launch/*.launch values ("launch" column)
LAUNCH_DEFAULTS = {
    "das": {},
    # launch/mvdr.launch:6-10
    "mvdr": dict(past_windows=10, freq_mag_threshold=0.001, freq_max=16000.0, freq_min=100.0, out_amp=1.0),
    # launch/lcmv.launch:6-11
    "lcmv": dict(past_windows=10, freq_mag_threshold=0.001, freq_max=16000.0, freq_min=100.0, out_amp=1.0),
    # launch/gss.launch:6-12
    "gss": dict(freq_mag_threshold=0.001, freq_max=16000.0, freq_min=100.0, out_amp=0.1, mu=0.001, lambda_=0.0),
    # launch/phase.launch:6 + phase.cpp:180,187 fallbacks (the launch file's
    # min_mag/smooth_size are never read by the node: SURVEY Q14)
    "phase": dict(min_phase=10.0, mag_mult=0.1, mag_threshold=0.05),
    # launch/phasempf.launch:6-21
    "phasempf": dict(min_phase=30.0, min_mag=0.05, smooth_size=3, mcra_alphaS=0.95, mcra_alphaD=0.95,
                     mcra_alphaD2=0.98, mcra_delta=0.001, mcra_L=50, mpf_alphaS=0.7, mpf_eta=0.3,
                     mpf_rev_gamma=0.9, mpf_rev_delta=1.0, out_amp=2.5, noise_floor=0.001,
                     out_only_noise=0, out_only_mcra=0),
    # launch/mcra.launch:6-12 (single-channel node)
    "mcra": dict(mcra_alphaS=0.95, mcra_alphaD=0.95, mcra_alphaD2=0.98, mcra_delta=0.001, mcra_L=300, out_amp=3.5,
                 out_only_noise=0),
    # launch/gsc.launch:6-11; write_mu is file I/O, not part of the path
    "gsc": dict(gsc_use_vad=0, gsc_vad_threshold=0.1, gsc_mu0=0.0001, gsc_mu_max=0.1, gsc_filter_size=128),
}

_BASE = dict(
    algo="das", n_mics=8, hop=512, sample_rate=48000.0, mics=None, theta=0.0, interf=(),
    past_windows=10, freq_mag_threshold=0.001, freq_max=16000.0, freq_min=100.0, out_amp=1.0,
    mu=0.001, lambda_=0.0, min_phase=10.0, mag_mult=0.1, mag_threshold=0.05, min_mag=0.05, smooth_size=3,
    mcra_alphaS=0.95, mcra_alphaD=0.95, mcra_alphaD2=0.98, mcra_delta=0.001, mcra_L=50,
    mpf_alphaS=0.7, mpf_eta=0.3, mpf_rev_gamma=0.9, mpf_rev_delta=1.0, noise_floor=0.001,
    out_only_noise=0, out_only_mcra=0,
    gsc_use_vad=0, gsc_vad_threshold=0.1, gsc_mu0=0.0001, gsc_mu_max=0.1, gsc_filter_size=128,
)


def make_params(algo: str, n_mics: int = 8, **overrides) -> dict:
    """Parameter dict for `algo` with the reference's launch-file values.

    `mics` defaults to the first `n_mics` entries of the aira16 layout
    (SURVEY.md 8d).  `interf` is the list of interferer angles (lcmv/gss).
    """
    if algo not in ALGO_ID:
        raise ValueError(f"unknown algo {algo!r}")
    p = copy.deepcopy(_BASE)
    p.update(LAUNCH_DEFAULTS[algo])
    p["algo"] = algo
    p["n_mics"] = n_mics
    p.update(overrides)
    if p["mics"] is None:
        if p["n_mics"] > len(AIRA16_XY):
            raise ValueError("n_mics > 16 needs an explicit mics=[(x,y),...]")
        p["mics"] = list(AIRA16_XY[: p["n_mics"]])
    p["mics"] = [tuple(map(float, xy)) for xy in p["mics"]]
    if len(p["mics"]) != p["n_mics"]:
        raise ValueError("len(mics) != n_mics")
    p["interf"] = [float(a) for a in p["interf"]]
    return p
