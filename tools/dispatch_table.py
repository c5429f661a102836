#!/usr/bin/env python3
"""Which kernels run for (node, JACK period, input layout, microphones, look directions, spectrum dump): one batch per row under
bf_trace_begin / bf_trace_end (every launch of the library goes through BF_LAUNCH, csrc/launch_trace.hpp), so the table is what the
library DID, not what a document says it does.  Needs a GPU.  `make dispatch-table` (= python tools/dispatch_table.py docs/DISPATCH.md)
rewrites docs/DISPATCH.md; DESIGN.md section 3 points there."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from beamform_amd.capi import (BF_DAS_F64, BF_DAS_FUSED_F32, BF_INTERLEAVED, BF_PLANAR, BF_PRECISION_MIXED, BF_PRECISION_REFERENCE, Beamformer,
                                launch_trace)
from beamform_amd.params import make_params

FRAMES = 96   # enough for every kernel's main path (chunks, tiles, rings) to be entered


def kernels_of(algo, hop=512, layout=BF_PLANAR, M=8, dirs=1, dump=False, impl=BF_DAS_FUSED_F32, interf=(), prec=BF_PRECISION_REFERENCE, streams=1):
    over = {}
    if M > 16:  # beyond the yaml's 16 positions: a circle of 0.2 m
        import math
        over["mics"] = [(0.2 * math.cos(2 * math.pi * m / M), 0.2 * math.sin(2 * math.pi * m / M)) for m in range(M)]
    p = make_params(algo, n_mics=M, hop=hop, interf=interf, **over)
    bf = Beamformer(p, layout=layout, das_impl=impl, n_dirs=dirs, precision=prec, n_streams=streams)
    if dirs > 1:
        bf.set_thetas([-180.0 + 360.0 * d / dirs for d in range(dirs)])
    F = FRAMES
    shape = (streams, M, F * hop) if layout == BF_PLANAR else (streams, F * hop, M)
    x = torch.rand(shape, device="cuda") - 0.5
    y = torch.empty((streams * dirs, F * hop), device="cuda")
    spec = torch.empty((streams * dirs, F, 2 * hop, 2), device="cuda", dtype=torch.float64) if dump else None
    bf.process_device(x.data_ptr(), F, y.data_ptr(), spec.data_ptr() if dump else 0)   # first call: table uploads etc.
    with launch_trace() as t:
        bf.process_device(x.data_ptr(), F, y.data_ptr(), spec.data_ptr() if dump else 0)
    torch.cuda.synchronize()
    bf.close()
    out = collections.OrderedDict()
    for k in t.kernels:
        k = k.replace("bf::", "")
        out[k] = out.get(k, 0) + 1
    return " + ".join(k if n == 1 else f"{n} x {k}" for k, n in out.items())


def main():
    rows = []
    nodes = [("das (double)", "das", BF_DAS_F64, ()), ("das (fp32)", "das", BF_DAS_FUSED_F32, ()), ("mvdr", "mvdr", 0, ()),
             ("lcmv, 2 interferers", "lcmv", 0, (-60.0, 90.0)), ("gss, 2 interferers", "gss", 0, (-60.0, 90.0)), ("phase", "phase", 0, ()),
             ("phasempf", "phasempf", 0, ()), ("gsc", "gsc", 0, ())]

    def row(name, algo, impl, interf, hop=512, layout=BF_PLANAR, M=8, dirs=1, dump=False, prec=BF_PRECISION_REFERENCE, streams=1):
        try:
            k = kernels_of(algo, hop, layout, M, dirs, dump, impl, interf, prec, streams)
        except Exception as e:  # a shape the node refuses: say so in the table
            k = f"(refused: {str(e)[:60]})"
        rows.append((name, hop, "planar" if layout == BF_PLANAR else "[sample][mic]", M, dirs, "yes" if dump else "no", k))

    for name, algo, impl, interf in nodes:                       # the tuned shape and the other JACK periods
        for hop in (512, 64, 128, 256, 1024, 2048, 4096):
            row(name, algo, impl, interf, hop=hop)
    for name, algo, impl, interf in nodes:                       # period 512: the other layout, microphone counts, directions, dump
        if algo != "gsc":
            row(name, algo, impl, interf, layout=BF_INTERLEAVED)
        for M in (3, 16, 24):
            row(name, algo, impl, interf, M=M)
        if algo in ("das", "phase", "mvdr"):
            row(name, algo, impl, interf, dump=True)
        if algo == "das":
            row(name, algo, impl, interf, dirs=4)
            row(name, algo, impl, interf, dirs=8)
    row("mcra (one channel)", "mcra", 0, (), M=1)
    row("lcmv, 3 interferers", "lcmv", 0, (-60.0, 90.0, 150.0), M=16)
    # bf_config.precision = BF_PRECISION_MIXED (every row above: the default, BF_PRECISION_REFERENCE), and batches of many streams
    for name, algo, interf, M in (("mvdr, mixed precision", "mvdr", (), 8), ("lcmv, 2 interferers, mixed precision", "lcmv", (-60.0, 90.0), 8),
                                  ("lcmv, 3 interferers, mixed precision", "lcmv", (-60.0, 90.0, 150.0), 16), ("mvdr, mixed precision", "mvdr", (), 12),
                                  ("phase, mixed precision", "phase", (), 8), ("phasempf, mixed precision", "phasempf", (), 8)):
        row(name, algo, 0, interf, M=M, prec=BF_PRECISION_MIXED)
    row("phasempf, 64 streams", "phasempf", 0, (), streams=64)
    row("gss, 2 interferers, 64 streams", "gss", 0, (-60.0, 90.0), streams=64)
    lines = ["# Dispatch table (generated: `make dispatch-table` on a GPU box; tools/dispatch_table.py)", "",
             "One 96-frame batch per row, traced by `bf_trace_begin` / `bf_trace_end`: the kernels the library launched, in order (default `bf_config.precision` unless the row says otherwise).",
             "Names as rocprofv3 prints them, `bf::` dropped; `n1024::` etc. = the FFT size the kernel file was compiled for.", "",
             "| node | period | layout | mics | dirs | dump | kernels |", "|---|---|---|---|---|---|---|"]
    lines += [f"| {a} | {b} | {c} | {d} | {e} | {f} | `{g}` |" for a, b, c, d, e, f, g in rows]
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
