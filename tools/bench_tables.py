#!/usr/bin/env python3
"""Rewrite the measured-numbers tables of BASELINE.md (section 4) and README.md from the bench records themselves, so that no figure in
either table is typed by hand:  python tools/bench_tables.py [--check]

  column "driver"   the newest BENCH_rNN.json at the repo root (written by the round driver on its own box: `parsed` = the headline fields of
                    bench.py's JSON line, `tail` = the last 2 000 characters of it, which is where bench.py puts `extra.ms` -- one figure
                    per secondary line -- for exactly this reason)
  column "builder"  min - max over the bench.py lines this round's profile runs left under profiles/ (rNN_*bench*.json, un-profiled ones)

The block between <!-- BENCH:BEGIN --> and <!-- BENCH:END --> is replaced.  --check (the CPU suite runs it) regenerates each block from the
records the block itself names -- its BENCH file and its profiles round -- and exits 1 if the text differs: no figure in a table can be
typed or edited by hand, and a table never silently mixes sources; a newer BENCH_rNN.json than the one a table names is reported."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- BENCH:BEGIN (tools/bench_tables.py: do not edit by hand) -->", "<!-- BENCH:END -->"

# (key in extra.ms, what it is)
ROWS = [
    ("das_f64_distinct_rows", "[1] the headline batch on a geometry WITHOUT coinciding microphones (4.0 transforms per frame instead of 3.5)"),
    ("mvdr", "[2] mvdr 8-mic, 65 536 frames, the library default (c128 spectra, fp64 backward transform)"),
    ("mvdr_mixed", "[2] same, `BF_PRECISION_MIXED` (z48 spectra, fp32 backward transform)"),
    ("phasempf", "[3] phasempf 8-mic, 256 streams x 256 frames"),
    ("lcmv16", "[4] lcmv 16-mic K=3, one GPU's shard of 32 768 frames, the library default"),
    ("lcmv16_mixed", "[4] same, `BF_PRECISION_MIXED`"),
    ("phase", "phase 8-mic, 65 536 frames (gates closed)"),
    ("phase_gate_open", "phase, every gate open"),
    ("das_f32", "[1] through the fused fp32 opt-in (`BF_DAS_FUSED_F32`)"),
    ("das_interleaved", "[1] `[sample][mic]` input, fp32 opt-in"),
    ("das_f64_interleaved", "[1] `[sample][mic]` input, double"),
    ("gss", "gss 8-mic K=2, 256 x 256 frames"),
    ("das_dirs16", "das fp32, 16 look directions of the headline batch"),
    ("das_period256", "das fp32 at JACK period 256"),
    ("das_period1024", "das fp32 at JACK period 1024"),
]


def driver_record(name=None):
    """(file name, record) of BENCH_rNN.json: the named one, else the newest"""
    files = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")), key=lambda f: int(re.search(r"BENCH_r(\d+)", f).group(1)))
    if name is not None:
        files = [f for f in files if os.path.basename(f) == name]
    if not files:
        return None, None
    f = files[-1]
    return os.path.basename(f), json.load(open(f))


def newest_profiles_round():
    rs = [int(m.group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_*bench*.json")) for m in [re.match(r"r(\d+)_", os.path.basename(f))] if m]
    return max(rs) if rs else 0


def extra_ms_from_tail(tail):
    """extra.ms = {...} sits at the very end of bench.py's line; older records: whatever per-line figures the tail still holds"""
    m = re.search(r'"ms": (\{[^{}]*\})\}\}\s*(?:$|\n)', tail or "")
    out = {}
    if m:
        try:
            out.update(json.loads(m.group(1)))
        except ValueError:
            pass
    for m in re.finditer(r'"(\w+)": \{"workload": "(?:[^"\\]|\\.)*", "ms_per_step": ([0-9.eE+-]+)', tail or ""):
        out.setdefault(m.group(1), float(m.group(2)))
    m = re.search(r'"mvdr_ms_per_step": ([0-9.eE+-]+)', tail or "")
    if m and "mvdr" not in out and "mvdr_strict" not in out:
        out["mvdr"] = float(m.group(1))
    return out


def builder_lines(round_no):
    """bench.py lines of this round's own runs of ONE tree (profiles/rNN_*bench*.json; the profiled ones run under rocprofv3 and are left
    out): the files of the newest tag (rNN_<letter>) and every other file whose line carries the same config.git_head (other boxes)"""
    found = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r{round_no:02d}_*bench*.json"))):
        if "profiled" in f:
            continue
        try:
            txt = [ln for ln in open(f).read().splitlines() if ln.startswith("{")]
            if txt:
                m = re.match(r"(r\d+_[a-z]+)_", os.path.basename(f))
                found.append((m.group(1) if m else "", json.loads(txt[-1])))
        except (ValueError, OSError):
            pass
    if not found:
        return []
    newest = max(t for t, _ in found)
    heads = {(d.get("config") or {}).get("git_head") for t, d in found if t == newest} - {None}
    return [d for t, d in found if t == newest or (d.get("config") or {}).get("git_head") in heads]


def rng(vals, fmt):
    vals = [v for v in vals if isinstance(v, (int, float))]
    if not vals:
        return "—"
    lo, hi = min(vals), max(vals)
    return fmt % lo if abs(hi - lo) < 1e-12 else f"{fmt % lo} – {fmt % hi}"


def build_block(bench_name=None, round_no=None):
    name, rec = driver_record(bench_name)
    if rec is None:
        return None
    if round_no is None:
        round_no = newest_profiles_round()
    p = rec.get("parsed") or {}
    rf = p.get("roofline") or {}
    dms = extra_ms_from_tail(rec.get("tail"))
    bl = builder_lines(round_no)
    drv_round = int(re.search(r"r(\d+)", name).group(1))
    own = lambda f: [f(b) for b in bl]  # noqa: E731
    g = lambda d, *ks: (g(d.get(ks[0], {}), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None  # noqa: E731
    out = [BEGIN, "",
           f"Driver's box = `{name}` (`head` {rec.get('head', '?')}, `{p.get('dtype', '?')}`, {p.get('steps', '?')} steps: the round driver's own run of `bench.py` "
           f"at the end of round {drv_round}); builder's boxes = the {len(bl)} un-profiled `bench.py` runs of round {round_no} "
           f"under `profiles/r{round_no:02d}_*` (range). A driver column older than the builder column is the previous round's tree.", "",
           f"| line | driver's box ({name[:-5]}) | builder's boxes (profiles r{round_no:02d}) |", "|---|---|---|"]
    fmt3 = "%.3f"
    out.append(f"| **[1] das 8-mic 1024-pt, 65 536 frames, double (headline)**: ms per step | **{p.get('ms_per_step', float('nan')):.4f}** | {rng(own(lambda b: b.get('ms_per_step')), '%.4f')} |")
    out.append(f"| … `{rf.get('kernel', 'kernel')}` ms per launch (HIP event pairs inside the timed steps) | {rf.get('kernel_ms', float('nan')):.4f} | {rng(own(lambda b: g(b, 'roofline', 'kernel_ms')), '%.4f')} |")
    out.append(f"| … `roofline.frac` of 8 TB/s (18 432 B per frame) | **{100 * rf.get('frac', float('nan')):.1f} %** | {rng([100 * v for v in own(lambda b: g(b, 'roofline', 'frac')) if v], '%.1f')} % |")
    tr, tb = rf.get("traffic"), [v for v in own(lambda b: g(b, "roofline", "traffic")) if v]
    algo_gb = 18432 * 65536 / 1e9
    tr_s = "null" if not tr else f"{tr / 1e9:.3f} GB = {tr / 1e9 / algo_gb:.2f} x"
    out.append(f"| … fabric traffic per launch (`profiles/traffic_das8_f64.json`) | {tr_s} | {rng([v / 1e9 for v in tb], fmt3)} GB |")
    out.append(f"| … frames/s | {p.get('value', float('nan')):.3e} | {rng(own(lambda b: b.get('value')), '%.3e')} |")
    cb = p.get("cpu_baseline") or {}
    out.append(f"| CPU baseline (`{cb.get('kind', '?')}`, {cb.get('cores', '?')} core) frames/s | {cb.get('value', float('nan')):.3e} | {rng(own(lambda b: g(b, 'cpu_baseline', 'value')), '%.3e')} |")
    for key, what in ROWS:
        dv = dms.get(key)
        if key == "mvdr" and dv is None:
            dv = dms.get("mvdr_strict")
        bv = own(lambda b, k=key: g(b, "extra", k, "ms_per_step"))
        if dv is None and not [v for v in bv if v]:
            continue
        out.append(f"| {what}: ms per step | {'—' if dv is None else fmt3 % dv} | {rng(bv, fmt3)} |")
    out += ["", END]
    return "\n".join(out)


def main():
    check = "--check" in sys.argv
    newest, _ = driver_record()
    if newest is None:
        print("no BENCH_r*.json")
        return 0
    bad = False
    for fn in ("BASELINE.md", "README.md"):
        path = os.path.join(ROOT, fn)
        s = open(path).read()
        if BEGIN not in s or END not in s:
            print(f"{fn}: no BENCH markers")
            bad = bad or check
            continue
        old = s[s.index(BEGIN):s.index(END) + len(END)]
        if check:  # regenerate from the sources the block names
            m = re.search(r"Driver's box = `(BENCH_r\d+\.json)`.*?bench\.py` runs of round (\d+)", old, re.S)
            if not m or build_block(m.group(1), int(m.group(2))) != old:
                print(f"{fn}: the bench table does not match its records: run python tools/bench_tables.py")
                bad = True
            elif m.group(1) != newest:
                print(f"{fn}: note: the table was written from {m.group(1)}; {newest} exists (python tools/bench_tables.py)")
            continue
        new = s[:s.index(BEGIN)] + build_block() + s[s.index(END) + len(END):]
        if new != s:
            open(path, "w").write(new)
            print(f"{fn}: table rewritten")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
