#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh output directory: kernel stats tables + HBM-side traffic per launch of the fusion kernel (plain, DEFER and
past-saturation runs), of the raycast launch and of one marching-cubes extraction.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (section HBM) FETCH_SIZE
on gfx950 counts a wide coalesced 16-B-per-lane read stream at exactly half its bytes, so the read side is doubled; the
write side is used as is.  The two counters come from separate passes (they do not fit in one).  Infinity-Cache hits are counted:
at 512^3 the touched set lives in the cache, so the C2 figures are fabric traffic, not DRAM traffic."""
import csv, glob, json, os, sys

out = sys.argv[1]
res = {}
lines = []


def per_launch(cfg, counter, kind, want, launches_per_unit=None, tail=None):
    """median counter value (bytes) per dispatch of the kernels `want(name)` selects (tail: over the last `tail` of those dispatches only); with
    launches_per_unit: summed over all their dispatches / units"""
    files = glob.glob(os.path.join(out, "pmc_%s_%s" % (kind, cfg), "*", "*counter_collection.csv"))
    if not files:
        return None
    per_dispatch = {}
    for r in csv.DictReader(open(files[0])):       # one row per (dispatch, XCD / instance): sum a dispatch's rows first
        name = r["Kernel_Name"]
        if r["Counter_Name"] != counter or not want(name):
            continue
        key = int(r.get("Dispatch_Id") or r.get("Dispatch_ID") or len(per_dispatch))
        per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
    if not per_dispatch:
        return None
    if tail:
        per_dispatch = {k: per_dispatch[k] for k in sorted(per_dispatch)[-tail:]}
    if launches_per_unit:
        return sum(per_dispatch.values()) * 1024.0 / launches_per_unit
    vals = sorted(per_dispatch.values())           # the MEDIAN launch: a run's first fusion launch (empty volume) and the launches right after bench.py's
    return vals[len(vals) // 2] * 1024.0           # plain-kernel leg (deferred states being re-established) move several times the steady amount


def hbm(cfg, want, units=None, tail=None):
    f, w = per_launch(cfg, "FETCH_SIZE", "fetch", want, units, tail), per_launch(cfg, "WRITE_SIZE", "write", want, units, tail)
    if f is None or w is None:
        return None
    return f, w, 2.0 * f + w


# every bench.py run launches BOTH forms of the fusion kernel: the DEFER form (k_integrate_pairs<BR, true, false>) on the stream's frames and the
# plain read-modify-write form (<BR, false, false>) on the frames of the `roofline` leg that follows the timed region (kf_set_defer(0))
import re
def _form(n):                       # k_integrate_pairs<BR, DEFER, COLOR, LAYERS>: the second template argument
    m = re.search(r"k_integrate_pairs<\d+, (true|false)", n) or re.search(r"k_integrate_pairs_pipe<(true|false)", n)    # (the pipelined one-brick form: <DEFER>)
    return m.group(1) if m else None
plain = lambda n: _form(n) == "false"
defer = lambda n: _form(n) == "true"
for c in ("c2", "c4", "c5"):
    for run_name, forms in ((c, (("plain", plain, c.upper(), None), ("DEFER", defer, c.upper() + "_deferred", None))),
                            (c + "_saturated", (("DEFER past saturation: last 40 launches", defer, c.upper() + "_saturated", 40),))):
        for label, want, key, tail in forms:
            h = hbm(run_name, want, tail=tail)
            if h:
                res[key] = int(h[2])
                lines.append("%s fusion kernel (k_integrate_pairs, %s) per launch: FETCH_SIZE %.1f MB (x2 gfx950 correction -> %.1f MB), WRITE_SIZE %.1f MB, HBM-side traffic %.1f MB"
                             % (c.upper(), label, h[0] / 1e6, 2 * h[0] / 1e6, h[1] / 1e6, h[2] / 1e6))
        if run_name == c:
            h = hbm(run_name, lambda n: "k_raycast" in n)
            if h:
                res[c.upper() + "_raycast"] = int(h[2])
                lines.append("%s raycast launch (k_raycast_prefetch incl. the next frame's riders) per launch: 2 x FETCH %.1f MB + WRITE %.1f MB = %.1f MB" % (c.upper(), 2 * h[0] / 1e6, h[1] / 1e6, h[2] / 1e6))
            h = hbm(run_name, lambda n: "k_integrate_cull<true>" in n or "k_integrate_cull_sift<true>" in n)
            if h:
                res[c.upper() + "_cull"] = int(h[2])
                lines.append("%s cull (k_integrate_cull<DEFER> / k_integrate_cull_sift<DEFER>) per launch: 2 x FETCH %.2f MB + WRITE %.2f MB = %.2f MB" % (c.upper(), 2 * h[0] / 1e6, h[1] / 1e6, h[2] / 1e6))
        st = glob.glob(os.path.join(out, "trace_%s" % run_name, "*", "*kernel_stats.csv"))
        if st:
            lines.append("%s kernel stats (rocprofv3 --kernel-trace --stats, see tools/profile_round.sh for the bench.py command):" % run_name.upper())
            for i, r in enumerate(csv.reader(open(st[0]))):
                if i < 12:
                    lines.append("  " + ",".join(r))
h = hbm("mc_c2", lambda n: "k_mc_" in n, units=3)            # tools/bench_mcubes.py c2 3: three extractions
if h:
    res["C2_marching_cubes"] = int(h[2])
    lines.append("C2 marching cubes (all k_mc_* kernels of one extraction): 2 x FETCH %.1f MB + WRITE %.1f MB = %.1f MB" % (2 * h[0] / 1e6, h[1] / 1e6, h[2] / 1e6))
res["source"] = ("profiles/%s_summary.txt: builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh), 2 x FETCH + WRITE per launch, median over the "
                 "run's launches of that kernel; not measured in the bench run itself" % os.path.basename(out.rstrip("/")))
json.dump(res, open(os.path.join(out, "integrate_traffic.json"), "w"))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
