#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh output directory: kernel stats tables + HBM traffic of the fusion kernel per launch.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (section HBM) FETCH_SIZE
on gfx950 counts a wide coalesced 16-B-per-lane read stream at exactly half its bytes, so the read side is doubled; the
write side is used as is.  The two counters come from separate passes (they do not fit in one)."""
import csv, glob, json, os, sys

out = sys.argv[1]
res = {}
lines = []
for cfg in ("c2", "c4", "c2_sat", "c4_sat", "c5"):
    vals = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(out, "pmc_%s_%s" % (kind, cfg), "*", "*counter_collection.csv"))
        if not files:
            continue
        # one row per (dispatch, XCD / instance): sum a dispatch's rows first, then average over the dispatches.  The saturated runs hold both
        # instantiations (the first max_weight frames run the plain kernel): only the SAT one counts there.
        per_dispatch = {}
        for r in csv.DictReader(open(files[0])):
            name = r["Kernel_Name"]
            if not ("k_integrate_pairs" in name or "k_integrate_bricks" in name) or r["Counter_Name"] != counter:
                continue
            if cfg.endswith("_sat") and ", true" not in name:
                continue
            key = r.get("Dispatch_Id") or r.get("Dispatch_ID") or len(per_dispatch)
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
        tot, n = sum(per_dispatch.values()), len(per_dispatch)
        if n:
            vals[kind] = tot / n * 1024.0
    if "fetch" in vals and "write" in vals:
        hbm = 2.0 * vals["fetch"] + vals["write"]
        res[cfg.upper().replace("_SAT", "_sat")] = int(hbm)
        lines.append("%s fusion kernel (k_integrate_pairs) per launch: FETCH_SIZE %.1f MB (x2 gfx950 correction -> %.1f MB), WRITE_SIZE %.1f MB, HBM traffic %.1f MB"
                     % (cfg.upper(), vals["fetch"] / 1e6, 2 * vals["fetch"] / 1e6, vals["write"] / 1e6, hbm / 1e6))
    st = glob.glob(os.path.join(out, "trace_%s" % cfg, "*", "*kernel_stats.csv"))
    if st:
        lines.append("%s kernel stats (rocprofv3 --kernel-trace --stats, see tools/profile_round.sh for the bench.py command):" % cfg.upper())
        for i, r in enumerate(csv.reader(open(st[0]))):
            if i < 16:
                lines.append("  " + ",".join(r))
json.dump(res, open(os.path.join(out, "integrate_traffic.json"), "w"))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
