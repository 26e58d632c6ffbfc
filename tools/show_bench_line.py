#!/usr/bin/env python3
"""Pretty-print the interesting parts of a bench.py JSON line.  usage: show_bench_line.py <file> (or the line on stdin)"""
import json, sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("value %.2f %s  ms/step %.4f  n_gpus %d  regime %s  env %s" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d.get("regime", {}).get("name"), d["config"].get("env")))
print("roofline: kernel_ms %s frac %s achieved %s stage %s" % (r.get("kernel_ms"), r.get("frac"), r.get("achieved"), (r.get("stage") or {}).get("ms")))
for k in ("stage_us", "multi_gpu_workload_on_1_gpu", "pcie_inclusive", "cpu_baseline", "lockstep"):
    if k in d:
        print(k, json.dumps(d[k])[:400])
for k, v in (d.get("other_configs") or {}).items():
    print("other", k, v["value"], "frames/s", v.get("stage_us"), v.get("mesh_extraction"))
for k, v in (d.get("steady_state") or {}).items():
    print("steady", k, {x: v[x] for x in ("value", "kernel_ms", "reference_bytes_rate", "reference_bytes_rate_over_peak", "kernel_traffic_rate", "kernel_traffic_frac_of_hbm_peak")})
for blk in ("per_rank", "c5"):
    if blk in d:
        b = d[blk]
        if blk == "c5":
            print("c5:", b["value"], "frames/s", b.get("mesh_extraction"), b.get("lockstep"))
            b = b["per_rank"]
        for row in b["ranks"]:
            print("  ", blk, row)
