#!/usr/bin/env python3
"""Per-call kernel durations of the bench's workload from a rocprofv3 --kernel-trace run:

    python scripts/summarize_trace.py <dir with *_kernel_trace.csv> <out.json> [particles]

rocprofv3's --stats CSV averages every launch of a kernel name, the 4,096-particle launches of bench.py's primer
included (two of pass B's 25 calls in round 2: 20 us each against 70).  This summary keeps, per kernel, only the calls
with the workload's grid (the largest Grid_Size_X that kernel was launched with) and reports their count, mean, min, max
and standard deviation; `pair` is the force pair by these per-call means: what bench.py prints as roofline.rocprof."""
import csv
import glob
import json
import statistics
import sys

src, out = sys.argv[1], sys.argv[2]
particles = int(sys.argv[3]) if len(sys.argv) > 3 else 1048576
files = sorted(glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True))
if not files:
    raise SystemExit(f"no *kernel_trace.csv under {src}")
calls = {}
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        calls.setdefault(name, []).append((int(r["Grid_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0,
                                           int(r.get("VGPR_Count", 0) or 0), int(r.get("SGPR_Count", 0) or 0), int(r.get("LDS_Block_Size", 0) or 0),
                                           int(r.get("Workgroup_Size_X", 0) or 0)))
res = {"particles": particles, "source": "rocprofv3 --kernel-trace of bench.py (--no-kernel-events --repeats 1 --clock-warmup 0); per kernel "
       "only the calls with the workload's grid (its largest Grid_Size_X): the primer's 4,096-particle launches are left out",
       "kernels": {}}
for name, rows in sorted(calls.items()):
    g = max(r[0] for r in rows)
    us = [r[1] for r in rows if r[0] == g]
    if len(us) < 3 or not name.startswith("sc::"):
        continue
    k = next(r for r in rows if r[0] == g)
    res["kernels"][name] = {"calls": len(us), "grid_size_x": g, "workgroup_size_x": k[5], "mean_us": round(statistics.mean(us), 3),
                            "min_us": round(min(us), 3), "max_us": round(max(us), 3), "stdev_us": round(statistics.pstdev(us), 3),
                            "other_calls_left_out": len(rows) - len(us), "vgpr": k[2], "sgpr": k[3], "lds_bytes": k[4]}


def steady(tag):
    cands = [(v["calls"], n) for n, v in res["kernels"].items() if tag in n]
    return max(cands)[1] if cands else None


a, b = steady("k_pass_a<"), steady("k_pass_b<")
if a and b:
    us = res["kernels"][a]["mean_us"] + res["kernels"][b]["mean_us"]
    gbps = 128 * particles / (us * 1e-6) / 1e9
    res["pair"] = {"pass_a": a, "pass_b": b, "pass_a_us": res["kernels"][a]["mean_us"], "pass_b_us": res["kernels"][b]["mean_us"],
                   "avg_launch_us": round(us, 3), "achieved_GBps": round(gbps, 1), "frac": round(gbps / 8000.0, 5)}
json.dump(res, open(out, "w"), indent=1)
for n, v in res["kernels"].items():
    print(f"{n[:70]:70s} calls {v['calls']:3d} (+{v['other_calls_left_out']} other)  mean {v['mean_us']:8.2f} us  min {v['min_us']:8.2f}  max {v['max_us']:8.2f}")
print("pair:", res.get("pair"))
