"""Average duration of every kernel over its last `last` launches in a rocprofv3 kernel trace (csv)."""
import csv, glob, sys, collections
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 50
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = []
for k, v in rows.items():
    v.sort()
    tail = v[-last:]
    out.append((sum(e - s for s, e in tail) / len(tail) / 1000.0, len(v), k))
tot = 0
for us, cnt, k in sorted(out, reverse=True):
    if cnt >= last:
        tot += us
        print(f"{us:9.1f} us  x{cnt:6d}  {k[:110]}")
print(f"{tot:9.1f} us  sum of the per-tick kernels")
