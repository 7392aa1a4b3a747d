"""Average duration of every kernel over the last `last` ticks of a rocprofv3 kernel trace (csv): the window starts
with the `last`-th launch of the search kernel from the end."""
import csv, glob, sys, collections
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 50
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
search = sorted(v for k, vs in rows.items() if "k_pass_a" in k for v in vs)
t0 = search[-last][0]
out = []
for k, v in rows.items():
    tail = [(s, e) for s, e in v if s >= t0]
    if tail:
        out.append((sum(e - s for s, e in tail) / last / 1000.0, len(tail), k))
tot = 0
for us, cnt, k in sorted(out, reverse=True):
    tot += us
    print(f"{us:9.1f} us per tick  ({cnt:4d} launches)  {k[:100]}")
print(f"{tot:9.1f} us per tick  all kernels, last {last} ticks")
