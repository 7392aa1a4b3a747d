"""Timeline (start, end, queue) of the last ticks in a kernel trace made with scripts/chain_trace.py."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# last 2 ticks worth: find the last 4 k_pass_a launches
idx = [i for i, r in enumerate(rows) if "k_pass_a" in r[2]]
start = idx[-4]
t0 = rows[start][0]
for s, e, k, q in rows[start:]:
    print(f"{(s - t0) / 1000:9.1f} -> {(e - t0) / 1000:9.1f} us  ({(e - s) / 1000:7.1f})  q{q}  {k[:70]}")
