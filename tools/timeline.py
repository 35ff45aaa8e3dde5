"""A launch-by-launch timeline out of a rocprofv3 --kernel-trace --memory-copy-trace run (csv):
python tools/timeline.py <trace dir> <out.txt> [last N events, default 48]"""
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 48
ev = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))[:40]))
ev.sort()
tail = ev[-n:]
t0 = tail[0][0] if tail else 0
with open(out, "w") as w:
    for a, b, name in tail:
        w.write("%9.1f us  +%8.1f us  %s\n" % ((a - t0) / 1e3, (b - a) / 1e3, name))
print(open(out).read())
