# round 4, seventh call: vpt_tokenize_batch -- the two-lane schedule (copies out behind the kernels on their stream) against the default (a
# copy-out stream), chunk sizes, without the dead memset per chunk; stress; a timeline of the lanes
O=gpurun_out/r04_g; mkdir -p $O
for E in "" "VPT_TOKENIZE_LANES=2" "VPT_TOKENIZE_LANES=2 VPT_TOKENIZE_CHUNK_BYTES=2000000" "VPT_TOKENIZE_LANES=2 VPT_TOKENIZE_CHUNK_BYTES=5000000" "VPT_TOKENIZE_LANES=2 VPT_TOKENIZE_CHUNK_BYTES=7000000" "VPT_TOKENIZE_CHUNK_BYTES=5000000"; do
  env $E python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
for E in "" "VPT_TOKENIZE_LANES=2" "VPT_TOKENIZE_LANES=2 VPT_TOKENIZE_CHUNK_BYTES=16000000"; do
  env $E python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
cat $O/tokenize.jsonl
VPT_TOKENIZE_LANES=2 python tools/tokenize_stress.py --iters 250 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500 > $O/stress_lanes.json; cat $O/stress_lanes.json
cd /tmp
VPT_TOKENIZE_LANES=2 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/tokenize_bench.py --iters 6 > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/r04_g/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob("gpurun_out/r04_g/trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))[:40]))
ev.sort()
tail = ev[-44:]
t0 = tail[0][0] if tail else 0
with open("gpurun_out/r04_g/timeline.txt", "w") as w:
    for a, b, n in tail:
        w.write("%9.1f us  +%8.1f us  %s\n" % ((a - t0) / 1e3, (b - a) / 1e3, n))
print(open("gpurun_out/r04_g/timeline.txt").read())
PY
