# round 4, fifth call: vpt_tokenize_batch after the flat char count, the look-back fix and the three-stage pipeline -- stress loops over the
# chunked paths, end-to-end timings (fused / separate kernels, copy-out / direct, chunk sizes), the bench line of configs[1], a timeline
O=gpurun_out/r04_e; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -x -q -k "tokenize or count_boundaries or predict_and_write or chars_left" 2>&1 | tail -5 ) > $O/gpu_tests.log
for E in "" "VPT_TOKENIZE_CHUNK_BYTES=1000000" "VPT_TOKENIZE_DIRECT=1" "VPT_TOKENIZE_SEPARATE=1 VPT_TOKENIZE_CHUNK_BYTES=3000000"; do
  env $E python tools/tokenize_stress.py --iters 250 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600 >> $O/stress.jsonl
done
cat $O/stress.jsonl
for E in "" "VPT_TOKENIZE_CHUNK_BYTES=1000000" "VPT_TOKENIZE_CHUNK_BYTES=2000000" "VPT_TOKENIZE_CHUNK_BYTES=6000000" "VPT_TOKENIZE_CHUNK_BYTES=1000000000" \
         "VPT_TOKENIZE_DIRECT=1" "VPT_TOKENIZE_DIRECT=1 VPT_TOKENIZE_CHUNK_BYTES=1000000000" \
         "VPT_TOKENIZE_SEPARATE=1" "VPT_TOKENIZE_SEPARATE=1 VPT_TOKENIZE_CHUNK_BYTES=4000000"; do
  env $E python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
for E in "" "VPT_TOKENIZE_CHUNK_BYTES=8000000" "VPT_TOKENIZE_DIRECT=1" "VPT_TOKENIZE_SEPARATE=1"; do
  env $E python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
python tools/tokenize_bench.py --config 4 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
cat $O/tokenize.jsonl
python bench.py --config 1 --steps 30 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; tail -3 $O/bench_c1.err
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/tokenize_bench.py --iters 6 > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
cat $O/trace/*/*kernel_stats.csv | cut -c1-170 | head -10
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/r04_e/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob("gpurun_out/r04_e/trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))[:40]))
ev.sort()
tail = ev[-52:]
t0 = tail[0][0] if tail else 0
with open("gpurun_out/r04_e/timeline.txt", "w") as w:
    for a, b, n in tail:
        w.write("%9.1f us  +%8.1f us  %s\n" % ((a - t0) / 1e3, (b - a) / 1e3, n))
print(open("gpurun_out/r04_e/timeline.txt").read())
PY
