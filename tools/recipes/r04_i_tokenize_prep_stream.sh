# round 4: vpt_tokenize_batch with the char count and tile search on a stream (and workspace) of their own, the scoring launches back to back
# on theirs; phase D of the scoring kernel on the chunk scan kept from phase A.  Parity first, then A/B against the one-stream schedule
# (VPT_TOKENIZE_SERIAL=1), chunk sizes, stress, a timeline, configs[1]'s fused-writer kernel.
O=gpurun_out/r04_i; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "tokenize or predict_and_write or flat_kernel or pinned" 2>&1 | tail -3 | tee $O/tests.log
for E in "" "VPT_TOKENIZE_SERIAL=1" "VPT_TOKENIZE_CHUNK_BYTES=2400000" "VPT_TOKENIZE_CHUNK_BYTES=4800000" "VPT_TOKENIZE_CHUNK_BYTES=6400000" "VPT_TOKENIZE_CHUNK_BYTES=1600000"; do
  env $E python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
for E in "" "VPT_TOKENIZE_SERIAL=1" "VPT_TOKENIZE_CHUNK_BYTES=16000000" "VPT_TOKENIZE_CHUNK_BYTES=48000000"; do
  env $E python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
cat $O/tokenize.jsonl | cut -c1-400
python tools/tokenize_stress.py --iters 250 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500 > $O/stress.json; cat $O/stress.json
python bench.py --config 1 --steps 20 --warmup 5 --no-e2e 2>$O/bench_c1.err | tail -1 > $O/bench_c1.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_i/bench_c1.json").read())
print("configs[1] kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "fused", d["emit"]["fused"])
PY
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/tokenize_bench.py --iters 6 > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/r04_i/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60]))
for f in glob.glob("gpurun_out/r04_i/trace/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))[:40]))
ev.sort()
tail = ev[-48:]
t0 = tail[0][0] if tail else 0
with open("gpurun_out/r04_i/timeline.txt", "w") as w:
    for a, b, n in tail:
        w.write("%9.1f us  +%8.1f us  %s\n" % ((a - t0) / 1e3, (b - a) / 1e3, n))
print(open("gpurun_out/r04_i/timeline.txt").read())
PY
rm -rf $O/trace
