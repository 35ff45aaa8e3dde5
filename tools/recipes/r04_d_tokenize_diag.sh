# round 4, fourth call: why vpt_tokenize_batch is slow and (once) wrong in chunks -- the bench line of configs[1] with its stderr, a stress
# loop over the chunked paths, and kernel + copy timelines of the tokenize pipeline (no counters: traces only)
O=gpurun_out/r04_d; mkdir -p $O
python bench.py --config 1 --steps 30 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; tail -5 $O/bench_c1.err
VPT_TOKENIZE_CHUNK_BYTES=4000000 python tools/tokenize_stress.py --iters 300 2>&1 | grep -v amdgpu.ids | tail -1 > $O/stress_direct_4m.json
VPT_TOKENIZE_CHUNK_BYTES=4000000 VPT_TOKENIZE_NO_DIRECT=1 python tools/tokenize_stress.py --iters 300 2>&1 | grep -v amdgpu.ids | tail -1 > $O/stress_copy_4m.json
python tools/tokenize_stress.py --iters 300 2>&1 | grep -v amdgpu.ids | tail -1 > $O/stress_default.json
VPT_TOKENIZE_CHUNK_BYTES=1000000000 python tools/tokenize_stress.py --iters 200 2>&1 | grep -v amdgpu.ids | tail -1 > $O/stress_one_chunk.json
cat $O/stress_*.json
cd /tmp
for V in default one; do
  if [ $V = one ]; then export VPT_TOKENIZE_CHUNK_BYTES=1000000000; fi
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OLDPWD/$O/trace_$V -- python $OLDPWD/tools/tokenize_bench.py --iters 6 > $OLDPWD/$O/trace_$V.log 2>&1
done
cd $OLDPWD
for V in default one; do
  echo "== $V"; cat $O/trace_$V/*/*kernel_stats.csv | cut -c1-180 | head -14; cat $O/trace_$V/*/*memory_copy_stats.csv 2>/dev/null | head -6
done
python - <<'PY'
import csv, glob
for V in ("default", "one"):
    ev = []
    for f in glob.glob("gpurun_out/r04_d/trace_%s/*/*kernel_trace.csv" % V):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:50]))
    for f in glob.glob("gpurun_out/r04_d/trace_%s/*/*memory_copy_trace.csv" % V):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?"))[:40]))
    ev.sort()
    tail = ev[-60:] if V == "default" else ev[-16:]
    t0 = tail[0][0] if tail else 0
    with open("gpurun_out/r04_d/timeline_%s.txt" % V, "w") as w:
        for a, b, n in tail:
            w.write("%9.1f us  +%8.1f us  %s\n" % ((a - t0) / 1e3, (b - a) / 1e3, n))
    print(open("gpurun_out/r04_d/timeline_%s.txt" % V).read())
PY
