# round 4: fill_tags' front-end launch flat over runs of sentences against the wave-per-sentence one (VPT_TAG_FRONT_BY_SENTENCE=1): parity on the
# tag tests, then configs[4] (1 M ragged sentences, 70 M tokens): step, fill_tags stand-alone, kernel trace; occupancy and run-length variants
O=gpurun_out/r04_o; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "tag" 2>&1 | tail -3 | tee $O/tests.log
for E in "" "VPT_TAG_FRONT_BY_SENTENCE=1" "VPT_TAG_FLAT_OCC=6" "VPT_TAG_FLAT_OCC=8" "VPT_TAG_FLAT_RUN=2048" "VPT_TAG_FLAT_RUN=16384"; do
  env $E python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --no-emit --no-cpu-baseline 2>/dev/null | tail -1 > $O/b.json
  python - "$E" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_o/b.json").read())
line = {"env": sys.argv[1], "step_ms": round(d["ms_per_step"], 4), "kernel_ms": round(d["roofline"]["kernel_ms"], 4), "tags_alone_ms": round(d["tags"]["ms_per_step"], 4), "G_per_s": round(d["value"] / 1e9, 2), "tags_parity": d["tags"].get("parity")}
print(json.dumps(line))
open("gpurun_out/r04_o/tag_front.jsonl", "a").write(json.dumps(line) + "\n")
PY
done
cd /tmp
for E in "X=1" "VPT_TAG_FRONT_BY_SENTENCE=1"; do
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_$E -- python $OLDPWD/bench.py --config 4 --steps 5 --warmup 2 --no-e2e --no-emit --no-cpu-baseline > /dev/null 2>&1
  python - $OLDPWD/$O/trace_$E <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:70].ljust(70), r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
done 2>&1 | tee $OLDPWD/$O/kernel_stats.txt
cd $OLDPWD; rm -rf $O/trace_*
