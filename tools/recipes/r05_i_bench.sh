# round 5: the whole bench on the sources with the kernel diet, the flat writer and the double-array CPU baseline (one compact line + the detail file)
O=gpurun_out/r05_i; mkdir -p $O
python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err; wc -c $O/bench.json
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05_i/bench.json").read().strip().splitlines()[-1])
print(l["value"], l["roofline"]["frac"], l["parity"], {k: (round(v) if isinstance(v, float) else v) for k, v in l["cpu_baseline"].items() if k != "sample"})
for w in l["workloads"]: print(w)
PY
