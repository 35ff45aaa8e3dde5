# round 5, second half of the evidence run (after profiles/traffic.json holds the entries of r05_z): the whole bench, the dry runs of the scaling jobs
O=gpurun_out/r05_zz2; mkdir -p $O
python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; wc -c $O/bench.json
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05_zz2/bench.json").read().strip().splitlines()[-1])
print(l["value"], l["roofline"]["frac"], l["roofline"]["traffic"], l["parity"])
for w in l["workloads"]: print({k: w[k] for k in ("name", "value_G", "kernel_ms", "frac", "traffic_ratio", "parity", "emit_ms") if k in w})
PY
python bench.py --dry-scale --steps 10 --warmup 2 --no-e2e --no-emit > $O/dry_scale.jsonl 2> $O/dry_scale.err; tail -1 $O/dry_scale.err
python bench.py --dry-scale --scale-sweep 1,4 --scale-config 4 --steps 5 --warmup 2 --no-e2e --no-emit > $O/dry_scale_c4.jsonl 2> $O/dry_scale_c4.err; tail -1 $O/dry_scale_c4.err
