# round 4, last: the tag tests and the bench line of every workload on the final tree (scoring sources as profiled in r04_zz: traffic from profiles/traffic.json)
O=gpurun_out/r04_q; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "tag" 2>&1 | tail -2 | tee $O/tag_tests.log
python bench.py --steps 20 --warmup 3 > $O/bench_all.json 2> $O/bench_all.err; tail -1 $O/bench_all.err | cut -c1-200
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_q/bench_all.json") if l.startswith("{")][-1])
print("primary", d["value"] / 1e9, d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
for w in d["workloads"]:
    print(w["workload"][:12], round(w["value"] / 1e9, 2), round(w["ms_per_step"], 4), round(w["roofline"]["frac"], 3), w["roofline"]["traffic"], {k: round(v, 3) for k, v in (w.get("tags") or {}).items() if k == "ms_per_step"})
PY
