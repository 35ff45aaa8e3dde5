# round 4, after the flat tag front end (kernels_tags.hip is not among the hashed scoring sources): parity suite, fuzz with fill_tags forced through the
# two-launch path (every model with tags goes through tag_front_flat_kernel), the bench line of every workload (traffic from profiles/traffic.json)
O=gpurun_out/r04_p; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
( VPT_TAG_SPLIT=1 VPT_FUZZ_SEED0=70000 timeout 200 python tools/fuzz_gpu.py 150 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_gpu_tag_split.log; cat $O/fuzz_gpu_tag_split.log
python bench.py --steps 20 --warmup 3 > $O/bench_all.json 2> $O/bench_all.err; tail -1 $O/bench_all.err | cut -c1-200
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_p/bench_all.json") if l.startswith("{")][-1])
print("primary", d["value"] / 1e9, d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
for w in d["workloads"]:
    print(w["workload"][:12], round(w["value"] / 1e9, 2), round(w["ms_per_step"], 4), round(w["roofline"]["frac"], 3), w["roofline"]["traffic"], {k: round(v, 3) for k, v in (w.get("tags") or {}).items() if k == "ms_per_step"})
PY
