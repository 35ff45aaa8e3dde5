# round 6, after the evidence pass: the writer changed (kernels_emit.hip: outside the hash of the scoring sources, profiles/traffic.json stays valid) -- the whole GPU suite,
# smoke, a short fuzz and the bench line on the tree as it is committed
O=gpurun_out/r06_zz; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/smoke.log
( timeout 300 python tools/fuzz_gpu.py 150 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_gpu.log; cat $O/fuzz_gpu.log
SECONDS=0; python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench.py wall clock: $SECONDS s" | tee $O/bench_wall.txt; tail -c 300 $O/bench.err
python - <<PY
import json
l = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(len(json.dumps(l)), l["value"], l["roofline"]["frac"], l["roofline"]["traffic"], l["parity"])
for w in l["workloads"]: print({k: w[k] for k in ("name", "value_G", "kernel_ms", "frac", "traffic_ratio", "parity", "tags_ms", "tags_dense_ms", "emit_ms", "emit_frac", "tokenize_ms") if k in w})
PY
