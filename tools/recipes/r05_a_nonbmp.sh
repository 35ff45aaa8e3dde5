# round 5, first call: the alphabet condition is off the packed tables (chars outside the BMP get ids through `xcid`): the parity suite on the
# GPU, a short fuzz whose alphabets hold non-BMP chars, and the bench with the new `nonbmp` workload beside configs[1] (one compact line + detail file)
O=gpurun_out/r05_a; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
( timeout 150 python tools/fuzz_gpu.py 100 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_gpu.log; cat $O/fuzz_gpu.log
python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; wc -c $O/bench.json
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05_a/bench.json").read().strip().splitlines()[-1])
print(l["value"], l["roofline"]["frac"], l["parity"])
for w in l["workloads"]: print(w)
PY
