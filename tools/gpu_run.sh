#!/bin/bash
set -u
O=gpurun_out/c33; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c33/bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"kernel_ms",d["roofline"]["kernel_ms"],"frac",d["roofline"]["frac"],"parity",d["parity"])
print("e2e",json.dumps(d.get("e2e")))
for w in d.get("workloads",[]): print(w["workload"][:50], w["value"], w["roofline"]["kernel_ms"], w["roofline"]["frac"], w.get("parity"), w.get("tags"))
PY
