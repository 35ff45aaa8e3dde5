#!/bin/bash
set -u
O=gpurun_out/c52; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c52/bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"kernel_ms",d["roofline"]["kernel_ms"],"frac",d["roofline"]["frac"],"traffic",d["roofline"]["traffic"],"parity",d["parity"])
e=d["e2e"]; print("e2e",round(e["ms_per_batch"],4),round(e["frac_of_pcie"],3),"labels",round(e["labels_only"]["ms_per_batch"],4),"tok",round(e["tokenize"]["ms_per_batch"],4),"big",round(e["large_batch"]["ms_per_batch"],3), e["parity"])
print("emit",d.get("emit")["ms_per_step"])
for w in d.get("workloads",[]): print(w["workload"][:30], round(w["value"]/1e9,2), round(w["roofline"]["kernel_ms"],4), round(w["roofline"]["frac"],3), w["roofline"]["traffic"], w.get("parity"), w.get("tags",{}) and round(w["tags"]["ms_per_step"],3), w.get("emit") and round(w["emit"]["ms_per_step"],3))
PY
