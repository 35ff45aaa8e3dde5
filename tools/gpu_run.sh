#!/bin/bash
set -u
O=gpurun_out/c41; mkdir -p $O
for i in 1 2 3; do timeout 200 python tools/e2e_bench.py > $O/e_$i.json 2>/dev/null; cut -c1-150 $O/e_$i.json; done
timeout 600 python bench.py --config 1 --quick --steps 20 --warmup 5 --no-emit > $O/b_noemit.json 2>/dev/null
timeout 600 python bench.py --config 1 --quick --steps 20 --warmup 5 > $O/b_emit.json 2>/dev/null
python - <<'PY'
import json
for n in ("b_noemit","b_emit"):
    d=json.loads(open("gpurun_out/c41/%s.json"%n).read().strip().splitlines()[-1])
    e=d["e2e"]; print(n, round(e["ms_per_batch"],4), round(e["labels_only"]["ms_per_batch"],4), round(e["tokenize"]["ms_per_batch"],4), round(e["large_batch"]["ms_per_batch"],3))
PY
