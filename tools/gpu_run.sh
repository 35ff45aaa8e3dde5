#!/bin/bash
O=gpurun_out/r03_k; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
for C in 1 4 6 7; do
( timeout 600 python bench.py --config $C --steps 20 --warmup 5 2>> $O/bench.err | tail -1 ) > $O/bench_c$C.json
python - <<PY
import json
w=json.load(open("$O/bench_c$C.json"))
print(w["config"]["workload"], "| value %.4g"%w["value"], "ms %.4f"%w["ms_per_step"], "kernel_ms %.4f"%w["roofline"]["kernel_ms"], "frac %.3f"%w["roofline"]["frac"], "parity", w["parity"], w["config"]["tile_plan"])
for k in ("tags","emit"):
    if k in w: print("    ", k, {a:b for a,b in w[k].items() if a in ("ms_per_step","parity","frac_of_hbm","tokens_checked","bytes_checked")})
if "e2e" in w: print("     e2e", w["e2e"]["frac_of_pcie"], "tokenize ms", w["e2e"]["tokenize"]["ms_per_batch"], w["e2e"]["tokenize"]["chars_per_s"])
PY
done
tail -3 $O/bench.err
