#!/bin/bash
O=gpurun_out/r03_n; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/gpu_tests.log
emit() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
e=d.get('emit') or {}
print(sys.argv[1], 'emit ms', e.get('ms_per_step'), 'parity', e.get('parity'), 'step ms', d['ms_per_step'])
PY
}
for k in 0 4 8 12 24 32 64; do
  VPT_EMIT_PER_BLOCK=$k timeout 600 python bench.py --config 1 --steps 20 --warmup 5 --no-e2e --quick > $O/bench_c1_k$k.json 2> $O/bench_c1_k$k.err; emit $O/bench_c1_k$k.json
done
for k in 0 2 4 16 32; do
  VPT_EMIT_PER_BLOCK=$k timeout 900 python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --quick > $O/bench_c4_k$k.json 2> $O/bench_c4_k$k.err; emit $O/bench_c4_k$k.json
done
VPT_FUZZ_SEED0=0 timeout 200 python tools/fuzz_gpu.py 100 2>&1 | tail -3 | tee $O/fuzz.log
