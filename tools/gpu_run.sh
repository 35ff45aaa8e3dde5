#!/bin/bash
O=gpurun_out/r03_r; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/gpu_tests.log
emit() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
e=d.get('emit') or {}; t=d.get('tags') or {}
print(sys.argv[1], 'emit ms', e.get('ms_per_step'), 'parity', e.get('parity'), 'step ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'tags ms', t.get('ms_per_step'), t.get('parity'))
PY
}
timeout 900 python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --quick > $O/bench_c4.json 2> $O/bench_c4.err; emit $O/bench_c4.json
for k in 0 32 128 64 192; do
  VPT_DEBUG_EMIT=$k timeout 900 python bench.py --config 4 --steps 6 --warmup 2 --no-e2e --quick --no-cpu-baseline --sentences 300000 > $O/bench_c4_dbg$k.json 2> $O/bench_c4_dbg$k.err; emit $O/bench_c4_dbg$k.json
done
VPT_FUZZ_SEED0=16000 timeout 200 python tools/fuzz_gpu.py 60 2>&1 | tail -2 | tee $O/fuzz.log
