#!/bin/bash
O=gpurun_out/r03_u; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_tests.log
emit() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
e=d.get('emit') or {}; t=d.get('tags') or {}
print(sys.argv[1], 'emit ms', e.get('ms_per_step'), 'step ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'frac', d['roofline'].get('frac'), 'tags ms', t.get('ms_per_step'), d['config'].get('tile_plan'), d.get('phase_share'))
PY
}
timeout 900 python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --quick > $O/bench_c4.json 2> $O/bench_c4.err; emit $O/bench_c4.json
VPT_FORCE_CUT_TILES=1 timeout 900 python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --quick > $O/bench_c4_cut.json 2> $O/bench_c4_cut.err; emit $O/bench_c4_cut.json
VPT_FORCE_CUT_TILES=1 timeout 900 python bench.py --config 1 --steps 20 --warmup 5 --no-e2e --quick > $O/bench_c1_cut.json 2> $O/bench_c1_cut.err; emit $O/bench_c1_cut.json
timeout 900 python bench.py --config 1 --steps 10 --warmup 3 --no-e2e --quick --phases --no-cpu-baseline > $O/bench_c1_phases.json 2> $O/bench_c1_phases.err; emit $O/bench_c1_phases.json
timeout 900 python bench.py --config 4 --steps 6 --warmup 2 --no-e2e --quick --phases --no-cpu-baseline > $O/bench_c4_phases.json 2> $O/bench_c4_phases.err; emit $O/bench_c4_phases.json
