#!/bin/bash
# the round's final GPU call: parity suite, the bench line of every workload, smoke, rocprofv3 trace + PMC passes (traffic entries), fuzz
O=gpurun_out/r03_x; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/gpu_tests.log; cat $O/gpu_tests.log
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_x/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'parity', d['parity'], 'cpu', d['cpu_baseline'])
for w in d.get('workloads', []):
    print(' ', w['workload'][:50], 'value %.3g' % w['value'], 'ms %.4f' % w['ms_per_step'], 'kernel %.4f' % w['roofline']['kernel_ms'], 'frac %.3f' % w['roofline']['frac'], w.get('parity'), w.get('tile_plan'), 'emit', (w.get('emit') or {}).get('ms_per_step'), 'tags', (w.get('tags') or {}).get('ms_per_step'))
print('emit', d.get('emit')); print('e2e', json.dumps(d.get('e2e'))[:1200])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
./tools/profile.sh r03_x --config 1 > $O/profile_c1.log 2>&1; tail -2 $O/profile_c1.log | cut -c1-400
VPT_PMC_GROUPS="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE" ./tools/profile.sh r03_x_c2 > $O/profile_c2.log 2>&1; grep "traffic entry" $O/profile_c2.log | cut -c1-400
VPT_PMC_GROUPS="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE" ./tools/profile.sh r03_x_c3 --config 3 > $O/profile_c3.log 2>&1; grep "traffic entry" $O/profile_c3.log | cut -c1-400
VPT_FUZZ_SEED0=24000 timeout 150 python tools/fuzz_gpu.py 90 2>&1 | tail -2 | tee $O/fuzz.log
