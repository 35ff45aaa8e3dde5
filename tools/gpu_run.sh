#!/bin/bash
O=gpurun_out/r03_p; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/gpu_tests.log
emit() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
e=d.get('emit') or {}; t=d.get('tags') or {}
print(sys.argv[1], 'emit ms', e.get('ms_per_step'), 'parity', e.get('parity'), 'step ms', d['ms_per_step'], 'tags ms', t.get('ms_per_step'), t.get('parity'))
PY
}
for k in 0 1 2 3 4 8 12 16 32; do
  VPT_DEBUG_EMIT=$k timeout 600 python bench.py --config 1 --steps 20 --warmup 5 --no-e2e --quick --no-cpu-baseline > $O/bench_c1_dbg$k.json 2> $O/bench_c1_dbg$k.err; emit $O/bench_c1_dbg$k.json
done
timeout 900 python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --quick > $O/bench_c4.json 2> $O/bench_c4.err; emit $O/bench_c4.json
VPT_PMC_GROUPS="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU|SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" ./tools/profile.sh r03_p_emit --config 1 > $O/profile.log 2>&1; grep -i "emit" gpurun_out/prof_r03_p_emit/summary.txt | cut -c1-160
VPT_FUZZ_SEED0=12000 timeout 200 python tools/fuzz_gpu.py 60 2>&1 | tail -2 | tee $O/fuzz.log
