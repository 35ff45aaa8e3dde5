#!/bin/bash
# The last GPU call of a round, most important first (every step writes its own file under gpurun_out/<tag>/, so
# whatever finished before the time limit survives):  ./tools/final_round.sh <tag>
TAG=${1:-final}; REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 150 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/gpu_tests.log
python bench.py --steps 30 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/bench.json
python tools/ab_bench.py --variants new,new_wg5,r01z 2>&1 | grep -v amdgpu.ids > $OUT/ab_m1.jsonl
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
VPT_PMC_GROUPS="FETCH_SIZE|WRITE_SIZE|SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" ./tools/profile.sh $TAG > $OUT/profile.log 2>&1
python tools/ab_bench.py --variants new --model-kind 2 2>&1 | grep -v amdgpu.ids > $OUT/ab_m2.jsonl
python tools/ab_bench.py --variants new --min-len 8 --max-len 512 2>&1 | grep -v amdgpu.ids > $OUT/ab_ragged.jsonl
