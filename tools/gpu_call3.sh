#!/bin/bash
# GPU call 3 of round 2: the double-array kernel on hardware -- parity suite, A/B against round 1's kernel, ablations, profile.
set -u
O=gpurun_out/c3; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python tools/ab_bench.py --variants new,nostream,new:VPT_INLINE_ASSIGN=1,v1:VPT_SEPARATE_ASSIGN=1 --ablate 256,257,260,264,258,272 --rounds 2 > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?"; tail -3 $O/ab.err
cat $O/ab.jsonl
timeout 600 python tools/ab_bench.py --model-kind 2 --variants new,v1:VPT_SEPARATE_ASSIGN=1 --rounds 2 > $O/ab_m2.jsonl 2> $O/ab_m2.err; echo "ab m2 rc=$?"; cat $O/ab_m2.jsonl
timeout 600 python tools/ab_bench.py --model-kind 1 --min-len 8 --max-len 512 --variants new,v1:VPT_SEPARATE_ASSIGN=1 --rounds 2 > $O/ab_ragged.jsonl 2> $O/ab_ragged.err; echo "ab ragged rc=$?"; cat $O/ab_ragged.jsonl
VPT_PMC_GROUPS="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE|TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU|SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" timeout 900 bash tools/profile.sh r02_b > $O/profile.log 2>&1; echo "profile rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | cut -c1-1500
