#!/bin/bash
set -u
O=gpurun_out/c6; mkdir -p $O
export TMPDIR=/tmp
timeout 120 ./tools/pcie_bench > $O/pcie.txt 2>&1; cat $O/pcie.txt
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python tools/ab_bench.py --variants new,new:VPT_FAST_CAP=1536,wg8 --rounds 3 > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?"; cat $O/ab.jsonl
timeout 600 python tools/ab_bench.py --min-len 8 --max-len 512 --variants new,new:VPT_FAST_CAP=1280,wg7 --rounds 2 > $O/ab_ragged.jsonl 2> $O/ab_ragged.err; echo "ab ragged rc=$?"; cat $O/ab_ragged.jsonl
