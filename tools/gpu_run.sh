#!/bin/bash
O=gpurun_out/r03_z; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) | tee $O/gpu_tests.log
timeout 300 ./tools/gather_bench 2>&1 | tail -12 | tee $O/gather_microbench.txt
