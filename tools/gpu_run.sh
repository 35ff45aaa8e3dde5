#!/bin/bash
O=gpurun_out/r03_a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/gpu_tests.log
( timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 ) > $O/bench.json
tail -5 $O/gpu_tests.log; head -c 1500 $O/bench.json
