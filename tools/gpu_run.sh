#!/bin/bash
O=gpurun_out/r03_b; mkdir -p $O
export TMPDIR=/tmp
python tools/ab_bench.py --variants new --ablate 64,128,192,16 --rounds 2 2>&1 | grep -v amdgpu.ids > $O/ab_tcp_model.jsonl
cat $O/ab_tcp_model.jsonl
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
( timeout 300 python bench.py --config 4 --sentences 100000 --steps 5 --warmup 2 2> $O/bench4.err | tail -1 ) > $O/bench4.json; head -c 600 $O/bench4.json; tail -3 $O/bench4.err
