#!/bin/bash
set -u
O=gpurun_out/c53; mkdir -p $O
timeout 260 python tools/fuzz_gpu.py 230 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
