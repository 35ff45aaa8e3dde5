#!/bin/bash
O=gpurun_out/c54; mkdir -p $O
VPT_FUZZ_SEED0=20000 timeout 110 python tools/fuzz_gpu.py 95 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
