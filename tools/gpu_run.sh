#!/bin/bash
# scratch: the script of the latest gpurun call -- here a long fuzz over two fresh seed ranges on the round's final sources
O=gpurun_out/r03_zy; mkdir -p $O
export TMPDIR=/tmp
VPT_FUZZ_SEED0=60000 timeout 230 python tools/fuzz_gpu.py 200 2>&1 | tail -2 | tee $O/fuzz.log
VPT_TAG_SPLIT=1 VPT_TAG_QUEUE=8 VPT_FUZZ_SEED0=70000 timeout 100 python tools/fuzz_gpu.py 70 2>&1 | tail -2 | tee $O/fuzz_split_overflow.log
