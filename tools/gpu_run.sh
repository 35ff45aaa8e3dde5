#!/bin/bash
O=gpurun_out/r03_j; mkdir -p $O
export TMPDIR=/tmp
python tools/ab_bench.py --variants new,r02,dense0,oldloop,nocheck,d0oldnc --rounds 3 2>&1 | grep -v amdgpu.ids > $O/ab_m1.jsonl; cut -c1-200 $O/ab_m1.jsonl
