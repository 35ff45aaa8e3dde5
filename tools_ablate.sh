#!/bin/bash
# ablation timing on the GPU box: kernel_ms under each VPT_DEBUG_ABLATE setting
for D in 0 1 2 4 8 9; do
  echo "== ablate $D"; VPT_DEBUG_ABLATE=$D python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'])"
done
