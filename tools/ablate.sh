#!/bin/bash
# ablation timing on the GPU box: kernel_ms under each VPT_DEBUG_ABLATE setting (results are wrong by design)
for D in "$@"; do
  echo "== ablate $D"; VPT_DEBUG_ABLATE=$D python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'])"
done
