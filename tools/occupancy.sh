#!/bin/bash
# occupancy / footprint experiments on the GPU box
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms',d['roofline']['kernel_ms'],'hot_MB',d['config']['hot_table_bytes']/1e6)"; }
for PADB in 0 4000 9000 17000 30000 57000; do echo "== lds pad $PADB"; VPT_DEBUG_LDS_PAD=$PADB run; done
for SC in 0.5 0.25 0.1 0.03; do echo "== model scale $SC"; run --model-scale $SC; done
