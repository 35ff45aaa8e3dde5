#!/bin/bash
# kernel time vs batch size (tail / round-quantisation effects)
for S in 25000 50000 100000 200000 400000 1000000; do
  echo "== sentences $S"; python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sentences $S 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']; print('kernel_ms',k,'tiles',d['config']['tiles'],'G boundaries/s (kernel)', d['config']['boundaries_per_gpu']/k/1e6, 'value', d['value']/1e9)"
done
