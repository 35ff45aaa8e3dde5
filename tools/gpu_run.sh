#!/bin/bash
set -u
O=gpurun_out/c47; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
run() { n=$1; c=$2; shift; shift; env "$@" timeout 400 python bench.py --config $c --quick --steps 10 --warmup 3 --no-e2e --no-emit > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json;d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1]);print('$n','kernel_ms',round(d['roofline']['kernel_ms'],4),'frac',round(d['roofline']['frac'],3),'value',round(d['value']/1e9,2),'tiles',d['config']['tiles'],'parity',d['parity'])"; }
run c1_packed 1 X=1
run c1_old 1 VPT_NO_PACKED_TILES=1
run c4_packed 4 X=1
run c4_old 4 VPT_NO_PACKED_TILES=1
run c3_packed 3 X=1
run c3_old 3 VPT_NO_PACKED_TILES=1
timeout 120 python tools/fuzz_gpu.py 45 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
