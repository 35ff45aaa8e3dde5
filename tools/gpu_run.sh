#!/bin/bash
set -u
O=gpurun_out/c25; mkdir -p $O
export TMPDIR=/tmp
run() { # name env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --config 4 --quick --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --sentences 300000 > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json;d=json.loads(open('$O/b_$n.json').read().strip().splitlines()[-1]);print('$n','tags_ms',round(d['tags']['ms_per_step'],4),'kernel_ms',round(d['roofline']['kernel_ms'],4))"
}
timeout 300 python -m pytest tests -m gpu -x -q -k "tag or emit or token" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
run base X=1
run wgs32 VPT_TAG_WGS_PER_CU=32
for d in 1 2; do run dbg$d VPT_DEBUG_TAGS=$d; done
