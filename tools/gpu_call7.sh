#!/bin/bash
set -u
O=gpurun_out/c7; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
for C in 524288 1048576 2097152; do
VPT_CHUNK_CHARS=$C timeout 300 python bench.py --quick --steps 20 --warmup 5 > $O/bench_$C.json 2> $O/bench.err; echo "bench rc=$? chunk=$C"; python -c "
import json;d=json.loads(open('$O/bench_$C.json').read().strip().splitlines()[-1]);print('e2e',d.get('e2e'));print('value',d['value'],'kernel_ms',d['roofline']['kernel_ms'],'frac',d['roofline']['frac'],'parity',d['parity'])"
done
