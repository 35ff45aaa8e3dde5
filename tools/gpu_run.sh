#!/bin/bash
set -u
O=gpurun_out/c38; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for c in 1 4; do
timeout 600 python bench.py --config $c --quick --steps 10 --warmup 2 --no-e2e > $O/b$c.json 2> $O/b$c.err; echo "rc=$?"; tail -2 $O/b$c.err | cut -c1-300
python -c "
import json;d=json.loads(open('$O/b$c.json').read().strip().splitlines()[-1]);print('emit',d.get('emit'));print('tags',d.get('tags'));print('kernel_ms',d['roofline']['kernel_ms'],'parity',d['parity'])"
done
timeout 120 python tools/fuzz_gpu.py 60 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
