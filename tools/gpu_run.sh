#!/bin/bash
set -u
O=gpurun_out/c50; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "tokeniz or tagged or writer or emit" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
timeout 600 python bench.py --config 4 --quick --steps 10 --warmup 2 --no-e2e --no-cpu-baseline > $O/b4.json 2> $O/b4.err; echo "rc=$?"
python -c "
import json;d=json.loads(open('$O/b4.json').read().strip().splitlines()[-1]);print('emit',d.get('emit')['ms_per_step'], 'tags', d['tags']['ms_per_step'])"
