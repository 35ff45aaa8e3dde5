#!/bin/bash
# scratch GPU-box script of the current experiment (rewritten per call)
set -u
O=gpurun_out/c10; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python bench.py --config 4 --quick --steps 10 --warmup 2 > $O/bench_tags.json 2> $O/bench_tags.err; echo "bench tags rc=$?"; tail -2 $O/bench_tags.err; python -c "
import json;d=json.loads(open('$O/bench_tags.json').read().strip().splitlines()[-1]);print('tags',d.get('tags'));print('ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'],'parity',d['parity'])"
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --config 4 --quick --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --sentences 300000 > $OLDPWD/$O/trace.log 2>&1; cd $OLDPWD; cat $O/trace/*/*kernel_stats.csv | cut -c1-200
