#!/bin/bash
# GPU call 5 of round 2: occupancy / speculative-W variants, the lane-per-token tag kernel, the two-engine host pipeline.
set -u
O=gpurun_out/c5; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python tools/ab_bench.py --variants new,wspec,wg7,wg8 --rounds 3 > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?"; cat $O/ab.jsonl
timeout 600 python tools/ab_bench.py --model-kind 2 --variants new,wspec,wg7,wg8 --rounds 2 > $O/ab_m2.jsonl 2> $O/ab_m2.err; echo "ab m2 rc=$?"; cat $O/ab_m2.jsonl
timeout 600 python tools/ab_bench.py --min-len 8 --max-len 512 --variants new,wspec,wg7,wg8 --rounds 2 > $O/ab_ragged.jsonl 2> $O/ab_ragged.err; echo "ab ragged rc=$?"; cat $O/ab_ragged.jsonl
timeout 600 python bench.py --config 4 --quick --sentences 300000 --steps 10 --warmup 2 > $O/bench_tags.json 2> $O/bench_tags.err; echo "bench tags rc=$?"; python -c "
import json;d=json.loads(open('$O/bench_tags.json').read().strip().splitlines()[-1]);print('tags',d.get('tags'));print('ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'],'parity',d['parity'])"
timeout 300 python bench.py --quick --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('e2e',d.get('e2e'));print('value',d['value'],'kernel_ms',d['roofline']['kernel_ms'],'frac',d['roofline']['frac'],'parity',d['parity'])"
