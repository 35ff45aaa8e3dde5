#!/bin/bash
# evidence for the scaling path on a 1-GPU box + the bench line with the measured traffic of the final sources
O=gpurun_out/r03_y; mkdir -p $O
export TMPDIR=/tmp
python bench.py --gpus 2 --steps 5 --warmup 2 > $O/gpus2_on_one_gpu.out 2> $O/gpus2_on_one_gpu.err; echo "exit code $?" >> $O/gpus2_on_one_gpu.out; tail -3 $O/gpus2_on_one_gpu.err | cut -c1-300; cat $O/gpus2_on_one_gpu.out | cut -c1-200
VPT_BENCH_ONE_DEVICE=1 VPT_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --no-e2e 2> $O/bench_2ranks.err | tail -1 > $O/bench_2ranks_one_gpu_gloo.json; python -c "
import json; d=json.load(open('$O/bench_2ranks_one_gpu_gloo.json')); print('2 ranks:', d['n_gpus'], d['value'], d['parity'], d['config'].get('launch'), d['config']['workload'][:40])"
VPT_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --in-process --steps 5 --warmup 2 --no-e2e 2> $O/bench_inproc.err | tail -1 > $O/bench_2shards_in_process.json; python -c "
import json; d=json.load(open('$O/bench_2shards_in_process.json')); print('in-process:', d['n_gpus'], d['value'], d['parity'], d['config'].get('launch'))"
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'], [ (w['workload'][:10], w['roofline']['traffic']) for w in d['workloads']])"
VPT_FUZZ_SEED0=30000 timeout 200 python tools/fuzz_gpu.py 120 2>&1 | tail -2 | tee $O/fuzz.log
