#!/bin/bash
# GPU call 4 of round 2: parity suite with the new entry points, the new bench line (all workloads), traffic entry, 2-rank path on one GPU.
set -u
O=gpurun_out/c4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?"; tail -3 $O/bench.time; tail -5 $O/bench.err; cut -c1-3000 $O/bench.json
timeout 900 bash tools/profile.sh r02_c > $O/profile.log 2>&1; echo "profile rc=$?"; grep "traffic entry" $O/profile.log
( time VPT_BENCH_ONE_DEVICE=1 VPT_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --sentences 1000000 > $O/bench2.json 2> $O/bench2.err ) 2> $O/bench2.time; echo "bench2 rc=$?"; tail -3 $O/bench2.time; tail -5 $O/bench2.err; cut -c1-2500 $O/bench2.json
