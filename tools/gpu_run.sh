#!/bin/bash
set -u
O=gpurun_out/c13; mkdir -p $O
export TMPDIR=/tmp
timeout 900 bash tools/profile.sh r02_d > $O/profile.log 2>&1; echo "profile rc=$?"; grep "traffic entry" $O/profile.log
timeout 900 bash tools/profile.sh r02_d_m2 --config 3 > $O/profile_m2.log 2>&1; echo "profile m2 rc=$?"; grep "traffic entry" $O/profile_m2.log
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?"; tail -3 $O/bench.time; cut -c1-600 $O/bench.json
