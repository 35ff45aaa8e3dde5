#!/bin/bash
# scratch: the script of the latest gpurun call -- rocprofv3 kernel trace of configs[4] (scoring, tags, tagged writer) on the final sources
O=gpurun_out/r03_zx; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --quick > $OLDPWD/$O/trace.log 2>&1; cd $OLDPWD
cat $O/trace/*/*kernel_stats.csv | cut -c1-200 | head -16
