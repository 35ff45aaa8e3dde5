#!/bin/bash
# GPU call 2 of round 2: whole-position access mixes (v1 vs double-array v2), read-request size classes for the traffic accounting.
set -u
O=gpurun_out/c2; mkdir -p $O
export TMPDIR=/tmp
timeout 200 ./tools/gather_bench > $O/gather.txt 2>&1; echo "gather rc=$?"; tail -8 $O/gather.txt
R=$(pwd); cd /tmp
C="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/$O/calib_rdsz -- $R/tools/calib_fetch > $R/$O/calib_rdsz.log 2>&1 || echo "calib pass failed"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/$O/bench_rdsz -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/$O/bench_rdsz.log 2>&1 || echo "bench pass failed"
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum --output-format csv -d $R/$O/bench_dram -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/$O/bench_dram.log 2>&1 || echo "bench dram pass failed"
cd $R
python - <<'PY'
import glob, csv, collections
O="gpurun_out/c2"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    for c,vals in sorted(v.items()):
        print("%-42s %-28s n=%d avg=%.1f"%(k,c,len(vals),sum(vals)/len(vals)))
PY
