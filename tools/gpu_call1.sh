#!/bin/bash
# GPU call 1 of round 2: parity suite, A/B of the round-1 experiments, timing ablations, microbenchmarks, PMC calibration.
set -u
O=gpurun_out/c1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python tools/ab_bench.py --variants new,new:VPT_SEPARATE_ASSIGN=1,nostream,nostream:VPT_SEPARATE_ASSIGN=1 --ablate 256,257,260,320,264,258,288,272,261,321,324,325 --rounds 2 > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?"
timeout 120 ./tools/gather_bench > $O/gather.txt 2>&1; echo "gather rc=$?"
R=$(pwd)
cd /tmp
for C in "FETCH_SIZE TCC_EA0_RDREQ_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCC_REQ_sum TCC_READ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-60)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/$O/calib_$N -- $R/tools/calib_fetch > $R/$O/calib_$N.log 2>&1 || echo "calib pass failed: $C"
done
cd $R
python - <<'PY'
import glob, csv, collections, json, os
O="gpurun_out/c1"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/calib_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O+"/calib_summary.txt","w") as w:
    for k,v in sorted(agg.items()):
        for c,vals in sorted(v.items()):
            w.write("%-28s %-28s n=%d avg=%.1f\n"%(k,c,len(vals),sum(vals)/len(vals)))
print(open(O+"/calib_summary.txt").read())
PY
grep -h '^{' $O/calib_FETCH*.log | head -8 > $O/calib_known_bytes.jsonl
VPT_PMC_GROUPS="FETCH_SIZE|WRITE_SIZE|TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU|SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" timeout 900 bash tools/profile.sh r02_a > $O/profile.log 2>&1; echo "profile rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | cut -c1-600
cat $O/ab.jsonl
cat $O/gather.txt
