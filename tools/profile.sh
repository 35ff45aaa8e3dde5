#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# Usage: tools/profile.sh <tag> [bench args...]     (extra counter groups: env VPT_PMC_GROUPS="A B|C D")
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
DEFAULT_GROUPS="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum|SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY|SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS|GRBM_GUI_ACTIVE"
GROUPS_STR="${VPT_PMC_GROUPS:-$DEFAULT_GROUPS}"
IFS='|' read -ra GRPS <<< "$GROUPS_STR"
for C in "${GRPS[@]}"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-80)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1 || echo "pass failed: $C" >> $OUT/failed.txt
done
cd $REPO
python - <<PY
import glob, csv, os, collections
out="$OUT"
def rows(pat):
    for f in glob.glob(os.path.join(out, pat), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)
with open(os.path.join(out,"summary.txt"),"w") as w:
    for f in glob.glob(os.path.join(out,"trace","**","*kernel_stats.csv"), recursive=True):
        w.write("== kernel stats (%s)\n" % os.path.basename(f)); w.write(open(f).read()+"\n")
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows("pmc_*/**/*counter_collection.csv"):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    w.write("== PMC per-dispatch averages\n")
    for k,v in agg.items():
        if "score" not in k: continue
        for c,vals in sorted(v.items()):
            w.write("%-62s %-34s n=%d avg=%.1f\n" % (k,c,len(vals),sum(vals)/len(vals)))
    if os.path.exists(os.path.join(out,"failed.txt")):
        w.write("== failed passes\n"+open(os.path.join(out,"failed.txt")).read())
print(open(os.path.join(out,"summary.txt")).read()[:8000])
PY
