#!/bin/bash
# instruction counts of the scoring kernel under ablations (where do the issue slots go?)
export TMPDIR=/tmp
REPO=$(pwd)
for D in 0 1 16; do
  OUT=$REPO/gpurun_out/inst_$D; mkdir -p $OUT
  (cd /tmp && VPT_DEBUG_ABLATE=$D rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1)
  echo "== ablate $D"; python - <<PY
import glob,csv,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_tiles_fast" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("  %-22s %.0f  per wave %.0f" % (k, sum(v)/len(v), sum(v)/len(v)/20032))
PY
done
