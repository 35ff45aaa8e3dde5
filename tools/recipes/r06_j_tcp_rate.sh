# round 6 (VERDICT r5 items 2 + 4): what bounds phase B.  (1) tools/tcp_bench: random loads by shape from L1- / L2- / HBM-resident tables: lane-items per clock and CU;
# (2) the same under --pmc (no trace domain): TCP_TOTAL_ACCESSES per lane-item of every shape = the counter's unit; (3) the counter on the scoring kernel, configs[1] and [2]
O=gpurun_out/r06_j; mkdir -p $O
export TMPDIR=/tmp; REPO=$(pwd)
tools/tcp_bench > $O/tcp_bench.jsonl 2>&1; cat $O/tcp_bench.jsonl | cut -c1-200
cd /tmp
for G in "TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $G --output-format csv -d $REPO/$O/pmc_tcp_$N -- $REPO/tools/tcp_bench > $REPO/$O/pmc_tcp_$N.log 2>&1 || echo "pass failed: $G" >> $REPO/$O/failed.txt
  for C in 1 2; do
    rocprofv3 --pmc $G --output-format csv -d $REPO/$O/pmc_c${C}_$N -- python $REPO/bench.py --config $C --quick --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-emit > $REPO/$O/pmc_c${C}_$N.log 2>&1 || echo "pass failed: $G (config $C)" >> $REPO/$O/failed.txt
  done
done
cd $REPO
python - <<PY
import glob, csv, os, collections
out = "$O"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    tag = f.split("/pmc_")[1].split("_")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gather_" in k or "score_tiles" in k:
            agg[(tag, k[:70])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as w:
    for (tag, k), v in sorted(agg.items()):
        w.write("%-4s %-72s %s\n" % (tag, k, "  ".join("%s=%.0f (n=%d)" % (c, sum(x) / len(x), len(x)) for c, x in sorted(v.items()))))
    if os.path.exists(os.path.join(out, "failed.txt")):
        w.write(open(os.path.join(out, "failed.txt")).read())
print(open(os.path.join(out, "summary.txt")).read()[:5000])
PY
rm -rf $O/pmc_*/
