# round 6: which unit the tag kernels of configs[4] run at: instruction counters per launch (tools/tag_bench.py under rocprofv3 --pmc)
O=gpurun_out/r06_zw; mkdir -p $O
cd /tmp
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$N -- python $GRAFT_REPO_ROOT/tools/tag_bench.py --variants new --steps 3 > $GRAFT_REPO_ROOT/$O/pmc_$N.log 2>&1 || echo "pass failed: $G" >> $GRAFT_REPO_ROOT/$O/failed.txt
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r06_zw/tag_counters.txt
import glob, csv, collections, os
out = "gpurun_out/r06_zw"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("tag_front_flat", "tag_pass", "tag_resolve", "decode_chars", "emit_flat_kernel<true>", "cut_count", "count_chars"):
            if name in k: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, v in agg.items():
    g = lambda c: (sum(v[c]) / len(v[c])) if c in v else 0.0
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    print("== %s: per launch: XCD cycles %.0f (%.3f ms at 2.4 GHz)  VALU %.1f M (%.0f %% of the SIMDs' issue slots)  SALU %.1f M (%.0f %% of the CUs' scalar slots)  LDS %.1f M  VMEM rd/wr %.1f / %.1f M  waves %.0f  wait share %.2f" % (
        name, cyc, cyc / 2.4e6, g("SQ_INSTS_VALU") / 1e6, 100 * g("SQ_INSTS_VALU") * 4 / (1024 * cyc) if cyc else 0, g("SQ_INSTS_SALU") / 1e6, 100 * g("SQ_INSTS_SALU") / (256 * cyc) if cyc else 0,
        g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_VMEM_RD") / 1e6, g("SQ_INSTS_VMEM_WR") / 1e6, g("SQ_WAVES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else 0))
PY
rm -rf $O/pmc_*/
