# round 6: the flat writer with a dword's bytes and spaces picked by v_perm_b32 (selectors in LDS), ranges in 32 bits, 8 waves per SIMD without spills; runs of
# 64 .. 256 sentences on configs[1], [2] and the documents; the writer tests of the GPU suite; the tagged pipeline; the instruction counters on configs[2]
O=gpurun_out/r06_u; mkdir -p $O
python tools/writer_bench.py --variants new --configs 1,2,5 --per-block 0,64,96,128,192,256 2>$O/bench.err | tee $O/writer_ab.jsonl | cut -c1-220
( timeout 900 python -m pytest tests -m gpu -x -q -n 4 -k "writ or tokeniz or emit or error" 2>&1 | tail -4 ) > $O/gpu_writer_tests.log; tail -2 $O/gpu_writer_tests.log
python tools/tag_bench.py --variants new 2>$O/tag.err | tee $O/tag_bench.jsonl | cut -c1-400
cd /tmp
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$N -- python $GRAFT_REPO_ROOT/tools/writer_bench.py --variants new --configs 2 --steps 3 --no-parity > $GRAFT_REPO_ROOT/$O/pmc_$N.log 2>&1 || echo "pass failed: $G" >> $GRAFT_REPO_ROOT/$O/failed.txt
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r06_u/writer_counters.txt
import glob, csv, collections, os
out = "gpurun_out/r06_u"
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "emit_flat" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== emit_flat_kernel<false> on configs[2], per launch")
for c, v in sorted(agg.items()):
    print("%-28s n=%d avg=%.0f" % (c, len(v), sum(v) / len(v)))
PY
rm -rf $O/pmc_*/
