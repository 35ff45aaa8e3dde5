# round 6: the untagged writer as a tile read once (emit_tile_kernel: t1 = what the compiler takes, 117 VGPRs; t5 / t6 = held to 5 / 6 waves per SIMD, with spills)
# against round 5's kernel with the cheaper flag gathers and unconditional byte stores (f1: 67 VGPRs; f8: 64 with two spills), configs[1], [2] and the documents;
# runs of 64 / 72 (what the library takes) / 85 sentences for t1; the writer tests of the GPU suite on the new library; the counters of t1 on configs[2]
O=gpurun_out/r06_s; mkdir -p $O
python tools/writer_bench.py --variants t1,t5,t6,f1,f8 --configs 1,2,5 2>$O/bench.err | tee $O/writer_ab.jsonl | cut -c1-220
python tools/writer_bench.py --variants t1 --configs 1,2 --per-block 48,64,85 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-220
( timeout 900 python -m pytest tests -m gpu -x -q -n 4 -k "writ or tokeniz or emit or error" 2>&1 | tail -4 ) > $O/gpu_writer_tests.log; tail -2 $O/gpu_writer_tests.log
cd /tmp
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_t1_$N -- python $GRAFT_REPO_ROOT/tools/writer_bench.py --variants t1 --configs 2 --steps 3 --no-parity > $GRAFT_REPO_ROOT/$O/pmc_t1_$N.log 2>&1 || echo "pass failed: $G" >> $GRAFT_REPO_ROOT/$O/failed.txt
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r06_s/writer_counters.txt
import glob, csv, collections, os
out = "gpurun_out/r06_s"
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_t1_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "emit_tile" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== t1: emit_tile_kernel on configs[2], per launch")
for c, v in sorted(agg.items()):
    print("%-28s n=%d avg=%.0f" % (c, len(v), sum(v) / len(v)))
PY
rm -rf $O/pmc_t1_*/
