# round 6: is the untagged writer bound by the vector ALU's issue rate?  A/B of the flag gathers as v_dot4_u32_u8 + the bytes going out unconditionally (w1: what the
# compiler takes, 95 VGPRs; w6 / w8: held to 6 / 8 waves per SIMD, with spills) against the round's sources before (base), configs[1] and configs[2]; then the
# instruction counters of emit_flat_kernel for base and w1 on configs[2]
O=gpurun_out/r06_r; mkdir -p $O
python tools/writer_bench.py --variants base,w1,w6,w8 --configs 1,2 2>$O/bench.err | tee $O/writer_ab.jsonl | cut -c1-200
cd /tmp
for V in base w1; do
  for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
    N=$(echo $G | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${V}_$N -- python $GRAFT_REPO_ROOT/tools/writer_bench.py --variants $V --configs 2 --steps 3 --no-parity > $GRAFT_REPO_ROOT/$O/pmc_${V}_$N.log 2>&1 || echo "pass failed: $V $G" >> $GRAFT_REPO_ROOT/$O/failed.txt
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r06_r/writer_counters.txt
import glob, csv, collections, os
out = "gpurun_out/r06_r"
for V in ("base", "w1"):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_%s_*" % V, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "emit_flat" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s: emit_flat_kernel<false> on configs[2], per launch" % V)
    for c, v in sorted(agg.items()):
        print("%-28s n=%d avg=%.0f" % (c, len(v), sum(v) / len(v)))
PY
