#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command, one pass per counter group
# (never combined with a trace domain).  Writes gpurun_out/prof_<tag>/summary.txt and -- from the read-request size
# classes and WRITE_SIZE of the same run -- the dominant kernel's HBM-side bytes per launch as an entry for
# profiles/traffic.json (gpurun_out/prof_<tag>/traffic_entry.json), keyed by the hash of the kernel sources it ran.
# Usage: tools/profile.sh <tag> [bench args...]     (extra counter groups: env VPT_PMC_GROUPS="A B|C D")
set -u
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-emit --quick $*"   # (--no-emit: no writer and no fused-writer instance of the scoring kernel, whose launches would be averaged in)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
DEFAULT_GROUPS="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE|TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum|TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU|SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
GROUPS_STR="${VPT_PMC_GROUPS:-$DEFAULT_GROUPS}"
IFS='|' read -ra GRPS <<< "$GROUPS_STR"
for C in "${GRPS[@]}"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-80)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- $CMD > $OUT/pmc_$N.log 2>&1 || echo "pass failed: $C" >> $OUT/failed.txt
done
cd $REPO
python - "$TAG" "$*" <<PY
import glob, csv, os, collections, json, sys
sys.path.insert(0, "$REPO")
out="$OUT"
def rows(pat):
    for f in glob.glob(os.path.join(out, pat), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows("pmc_*/**/*counter_collection.csv"):
    agg[r["Kernel_Name"][:96]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out,"summary.txt"),"w") as w:
    for f in glob.glob(os.path.join(out,"trace","**","*kernel_stats.csv"), recursive=True):
        w.write("== kernel stats (%s)\n" % os.path.basename(f)); w.write(open(f).read()+"\n")
    w.write("== PMC per-dispatch averages\n")
    for k,v in agg.items():
        if not any(x in k for x in ("score", "tag_tokens", "tag_front", "tag_pass", "decode_chars", "emit", "count", "cut_", "assign_tiles")): continue
        for c,vals in sorted(v.items()):
            w.write("%-98s %-34s n=%d avg=%.1f\n" % (k,c,len(vals),sum(vals)/len(vals)))
    if os.path.exists(os.path.join(out,"failed.txt")):
        w.write("== failed passes\n"+open(os.path.join(out,"failed.txt")).read())
# traffic entry for the dominant kernel
import bench
line = None
for l in open(os.path.join(out, "trace.log")):
    if l.startswith("{"): line = json.loads(l)
for k, v in agg.items():
    if "score_tiles" not in k or ", true>" in k or "TCC_EA0_RDREQ_128B_sum" not in v or "WRITE_SIZE" not in v: continue   # (", true>": the diagnostics instance that counts node reads)
    avg = lambda c: sum(v[c]) / len(v[c]) if c in v else 0.0
    entry = {"round": sys.argv[1], "kernel": line["roofline"]["kernel"] if line else "score_tiles_fast_kernel",
             "model": line["config"]["tokenizer_model"] if line else None, "workload": line["config"]["workload"].split(":")[0] if line else None,
             "source_hash": bench.kernel_source_hash(),
             "read_bytes": int(128 * avg("TCC_EA0_RDREQ_128B_sum") + 64 * avg("TCC_EA0_RDREQ_64B_sum") + 32 * avg("TCC_EA0_RDREQ_32B_sum")),
             "write_bytes": int(1024 * avg("WRITE_SIZE")),
             # the gather side (bench.py, roofline.gather): the vector L1's lookups, its requests to the L2, the L2's misses, the L1s' busy cycles
             "tcp_lookups": int(avg("TCP_TOTAL_CACHE_ACCESSES_sum")), "tcp_tcc_read_req": int(avg("TCP_TCC_READ_REQ_sum")), "tcc_miss": int(avg("TCC_MISS_sum")),
             "tcc_hit": int(avg("TCC_HIT_sum")), "tcp_busy_cycles": int(avg("TCP_GATE_EN1_sum")), "tcp_cycles": int(avg("GRBM_GUI_ACTIVE") / 8 * 256),
             "source": "gpurun_out/prof_%s (rocprofv3 --pmc, separate passes: read-request size classes, WRITE_SIZE), bench args: %s" % (sys.argv[1], sys.argv[2])}
    json.dump(entry, open(os.path.join(out, "traffic_entry.json"), "w"), indent=1)
    print("traffic entry:", json.dumps(entry))
print(open(os.path.join(out,"summary.txt")).read()[:6000])
PY
