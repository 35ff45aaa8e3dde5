# round 5: the lighter kernel issues 22 % MORE vector-memory reads per launch than the round-4 one (profiles/r05_e_*): which phase?  Both diagnostics instances
# with nothing off (64), the deep-trie replays off (72), the pattern phase off (16)
O=gpurun_out/r05_f; mkdir -p $O; REPO=$(pwd); export TMPDIR=/tmp
for V in r05base r05e2; do
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  for D in 64 72 16; do
    P=$REPO/$O/pmc_${V}_$D; mkdir -p $P
    (cd /tmp && VPT_DEBUG_ABLATE=$D rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU TCP_TOTAL_ACCESSES_sum SQ_WAVES --output-format csv -d $P -- python $REPO/bench.py --config 1 --quick --steps 5 --warmup 1 --no-cpu-baseline --no-emit --no-e2e > $P/log.txt 2>&1)
    python - <<PY
import glob,csv,collections
agg=collections.defaultdict(list)
for f in glob.glob("$P/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_tiles_fast" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
line = "$V VPT_DEBUG_ABLATE=$D " + str({k: round(sum(v)/len(v)) for k,v in sorted(agg.items())})
print(line); open("$REPO/$O/vmem_where.txt","a").write(line + "\n")
PY
  done
done
