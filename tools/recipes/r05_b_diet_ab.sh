# round 5: the scoring kernel's instruction diet (steps 1, 2) against the round-4 kernel, same process (tools/ab_bench.py over prebuilt libraries), three
# workloads; configs[2] with parity through bench.py; and the attribution run VERDICT r4 item 2(e) asked for: LDS bank conflicts / instruction counts
# of the diagnostics instance with the deep-trie replays off (8), the trigram stage off (2), the pattern phase off (16) -- by subtraction, who conflicts
O=gpurun_out/r05_b; mkdir -p $O
export TMPDIR=/tmp
python tools/ab_bench.py --variants r05base,r05d1,new --rounds 3 2>/dev/null > $O/ab_m1.jsonl; cut -c1-230 $O/ab_m1.jsonl
python tools/ab_bench.py --variants r05base,new --rounds 2 --model-kind 2 2>/dev/null > $O/ab_m2.jsonl; cut -c1-230 $O/ab_m2.jsonl
python tools/ab_bench.py --variants r05base,new --rounds 2 --sentences 200000 --min-len 8 --max-len 512 2>/dev/null > $O/ab_ragged.jsonl; cut -c1-230 $O/ab_ragged.jsonl
python bench.py --config 2 --quick --no-emit --no-e2e --detail-out $O/c2_detail.json > $O/bench_c2.json 2> $O/bench_c2.err; python -c "
import json; l=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('configs[2]', l['value'], l['roofline']['kernel_ms'], l['roofline']['frac'], l['parity'])"
REPO=$(pwd)
for D in 64 72 66 16; do
  P=$REPO/$O/pmc_$D; mkdir -p $P
  (cd /tmp && VPT_DEBUG_ABLATE=$D rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $P -- python $REPO/bench.py --config 1 --quick --steps 5 --warmup 1 --no-cpu-baseline --no-emit --no-e2e > $P/log.txt 2>&1)
  python - <<PY
import glob,csv,collections
agg=collections.defaultdict(list)
for f in glob.glob("$P/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_tiles_fast" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate $D:", {k: round(sum(v)/len(v)) for k,v in sorted(agg.items())})
open("$REPO/$O/lds_attribution.txt","a").write("VPT_DEBUG_ABLATE=$D " + str({k: round(sum(v)/len(v)) for k,v in sorted(agg.items())}) + "\n")
PY
done
