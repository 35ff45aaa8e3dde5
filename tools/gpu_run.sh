#!/bin/bash
set -u
O=gpurun_out/c17; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python bench.py --config 4 --quick --steps 10 --warmup 2 > $O/bench_tags.json 2> $O/bench_tags.err; echo "bench tags rc=$?"; tail -2 $O/bench_tags.err; python -c "
import json;d=json.loads(open('$O/bench_tags.json').read().strip().splitlines()[-1]);print('tags',d.get('tags'));print('ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'],'parity',d['parity'])"
R=$(pwd); cd /tmp
CMD="python $R/bench.py --config 4 --quick --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --sentences 300000"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- $CMD > $R/$O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $R/$O/pmc -- $CMD > $R/$O/pmc.log 2>&1
cd $R; cat $O/trace/*/*kernel_stats.csv | cut -c1-120 | head -5
python - <<'PY'
import glob, csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/c17/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "tag_tokens" not in k: continue
    for c,vals in sorted(v.items()): print("%-30s avg=%.0f"%(c,sum(vals)/len(vals)))
PY
timeout 200 python tools/fuzz_gpu.py 45 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
