# round 6, the evidence run on the final sources, ONE call: the GPU parity suite, smoke(), a fuzz, the counters of every workload (read-request size classes, WRITE_SIZE,
# the vector L1's lookups / its requests to the L2 / the L2's misses / busy cycles; the full counter set for configs[1], [2] and [4]), the entries merged into
# profiles/traffic.json (a copy comes back in gpurun_out), a kernel trace of configs[4] with the writer, then the whole bench -- its line carries the traffic and the
# gather block of THESE sources -- and the dry runs of the scaling jobs.   Usage: tools/gpu_run.sh r06_z_final [tag]
T=${1:-r06_z}; O=gpurun_out/$T; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/smoke.log
( timeout 400 python tools/fuzz_gpu.py 240 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_gpu.log; cat $O/fuzz_gpu.log
SMALL="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE|TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"
for C in 1 2 4; do
  ./tools/profile.sh ${T}_c$C --config $C > $O/profile_c$C.log 2>&1; grep "traffic entry" $O/profile_c$C.log | cut -c1-260
done
export VPT_PMC_GROUPS="$SMALL"
for C in 3 5 6 7 8; do
  ./tools/profile.sh ${T}_c$C --config $C > $O/profile_c$C.log 2>&1; grep "traffic entry" $O/profile_c$C.log | cut -c1-260
done
unset VPT_PMC_GROUPS
python tools/merge_traffic.py ${T}_c1 ${T}_c2 ${T}_c3 ${T}_c4 ${T}_c5 ${T}_c6 ${T}_c7 ${T}_c8 > $O/merge.log 2>&1; tail -3 $O/merge.log
cp profiles/traffic.json $O/traffic.json; mkdir -p $O/summaries; cp profiles/${T}_c*_summary.txt $O/summaries/ 2>/dev/null
export TMPDIR=/tmp; REPO=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_c4_emit -- python $REPO/bench.py --config 4 --quick --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/$O/trace_c4_emit.log 2>&1)
cat $(find $O/trace_c4_emit -name "*kernel_stats.csv" | head -1) > $O/c4_with_writer_kernel_stats.csv; head -10 $O/c4_with_writer_kernel_stats.csv | cut -c1-170; rm -rf $O/trace_c4_emit
SECONDS=0; python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench.py wall clock: $SECONDS s" | tee $O/bench_wall.txt; tail -c 300 $O/bench.err
python - <<PY
import json
l = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(len(json.dumps(l)), l["value"], l["roofline"]["frac"], l["roofline"]["traffic"], l["parity"])
print(json.dumps(l["roofline"].get("gather"))[:1400])
print(json.dumps({k: v for k, v in l["cpu_baseline"].items() if k != "sample"})[:900])
for w in l["workloads"]: print({k: w[k] for k in ("name", "value_G", "kernel_ms", "frac", "traffic_ratio", "parity", "tags_ms", "tags_dense_ms", "emit_ms", "tokenize_ms") if k in w})
PY
python bench.py --dry-scale --steps 10 --warmup 2 --no-e2e --no-emit > $O/dry_scale.jsonl 2> $O/dry_scale.err; tail -1 $O/dry_scale.err
python bench.py --dry-scale --scale-sweep 1,4 --scale-config 4 --steps 5 --warmup 2 --no-e2e --no-emit > $O/dry_scale_c4.jsonl 2> $O/dry_scale_c4.err; tail -1 $O/dry_scale_c4.err
rm -rf gpurun_out/prof_${T}_c*/pmc_* gpurun_out/prof_${T}_c*/trace
