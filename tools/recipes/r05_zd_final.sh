# round 5, the evidence run on the final sources after the rewrite of fill_tags' front end (kernels_tags.hip only: the scoring kernel's sources and their traffic entries
# are the ones of r05_z2): the GPU parity suite, smoke(), the default bench (its line and wall clock), the counters and the kernel trace of configs[4]
O=gpurun_out/r05_zd; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
SECONDS=0; python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench.py wall clock: $SECONDS s" | tee $O/bench_wall.txt
python -c "
import json; l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(len(json.dumps(l)), l['value'], l['roofline']['frac'], l['roofline']['traffic'], l['parity'], [(w['name'], w.get('frac'), w.get('parity'), w.get('tags_ms'), w.get('emit_ms')) for w in l['workloads']])"
./tools/profile.sh r05_zd_c4 --config 4 > $O/profile_c4.log 2>&1; grep "traffic entry" $O/profile_c4.log | cut -c1-200
export TMPDIR=/tmp; REPO=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_c4_emit -- python $REPO/bench.py --config 4 --quick --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/$O/trace_c4_emit.log 2>&1)
cat $(find $O/trace_c4_emit -name "*kernel_stats.csv" | head -1) > $O/c4_with_writer_kernel_stats.csv; head -12 $O/c4_with_writer_kernel_stats.csv | cut -c1-180
rm -rf $O/trace_c4_emit
