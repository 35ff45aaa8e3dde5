# round 6, compact tag records: the GPU parity tests that touch tags and the writers, bench.py --config 4 (its tags / dense tags / tagged
# writer times, parity of every token against the oracle), and the kernel trace of the same with the writer: what the front end, the passes and the tagged writer take now
O=gpurun_out/r06_b; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "tag or write or tokenize or emit or writer" 2>&1 | tail -6 ) > $O/gpu_tests_tags.log; tail -2 $O/gpu_tests_tags.log
python bench.py --config 4 --quick --detail-out $O/bench_c4_detail.json > $O/bench_c4.json 2> $O/bench_c4.err; tail -3 $O/bench_c4.err
python -c "
import json; l=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['parity'], json.dumps(l.get('tags'))[:900], json.dumps(l.get('emit'))[:500])"
export TMPDIR=/tmp; REPO=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_c4 -- python $REPO/bench.py --config 4 --quick --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/$O/trace_c4.log 2>&1)
cat $(find $O/trace_c4 -name "*kernel_stats.csv" | head -1) > $O/c4_kernel_stats.csv; head -16 $O/c4_kernel_stats.csv | cut -c1-200
rm -rf $O/trace_c4
