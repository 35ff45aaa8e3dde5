# round 6: the lookups as a launch of their own (tag_resolve_kernel), candidates stored by the front end at their runs' places, the writer's suffixes from
# rec_str with the words and first strings of a piece's records in LDS -- GPU parity of everything that touches tags and the writers, tag_bench, kernel trace of configs[4]
O=gpurun_out/${VPT_OUT:-r06_d}; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -n 4 -k "tag or write or tokenize or emit or writer" 2>&1 | tail -6 ) > $O/gpu_tests_tags.log; tail -2 $O/gpu_tests_tags.log
python tools/tag_bench.py --variants new,new > $O/tag_bench.jsonl 2> $O/tag_bench.err; cat $O/tag_bench.jsonl | cut -c1-700; tail -3 $O/tag_bench.err
export TMPDIR=/tmp; REPO=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_c4 -- python $REPO/bench.py --config 4 --quick --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/$O/trace_c4.log 2>&1)
cat $(find $O/trace_c4 -name "*kernel_stats.csv" | head -1) > $O/c4_kernel_stats.csv; head -12 $O/c4_kernel_stats.csv | cut -c1-160
rm -rf $O/trace_c4
