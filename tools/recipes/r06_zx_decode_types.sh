# round 6: decode_chars_kernel with the char types of ASCII / U+30xx / U+FFxx as nibbles in LDS and the main kanji block as one compare (char_type's twenty range tests only
# for what is left), a chunk's chars in one loop, 32-bit positions -- `new` against `head`, same box, twice (stand-alone fill_tags includes the decode); the kernel's time
# by rocprofv3; the tests that compare char types and tags with the oracle
O=gpurun_out/r06_zx; mkdir -p $O
for R in 1 2; do python tools/tag_bench.py --variants head,new 2>>$O/tag.err | tee -a $O/tag_bench.jsonl | cut -c1-300; done
( timeout 900 python -m pytest tests -m gpu -x -q -n 4 -k "tag or char_types or fullwidth or kytea" 2>&1 | tail -4 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/tools/tag_bench.py --variants head,new --steps 5 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; cat $(find $O/trace -name "*kernel_stats.csv" | head -1) | grep -i "decode\|Name" | cut -c1-200 | tee $O/decode_kernel_stats.csv; rm -rf $O/trace
