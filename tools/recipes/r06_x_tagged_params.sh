# round 6: the tagged writer with its parameter block in scalar registers (the suffix routines take what they read by value: t4; t5 = the same held to 5 waves per SIMD)
# against the block in scratch memory, a reference to it passed to a called routine (tpre = f8825df), same box, twice; then the tagged tests of the GPU suite on the new library
O=gpurun_out/r06_x; mkdir -p $O
for R in 1 2; do python tools/tag_bench.py --variants tpre,t4,t5 2>>$O/tag.err | tee -a $O/tag_bench.jsonl | cut -c1-420; done
( timeout 900 python -m pytest tests -m gpu -x -q -n 4 -k "tag or writ or tokeniz" 2>&1 | tail -4 ) > $O/gpu_tag_tests.log; tail -2 $O/gpu_tag_tests.log
