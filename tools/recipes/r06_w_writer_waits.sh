# round 6: the writer's waits: one barrier for a block prefix sum that comes once a piece (two of a piece's seven gone) and the size pass's loads in one loop, six in
# flight (three trips for a run of 48 KB + 16 KB of labels where seven were) -- `new` against `pre` (a3085b1), same box, twice; the writer tests; the tagged pipeline
O=gpurun_out/r06_w; mkdir -p $O
for R in 1 2; do python tools/writer_bench.py --variants pre,new --configs 1,2,5 --no-parity 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-200; done
python tools/writer_bench.py --variants new --configs 1,2 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-220
( timeout 900 python -m pytest tests -m gpu -x -q -n 4 -k "writ or tokeniz or emit or error" 2>&1 | tail -4 ) > $O/gpu_writer_tests.log; tail -2 $O/gpu_writer_tests.log
python tools/tag_bench.py --variants new 2>$O/tag.err | tee $O/tag_bench.jsonl | cut -c1-400
