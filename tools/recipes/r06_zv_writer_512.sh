# round 6: the writer's runs up to 512 sentences (two a thread) -- `new` (auto: 32 K chars on a big batch) and forced 256 / 384 / 512 against `head` (at most 256), same box, twice;
# the tagged pipeline (its runs stay multiples of fill_tags'); the writer tests
O=gpurun_out/r06_zv; mkdir -p $O
for R in 1 2; do
  python tools/writer_bench.py --variants head --configs 1,2,5 --no-parity 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-200
  python tools/writer_bench.py --variants new --configs 1,2,5 --per-block 0,256,384,512 --no-parity 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-200
done
python tools/tag_bench.py --variants head,new 2>$O/tag.err | tee $O/tag_bench.jsonl | cut -c1-420
( timeout 900 python -m pytest tests -m gpu -x -q -n 4 -k "writ or tokeniz or emit" 2>&1 | tail -4 ) > $O/gpu_writer_tests.log; tail -2 $O/gpu_writer_tests.log
