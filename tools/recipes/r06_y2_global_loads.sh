# round 6: loads whose address went through an integer as GLOBAL loads (they were FLAT: counted against the LDS's counter too) -- `new` against `fpre` (6ab4c49), same box:
# the scoring kernel on configs[1]'s batch and on 2 M sentences, the writer, the tagged pipeline (its stand-alone fill_tags includes the decode), tokenize
O=gpurun_out/r06_y2; mkdir -p $O
python tools/ab_bench.py --variants fpre,new --rounds 3 2>$O/ab.err | tee $O/ab_c1.jsonl | cut -c1-260
python tools/ab_bench.py --variants fpre,new --rounds 2 --sentences 2000000 --steps 10 2>>$O/ab.err | tee $O/ab_2m.jsonl | cut -c1-260
for R in 1 2; do python tools/writer_bench.py --variants fpre,new --configs 1,2,5 --no-parity 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-200; done
for R in 1 2; do python tools/tag_bench.py --variants fpre,new 2>>$O/tag.err | tee -a $O/tag_bench.jsonl | cut -c1-420; done
