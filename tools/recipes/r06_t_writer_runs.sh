# round 6: round 5's writer with this round's cheaper instructions (f1: 67 VGPRs, f8: 64 with two spills) at ITS run sizes (128 / 256 sentences; r06_s measured it at the
# tile kernel's 72), the tile kernel held to 5 waves at 85 sentences, and the tagged pipeline of configs[4] on the new sources (the tagged writer shares the gathers and the
# unconditional byte stores)
O=gpurun_out/r06_t; mkdir -p $O
python tools/writer_bench.py --variants f1,f8 --configs 1,2 --per-block 128,256 2>$O/bench.err | tee $O/writer_ab.jsonl | cut -c1-220
python tools/writer_bench.py --variants t5,t6 --configs 1,2 --per-block 85 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-220
python tools/tag_bench.py --variants new 2>$O/tag.err | tee $O/tag_bench.jsonl | cut -c1-400
