# round 6: the bigram node's two halves fetched by lane pairs (one L2 request per node instead of two: tools/tcp_bench's b32 against b16) -- A/B of the built
# library against -DVPT_FAST_PAIR_BI=0 in one process, configs[1]-sized batches on M1 and M2, then configs[2] through bench.py for both
O=gpurun_out/r06_m; mkdir -p $O
python tools/ab_bench.py --variants new,nopair --rounds 3 > $O/ab_m1.jsonl 2> $O/ab_m1.err; cut -c1-260 $O/ab_m1.jsonl
python tools/ab_bench.py --variants new,nopair --rounds 3 --model-kind 2 > $O/ab_m2.jsonl 2> $O/ab_m2.err; cut -c1-260 $O/ab_m2.jsonl
python tools/ab_bench.py --variants new,nopair --rounds 2 --sentences 2000000 --steps 10 > $O/ab_m1_2M.jsonl 2> $O/ab_m1_2M.err; cut -c1-260 $O/ab_m1_2M.jsonl
