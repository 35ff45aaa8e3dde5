# round 5: the bigram node's second half (trigram base + child filter) asked for only by the lanes whose key matched -- a fourth trip in flight (t3 = the sources)
# against both halves together (fin): three workloads + the 10 M-sentence batch through a short bench run of each library
O=gpurun_out/r05_q; mkdir -p $O
python tools/ab_bench.py --variants r05fin,r05t3 --rounds 3 2>/dev/null > $O/ab_m1.jsonl; cut -c1-200 $O/ab_m1.jsonl
python tools/ab_bench.py --variants r05fin,r05t3 --rounds 3 --model-kind 2 2>/dev/null > $O/ab_m2.jsonl; cut -c1-200 $O/ab_m2.jsonl
python tools/ab_bench.py --variants r05fin,r05t3 --rounds 2 --sentences 200000 --min-len 8 --max-len 512 2>/dev/null > $O/ab_ragged.jsonl; cut -c1-200 $O/ab_ragged.jsonl
python tools/ab_bench.py --variants r05fin,r05t3 --rounds 2 --sentences 2000000 --steps 10 2>/dev/null > $O/ab_2m.jsonl; cut -c1-200 $O/ab_2m.jsonl
