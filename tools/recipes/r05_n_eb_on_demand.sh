# round 5: the trie replay reading a mini-table's SECOND entry only for lanes whose home entry holds another child (t2) against reading both together (w0 = the sources)
O=gpurun_out/r05_n; mkdir -p $O
python tools/ab_bench.py --variants r05w0,r05t2 --rounds 3 2>/dev/null > $O/ab_m1.jsonl; cut -c1-200 $O/ab_m1.jsonl
python tools/ab_bench.py --variants r05w0,r05t2 --rounds 3 --model-kind 2 2>/dev/null > $O/ab_m2.jsonl; cut -c1-200 $O/ab_m2.jsonl
