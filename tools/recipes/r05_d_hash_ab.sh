# round 5: is it the table hashes?  The lighter kernel with the 16 x 24-bit hashes (e1 = the sources), with the round-4 32-bit hashes (e2), the 16 x 16-bit
# ones (d2), against the round-4 kernel; and the node reads per trie level of the sources' tables (round 4: 6.24 / 6.01 / 2.94 / 2.07 M at share 0.7)
O=gpurun_out/r05_d; mkdir -p $O
python tools/ab_bench.py --variants r05base,r05d2,r05e1,r05e2 --rounds 3 2>/dev/null > $O/ab_m1.jsonl; cut -c1-200 $O/ab_m1.jsonl
python tools/hit_share_sweep.py --shares 0.7 2>/dev/null > $O/node_reads.jsonl; cut -c1-600 $O/node_reads.jsonl
true
