# round 5: the lighter kernel with the wide rows batched again and Fibonacci 24-bit table hashes (f1 = the sources) against the round-4 kernel: three workloads,
# the node reads per trie level, configs[2] through bench.py
O=gpurun_out/r05_g; mkdir -p $O
python tools/ab_bench.py --variants r05base,r05f1 --rounds 3 2>/dev/null > $O/ab_m1.jsonl; cut -c1-200 $O/ab_m1.jsonl
python tools/ab_bench.py --variants r05base,r05f1 --rounds 2 --model-kind 2 2>/dev/null > $O/ab_m2.jsonl; cut -c1-200 $O/ab_m2.jsonl
python tools/ab_bench.py --variants r05base,r05f1 --rounds 2 --sentences 200000 --min-len 8 --max-len 512 2>/dev/null > $O/ab_ragged.jsonl; cut -c1-200 $O/ab_ragged.jsonl
python tools/hit_share_sweep.py --shares 0.7 2>/dev/null > $O/node_reads.jsonl; python -c "
import json; d=json.loads(open('$O/node_reads.jsonl').read().strip().splitlines()[-1]); print(d['kernel_ms'], d['node_reads_per_launch'])"
python bench.py --config 2 --quick --no-emit --no-e2e --detail-out $O/c2_detail.json > $O/bench_c2.json 2> $O/bench_c2.err; python -c "
import json; l=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('configs[2]', l['value'], l['roofline']['kernel_ms'], l['roofline']['frac'], l['parity'])"
