# round 5: an 8-bit child filter in the trigram node for the first deep step (t1 = the sources) against none (w0): three workloads, the node reads per level
O=gpurun_out/r05_p; mkdir -p $O
python tools/ab_bench.py --variants r05w0,r05t1 --rounds 3 2>/dev/null > $O/ab_m1.jsonl; cut -c1-200 $O/ab_m1.jsonl
python tools/ab_bench.py --variants r05w0,r05t1 --rounds 3 --model-kind 2 2>/dev/null > $O/ab_m2.jsonl; cut -c1-200 $O/ab_m2.jsonl
python tools/ab_bench.py --variants r05w0,r05t1 --rounds 2 --sentences 200000 --min-len 8 --max-len 512 2>/dev/null > $O/ab_ragged.jsonl; cut -c1-200 $O/ab_ragged.jsonl
python tools/hit_share_sweep.py --shares 0.7 2>/dev/null > $O/node_reads.jsonl; python -c "
import json; d=json.loads(open('$O/node_reads.jsonl').read().strip().splitlines()[-1]); print(d['kernel_ms'], d['node_reads_per_launch'])"
python tools/hit_share_sweep.py --shares 0.7 --model-kind 2 2>/dev/null > $O/node_reads_m2.jsonl; python -c "
import json; d=json.loads(open('$O/node_reads_m2.jsonl').read().strip().splitlines()[-1]); print(d['kernel_ms'], d['node_reads_per_launch'])"
