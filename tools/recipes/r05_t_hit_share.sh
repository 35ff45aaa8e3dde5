# round 5: the text-realism sweep (hit shares 0.1 .. 0.9) on the final kernel, M1 and M2, with the node reads per trie level
O=gpurun_out/r05_t; mkdir -p $O
python tools/hit_share_sweep.py 2>/dev/null > $O/hit_share_sweep.jsonl; python -c "
import json
for l in open('$O/hit_share_sweep.jsonl'):
    d=json.loads(l); print(d['hit_share'], round(d['kernel_ms'],4), round(d['frac'],3), d['parity'], d['node_reads_per_launch'])"
python tools/hit_share_sweep.py --model-kind 2 2>/dev/null > $O/hit_share_sweep_m2.jsonl; python -c "
import json
for l in open('$O/hit_share_sweep_m2.jsonl'):
    d=json.loads(l); print(d['hit_share'], round(d['kernel_ms'],4), round(d['frac'],3), d['parity'])"
