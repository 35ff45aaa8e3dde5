# round 5: why is the lighter kernel (15 % fewer VALU instructions, no SGPR spills) 4 % SLOWER?  Per-phase wave cycles of the round-4 kernel and the new one
# (diagnostics instances), and the new one without the parameter-block fences (d3) / with the node registers zeroed again (d5)
O=gpurun_out/r05_c; mkdir -p $O
python tools/ab_bench.py --variants r05base,r05d2,r05d3,r05d5 --rounds 3 2>/dev/null > $O/ab_m1.jsonl; cut -c1-200 $O/ab_m1.jsonl
python tools/ab_bench.py --variants r05base,r05d2 --rounds 1 --phases 2>/dev/null > $O/ab_m1_phases.jsonl; grep phases $O/ab_m1_phases.jsonl
