# round 4, sixth call: same-process A/B of the LDS char table (VPT_FAST_CTAB: symbol words of ASCII / U+3000..30FF / U+FF00..FFEF in LDS) and
# of smaller tiles (more rounds on a 100 K-sentence batch) against the built library; configs[1]'s batch and a 1 M-sentence one; M2
O=gpurun_out/r04_f; mkdir -p $O
python tools/ab_bench.py --variants new,ctab,new:VPT_TILE_FLAT=960,new:VPT_TILE_FLAT=640,ctab:VPT_TILE_FLAT=960 --rounds 3 2>&1 | grep -v amdgpu.ids > $O/ab_m1.jsonl
python tools/ab_bench.py --variants new,ctab --sentences 1000000 --rounds 2 --steps 15 2>&1 | grep -v amdgpu.ids > $O/ab_m1_1m.jsonl
python tools/ab_bench.py --variants new,ctab --model-kind 2 --rounds 2 2>&1 | grep -v amdgpu.ids > $O/ab_m2.jsonl
python tools/ab_bench.py --variants new,ctab --min-len 8 --max-len 512 --rounds 2 2>&1 | grep -v amdgpu.ids > $O/ab_ragged.jsonl
cat $O/*.jsonl | cut -c1-250
