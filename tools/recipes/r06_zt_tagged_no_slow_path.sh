# round 6 (what-if, not a product build): the tagged writer WITHOUT its char-by-char path for threads with three tags in sixteen bytes (wrong where that happens) and 192 stash entries:
# 96 VGPRs, no scratch, 32.0 KB of LDS -- five waves per SIMD for real (ns5) against the tree (new): what the fifth wave is worth before building it
O=gpurun_out/r06_zt; mkdir -p $O
for R in 1 2; do python tools/tag_bench.py --variants new,ns5 2>>$O/tag.err | tee -a $O/tag_bench.jsonl | cut -c1-420; done
