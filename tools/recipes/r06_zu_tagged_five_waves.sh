# round 6: the tagged writer was held to 4 workgroups per CU by its LDS (33.6 KB) as well as by its registers: 192 stash entries (32.0 KB) and 5 waves per SIMD
# (96 VGPRs, 48 bytes of scratch: s5) against the same at 4 (s4) and the tree as it is (h0), same box, twice
O=gpurun_out/r06_zu; mkdir -p $O
for R in 1 2; do python tools/tag_bench.py --variants h0,s4,s5 2>>$O/tag.err | tee -a $O/tag_bench.jsonl | cut -c1-420; done
