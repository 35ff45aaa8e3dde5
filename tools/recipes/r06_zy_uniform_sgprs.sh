# round 6: the writer's workgroup-wide values (offsets read back from LDS, prefix-sum totals, the carried record) in scalar registers: tagged instance 115 -> 107 VGPRs
# (n4; n5 = held to 5 waves per SIMD, 32 bytes of scratch), untagged 64 -> 55 -- against the tree before (head), same box, twice
O=gpurun_out/r06_zy; mkdir -p $O
for R in 1 2; do python tools/tag_bench.py --variants head,n4,n5 2>>$O/tag.err | tee -a $O/tag_bench.jsonl | cut -c1-420; done
for R in 1 2; do python tools/writer_bench.py --variants head,n4 --configs 1,2,5 --no-parity 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-200; done
