# round 6: the same box, the same minute: the writer with a dword's bytes and spaces picked by v_perm_b32 (sp) against the bytes stored one by one (nosp), twice
O=gpurun_out/r06_v; mkdir -p $O
for R in 1 2; do python tools/writer_bench.py --variants sp,nosp --configs 1,2 --no-parity 2>>$O/bench.err | tee -a $O/writer_ab.jsonl | cut -c1-200; done
