# round 6 ("decide phase D"): vpt_tokenize_batch with every chunk as predict + the flat writer (chained through device words like the fused
# one; VPT_TOKENIZE_FLAT=1) against the scoring kernel with the writer fused in (default): end to end, 100 K and 1 M lines, several chunk sizes; a stress loop
O=gpurun_out/r06_k; mkdir -p $O
for E in "" "VPT_TOKENIZE_FLAT=1" "VPT_TOKENIZE_FLAT=1 VPT_TOKENIZE_CHUNK_BYTES=2000000" "VPT_TOKENIZE_FLAT=1 VPT_TOKENIZE_CHUNK_BYTES=5000000" "VPT_TOKENIZE_FLAT=1 VPT_TOKENIZE_CHUNK_BYTES=7000000" "" "VPT_TOKENIZE_FLAT=1"; do
  env $E python tools/tokenize_bench.py --iters 25 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
for E in "" "VPT_TOKENIZE_FLAT=1" "VPT_TOKENIZE_FLAT=1 VPT_TOKENIZE_CHUNK_BYTES=16000000"; do
  env $E python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
python tools/tokenize_bench.py --config 4 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
cat $O/tokenize.jsonl | cut -c1-330
VPT_TOKENIZE_FLAT=1 python tools/tokenize_stress.py --iters 200 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | tee $O/stress.jsonl
