# round 4: vpt_tokenize_batch with tapered chunks (first and last a third of the middle ones) and the status words read through pinned memory,
# against chunks of one size (VPT_TOKENIZE_EQUAL_CHUNKS=1) and the one-stream schedule (VPT_TOKENIZE_SERIAL=1).  Parity, A/B, chunk sizes, stress.
O=gpurun_out/r04_n; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "tokenize or predict_and_write or flat_kernel or pinned" 2>&1 | tail -3 | tee $O/tests.log
for E in "" "VPT_TOKENIZE_EQUAL_CHUNKS=1" "VPT_TOKENIZE_SERIAL=1" "VPT_TOKENIZE_CHUNK_BYTES=2400000" "VPT_TOKENIZE_CHUNK_BYTES=2800000" "VPT_TOKENIZE_CHUNK_BYTES=3900000" "VPT_TOKENIZE_CHUNK_BYTES=4800000"; do
  env $E python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
for E in "" "VPT_TOKENIZE_EQUAL_CHUNKS=1" "VPT_TOKENIZE_SERIAL=1" "VPT_TOKENIZE_CHUNK_BYTES=6000000" "VPT_TOKENIZE_CHUNK_BYTES=12000000"; do
  env $E python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
cat $O/tokenize.jsonl | cut -c1-400
python tools/tokenize_stress.py --iters 250 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500 > $O/stress.json; cat $O/stress.json
