# round 4: a long fuzz of the final tree -- random models x ragged batches x flags against the oracle (boundaries, tags, writers, tokenize, predict_write);
# a third of the time with fill_tags forced through the two-launch path (the flat front end) and tokenize cut into small chunks
O=gpurun_out/r04_r; mkdir -p $O
( VPT_FUZZ_SEED0=100000 timeout 260 python tools/fuzz_gpu.py 220 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_a.log; cat $O/fuzz_a.log
( VPT_TAG_SPLIT=1 VPT_TOKENIZE_CHUNK_BYTES=700 VPT_FUZZ_SEED0=200000 timeout 200 python tools/fuzz_gpu.py 160 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_b.log; cat $O/fuzz_b.log
python tools/tokenize_stress.py --iters 400 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 > $O/stress.json; cat $O/stress.json
