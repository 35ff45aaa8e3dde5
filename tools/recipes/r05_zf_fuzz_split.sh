# round 5: the fuzz (random models x ragged batches x flags against the oracle: boundaries, tags, writers, vpt_tokenize_batch) on the device with fill_tags always as
# two launches, so that every batch goes through the rewritten front end (small batches otherwise take the one-launch kernel)
O=gpurun_out/r05_zf; mkdir -p $O
( VPT_TAG_SPLIT=1 VPT_FUZZ_SEED0=700000 timeout 100 python tools/fuzz_gpu.py ${1:-55} 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee $O/fuzz_gpu_split.log
