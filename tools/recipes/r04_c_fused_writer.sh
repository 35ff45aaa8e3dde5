# round 4, third call: the writer fused into the scoring kernel -- parity suite, bench of configs[1] / [3] / documents (fused vs predict + emit),
# vpt_tokenize_batch end to end over chunk sizes, with the kernels writing into the pinned output and with the copy-out path
O=gpurun_out/r04_c; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/gpu_tests.log
python bench.py --config 1 --steps 30 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_c1.json
python bench.py --config 3 --steps 20 --warmup 3 --no-e2e 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_c3.json
python bench.py --config 5 --steps 10 --warmup 3 --no-e2e 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_docs.json
for C in "" 2000000 4000000 8000000 1000000000; do
  VPT_TOKENIZE_CHUNK_BYTES=$C python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
VPT_TOKENIZE_NO_DIRECT=1 python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
VPT_TOKENIZE_NO_DIRECT=1 VPT_TOKENIZE_CHUNK_BYTES=1000000000 python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
for C in "" 8000000 32000000 1000000000; do
  VPT_TOKENIZE_CHUNK_BYTES=$C python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
VPT_TOKENIZE_NO_DIRECT=1 python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
cat $O/tokenize.jsonl
