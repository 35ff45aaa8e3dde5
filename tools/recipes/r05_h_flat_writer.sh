# round 5: the untagged writer as a workgroup per run of sentences (emit_flat_kernel) against the wave-per-block kernel it replaces, configs[1] and the documents
# workload, run sizes 2.5 K / 5 K / 10 K / 20 K chars; then the GPU parity suite on the new sources (scoring kernel diet + flat writer + dist paths)
O=gpurun_out/r05_h; mkdir -p $O
python tools/emit_bench.py --config 1 --env "" "VPT_EMIT_WAVE_BLOCKS=1" "VPT_EMIT_RUN_CHARS=2560" "VPT_EMIT_RUN_CHARS=10240" "VPT_EMIT_RUN_CHARS=20480" 2>/dev/null > $O/emit_c1.jsonl; cat $O/emit_c1.jsonl
python tools/emit_bench.py --config 5 --env "" "VPT_EMIT_WAVE_BLOCKS=1" 2>/dev/null > $O/emit_docs.jsonl; cat $O/emit_docs.jsonl
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
