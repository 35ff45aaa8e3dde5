# round 5: the TAGGED writer as a workgroup per run of sentences (emit_flat_kernel<true>) against the wave-per-block kernel (VPT_EMIT_WAVE_TAGGED), configs[4]:
# 1 M sentences of 8..512 chars, 444 MB of tagged text, byte for byte against the oracle's writer
O=gpurun_out/r05_j; mkdir -p $O
python tools/emit_bench.py --config 4 --steps 15 --env "" "VPT_EMIT_WAVE_TAGGED=1" "VPT_EMIT_RUN_CHARS=10240" "VPT_EMIT_RUN_CHARS=2560" 2>/dev/null > $O/emit_c4.jsonl; cat $O/emit_c4.jsonl
