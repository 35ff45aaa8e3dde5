# round 5: run sizes of the flat writer on the big batches: tagged configs[4] at 10 K / 20 K / 40 K chars per run, untagged configs[2] (10 M sentences) at 5 K / 10 K / 20 K
O=gpurun_out/r05_k; mkdir -p $O
python tools/emit_bench.py --config 4 --steps 15 --env "VPT_EMIT_RUN_CHARS=10240" "VPT_EMIT_RUN_CHARS=20480" "VPT_EMIT_RUN_CHARS=40960" 2>/dev/null > $O/emit_c4.jsonl; cat $O/emit_c4.jsonl
python tools/emit_bench.py --config 2 --steps 10 --env "" "VPT_EMIT_RUN_CHARS=10240" "VPT_EMIT_RUN_CHARS=20480" "VPT_EMIT_WAVE_BLOCKS=1" 2>/dev/null > $O/emit_c2.jsonl; cat $O/emit_c2.jsonl
