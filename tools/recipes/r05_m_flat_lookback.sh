# round 5: the flat writer's look-back four waves wide + one generation of workgroups on small batches (w2 = the sources) against w0
O=gpurun_out/r05_m; mkdir -p $O
for V in r05w0 r05w2; do
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  echo "== $V" | tee -a $O/emit.jsonl
  python tools/emit_bench.py --config 1 --env "" "VPT_EMIT_RUN_CHARS=5120" 2>/dev/null | tee -a $O/emit.jsonl | cut -c1-160
  python tools/emit_bench.py --config 5 --env "" 2>/dev/null | tee -a $O/emit.jsonl | cut -c1-160
  python tools/emit_bench.py --config 2 --steps 10 --env "" 2>/dev/null | tee -a $O/emit.jsonl | cut -c1-160
  python tools/emit_bench.py --config 4 --steps 10 --env "" 2>/dev/null | tee -a $O/emit.jsonl | cut -c1-160
done
cp tools/prebuilt/libvaporetto_r05w2.so vaporetto_amd/lib/libvaporetto_hip.so
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "write or writer or tokenize or emit" 2>&1 | tail -3 ) | tee $O/gpu_writer_tests.log
