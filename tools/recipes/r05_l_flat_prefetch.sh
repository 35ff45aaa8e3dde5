# round 5: the flat writer with the next piece's loads issued a piece ahead (w1: 74 / 140 VGPRs) against without (w0: 62 / 122), three workloads
O=gpurun_out/r05_l; mkdir -p $O
for V in r05w0 r05w1; do
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  echo "== $V" | tee -a $O/emit.jsonl
  python tools/emit_bench.py --config 1 --env "" 2>/dev/null | tee -a $O/emit.jsonl | cut -c1-160
  python tools/emit_bench.py --config 2 --steps 10 --env "" 2>/dev/null | tee -a $O/emit.jsonl | cut -c1-160
  python tools/emit_bench.py --config 4 --steps 10 --env "" 2>/dev/null | tee -a $O/emit.jsonl | cut -c1-160
done
