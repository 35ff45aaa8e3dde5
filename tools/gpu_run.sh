#!/bin/bash
O=gpurun_out/r03_za; mkdir -p $O
export TMPDIR=/tmp
emit() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
t=d.get('tags') or {}
print(sys.argv[1], 'step ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'tags ms', t.get('ms_per_step'))
PY
}
cp vaporetto_amd/lib/libvaporetto_hip.so /tmp/lib_keep.so
for v in keep np6 np8; do
  if [ $v = keep ]; then cp /tmp/lib_keep.so vaporetto_amd/lib/libvaporetto_hip.so; else cp tools/prebuilt/libvaporetto_$v.so vaporetto_amd/lib/libvaporetto_hip.so; fi
  for w in 32 64; do
  VPT_TAG_WGS_PER_CU=$w timeout 900 python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --quick --no-cpu-baseline --no-emit > $O/bench_c4_${v}_$w.json 2> $O/bench_c4_${v}_$w.err; emit $O/bench_c4_${v}_$w.json
  done
done
cp /tmp/lib_keep.so vaporetto_amd/lib/libvaporetto_hip.so
