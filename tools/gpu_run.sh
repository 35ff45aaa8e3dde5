#!/bin/bash
O=gpurun_out/r03_zc; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests -m gpu -x -q -k "tag or write or tokenize" 2>&1 | tail -3 ) | tee $O/gpu_tests_tags.log
emit() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
t=d.get('tags') or {}
print(sys.argv[1], 'step ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'tags ms', t.get('ms_per_step'), 'parity', d.get('parity'), t.get('parity'))
PY
}
timeout 900 python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --quick --no-emit > $O/bench_c4_split.json 2> $O/bench_c4_split.err; emit $O/bench_c4_split.json
cp vaporetto_amd/lib/libvaporetto_hip.so /tmp/lib_keep.so
cp tools/prebuilt/libvaporetto_f8.so vaporetto_amd/lib/libvaporetto_hip.so
timeout 900 python bench.py --config 4 --steps 10 --warmup 3 --no-e2e --quick --no-emit --no-cpu-baseline > $O/bench_c4_f8.json 2> $O/bench_c4_f8.err; emit $O/bench_c4_f8.json
cp /tmp/lib_keep.so vaporetto_amd/lib/libvaporetto_hip.so
VPT_TAG_SPLIT=1 VPT_FUZZ_SEED0=41000 timeout 100 python tools/fuzz_gpu.py 45 2>&1 | tail -2 | tee $O/fuzz_split.log
