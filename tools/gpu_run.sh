#!/bin/bash
set -u
O=gpurun_out/c45; mkdir -p $O
timeout 500 python tools/ab_bench.py --variants new,new:VPT_TILE_ROUNDS_MULT=2,new:VPT_TILE_ROUNDS_MULT=3,new:VPT_TILE_ROUNDS_MULT=4 --rounds 3 > $O/ab.jsonl 2> $O/ab.err; echo "ab rc=$?"; tail -2 $O/ab.err
python - <<'PY'
import json
for l in open("gpurun_out/c45/ab.jsonl"):
    try: d=json.loads(l)
    except Exception: continue
    print({k:d[k] for k in d if k in ("variant","kernel_ms","parity","tiles")})
PY
