# round 4: the tagged writer (emit_fused_kernel<tags>: 135 VGPRs, 3 waves per SIMD) compiled for 4 / 5 / 6 waves per SIMD (128 / 96 / 80 VGPRs with
# 176 / 304 / 384 bytes of scratch per lane): configs[4]'s emit (1 M ragged sentences, 444 MB of tagged text), parity in every line
O=gpurun_out/r04_u; mkdir -p $O
for E in "X=1" "VPT_EMIT_TAG_OCC=4" "VPT_EMIT_TAG_OCC=5" "VPT_EMIT_TAG_OCC=6"; do
  env $E python bench.py --config 4 --steps 10 --warmup 3 --no-e2e 2>/dev/null | tail -1 > $O/b.json
  python - "$E" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_u/b.json").read())
line = {"env": sys.argv[1], "emit_ms": round(d["emit"]["ms_per_step"], 4), "emit_parity": d["emit"].get("parity"), "step_ms": round(d["ms_per_step"], 4), "parity": d["parity"]}
print(json.dumps(line))
open("gpurun_out/r04_u/tagged_writer.jsonl", "a").write(json.dumps(line) + "\n")
PY
done
