# round 4: the scoring kernel with and without the writer's phase over batches of 4 K .. 100 K sentences of configs[1]'s text (device resident):
# what a chunk of vpt_tokenize_batch costs as a launch of its own (r04_e timeline: 88 us per 16.7 K sentences, 206 us per 100 K).
O=gpurun_out/r04_l; mkdir -p $O
for S in 4000 8000 16700 33400 50000 100000; do
  python bench.py --config 1 --sentences $S --steps 20 --warmup 5 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 > $O/b_$S.json
  python - $S <<'PY'
import json, sys
S = sys.argv[1]
d = json.loads(open("gpurun_out/r04_l/b_%s.json" % S).read())
f = d["emit"]["fused"]
line = {"sentences": int(S), "tiles": d["config"].get("tiles"), "kernel_ms": round(d["roofline"]["kernel_ms"], 4), "step_ms": round(d["ms_per_step"], 4),
        "fused_kernel_ms": round(f["kernel_ms"], 4), "fused_step_ms": round(f["ms_per_step"], 4), "writer_ms": round(d["emit"]["ms_per_step"], 4)}
print(json.dumps(line))
open("gpurun_out/r04_l/by_size.jsonl", "a").write(json.dumps(line) + "\n")
PY
done
