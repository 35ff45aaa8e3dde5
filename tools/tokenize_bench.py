"""vpt_tokenize_batch end to end on the GPU box (lines in, tokenized lines out over PCIe, pinned buffers):
python tools/tokenize_bench.py [--config 1|4] [--repeat R] -- one JSON line.  VPT_TOKENIZE_CHUNK_BYTES is read by the library."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaporetto_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--iters", type=int, default=15)
args = ap.parse_args()
tagged = args.config == 4
raw = synth.synth_model(3 if tagged else 1)
if tagged:
    utf8, boff = synth.synth_sentences(raw, 100000, 8, 512, seed=synth.SEED_BASE + 2)
else:
    utf8, boff = synth.synth_sentences(raw, 100000, 64, 64, seed=synth.SEED_BASE + 2)
pred = api.Predictor(api.Model.read_slice(raw)[0], tagged)
R = args.repeat
if R > 1:
    n1 = len(utf8)
    boff = np.concatenate([boff[:-1] + np.uint64(k * n1) for k in range(R)] + [np.array([R * n1], dtype=np.uint64)])
    utf8 = np.tile(utf8, R)
S = len(boff) - 1
cap = 3 * len(utf8) + (len(utf8) * pred.max_tag_suffix() if tagged else 0)
keep = [api.PinnedArray((len(utf8),), np.uint8), api.PinnedArray((cap,), np.uint8), api.PinnedArray((S + 1,), np.uint64)]
keep[0].array[:] = utf8
ref_text, ref_off = pred.tokenize_packed(utf8, boff, tagged=tagged)
ref_text = ref_text.copy(); ref_off = ref_off.copy()
for _ in range(3):
    t, o = pred.tokenize_packed(keep[0].array, boff, tagged=tagged, text_out=keep[1].array, offsets_out=keep[2].array)
ts = []
for _ in range(args.iters):
    t0 = time.perf_counter()
    t, o = pred.tokenize_packed(keep[0].array, boff, tagged=tagged, text_out=keep[1].array, offsets_out=keep[2].array)
    ts.append(time.perf_counter() - t0)
dt = float(np.median(ts))
ooff = api.count_boundaries(utf8, boff)
chars = int(ooff[-1]) + S
print(json.dumps({"config": args.config, "sentences": S, "ms_per_batch": round(1e3 * dt, 4), "ms_min": round(1e3 * min(ts), 4), "G_chars_per_s": round(chars / dt / 1e9, 3),
                  "h2d_GBps": round(len(utf8) / dt / 1e9, 2), "d2h_GBps": round(len(t) / dt / 1e9, 2), "same": bool(np.array_equal(t, ref_text) and np.array_equal(o, ref_off)),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("VPT_TOKENIZE") and v}}))
