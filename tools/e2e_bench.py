"""End-to-end timing of vpt_predict_batch over PCIe (pinned caller buffers) on the GPU box:
python tools/e2e_bench.py [--repeat R] [--labels-only] -- one JSON line.  configs[1]'s batch (100 K x 64 chars), tiled R
times into one batch of R x 100 K sentences.  Chunking knobs are read by the library from the environment
(VPT_CHUNK_CHARS, VPT_CHUNK_FIRST, VPT_CHUNK_GROWTH_PCT)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaporetto_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--labels-only", action="store_true")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
raw = synth.synth_model(1)
utf8, boff = synth.synth_sentences(raw, 100000, 64, 64, seed=synth.SEED_BASE + 2)
pred = api.Predictor(api.Model.read_slice(raw)[0], False)
ref_scores, ref_labels, ooff1 = pred.predict_packed(utf8, boff)
R = args.repeat
if R > 1:
    n1 = len(utf8)
    boff = np.concatenate([boff[:-1] + np.uint64(k * n1) for k in range(R)] + [np.array([R * n1], dtype=np.uint64)])
    utf8 = np.tile(utf8, R)
ooff = api.count_boundaries(utf8, boff)
nb, S = int(ooff[-1]), len(boff) - 1
keep = [api.PinnedArray((len(utf8),), np.uint8), api.PinnedArray((nb,), np.int32), api.PinnedArray((nb,), np.uint8)]
keep[0].array[:] = utf8
scores = None if args.labels_only else keep[1].array


def once():
    api.predict_packed_sharded([pred], keep[0].array, boff, out_offsets=ooff, scores=scores, labels=keep[2].array, want_scores=not args.labels_only)


for _ in range(3):
    once()
ts = []
for _ in range(args.iters):
    t0 = time.perf_counter()
    once()
    ts.append(time.perf_counter() - t0)
dt = float(np.median(ts))
ok = bool(np.array_equal(keep[2].array[:nb], np.tile(ref_labels, R)) and (args.labels_only or np.array_equal(keep[1].array[:nb], np.tile(ref_scores, R))))
bytes_in, bytes_out = len(utf8) + 16 * (S + 1), (1 if args.labels_only else 5) * nb
print(json.dumps({"sentences": S, "labels_only": args.labels_only, "ms_per_batch": round(1e3 * dt, 4), "ms_min": round(1e3 * min(ts), 4),
                  "G_boundaries_per_s": round(nb / dt / 1e9, 3), "h2d_GBps": round(bytes_in / dt / 1e9, 2), "d2h_GBps": round(bytes_out / dt / 1e9, 2),
                  "parity": ok, "env": {k: v for k, v in os.environ.items() if k.startswith("VPT_CHUNK")}}))
