"""vpt_tokenize_batch over and over into pinned buffers, every output compared with the first (diagnostics of the chunked / direct paths):
python tools/tokenize_stress.py [--iters N]   (VPT_TOKENIZE_CHUNK_BYTES / VPT_TOKENIZE_NO_DIRECT are read by the library)"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaporetto_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--sentences", type=int, default=100000)
args = ap.parse_args()
raw = synth.synth_model(1)
utf8, boff = synth.synth_sentences(raw, args.sentences, 64, 64, seed=synth.SEED_BASE + 2)
pred = api.Predictor(api.Model.read_slice(raw)[0], False)
S = len(boff) - 1
cap = 3 * len(utf8)
os.environ_backup = dict(os.environ)
ref_text, ref_off = pred.tokenize_packed(utf8, boff)
ref_text, ref_off = ref_text.copy(), ref_off.copy()
keep = [api.PinnedArray((len(utf8),), np.uint8), api.PinnedArray((cap,), np.uint8), api.PinnedArray((S + 1,), np.uint64)]
keep[0].array[:] = utf8
bad = []
for it in range(args.iters):
    keep[1].array[:len(ref_text) + 64] = 0xEE
    keep[2].array[:] = 0
    t, o = pred.tokenize_packed(keep[0].array, boff, text_out=keep[1].array, offsets_out=keep[2].array)
    if not (np.array_equal(o, ref_off) and np.array_equal(t, ref_text)):
        d_off = np.flatnonzero(o != ref_off)
        n = min(len(t), len(ref_text))
        d_txt = np.flatnonzero(t[:n] != ref_text[:n])
        first = int(d_txt[0]) if len(d_txt) else -1
        sent = int(np.searchsorted(ref_off, first, side="right") - 1) if first >= 0 else -1
        bad.append({"iter": it, "offsets_differing": int(len(d_off)), "first_offset": int(d_off[0]) if len(d_off) else -1, "text_bytes_differing": int(len(d_txt)),
                    "first_text_byte": first, "in_sentence": sent, "last_text_byte": int(d_txt[-1]) if len(d_txt) else -1,
                    "got": bytes(t[first:first + 24]).hex() if first >= 0 else "", "want": bytes(ref_text[first:first + 24]).hex() if first >= 0 else "",
                    "len_got": int(len(t)), "len_want": int(len(ref_text))})
        if len(bad) >= 5:
            break
print(json.dumps({"iters": args.iters, "mismatches": len(bad), "details": bad, "env": {k: v for k, v in os.environ.items() if k.startswith("VPT_TOKENIZE") and v}}))
