import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VPT_EMIT_PER_BLOCK"] = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["VPT_TILE_FLAT"] = "64"; os.environ["VPT_FORCE_CUT_TILES"] = "1"; os.environ["VPT_TOKENIZE_CHUNK_BYTES"] = "20000"
import numpy as np
from oracle import cbind
from tests import randmodel
from vaporetto_amd import api
from vaporetto_amd.modelfmt import encode_model
m = randmodel.rand_model(853, alphabet="kana", wc=3, wt=3, n_char=80, n_dict=80, max_word=6)
raw = encode_model(m)
pred = api.Predictor(api.Model.read_slice(raw)[0], False); orc = cbind.OraclePredictor(raw)
rng = np.random.default_rng(4)
alphabet = randmodel.ALPHABETS["kana"][:12] + list("漢字 /\\aé🤌")
texts = ["".join(rng.choice(alphabet, size=int(n))) for n in list(rng.integers(1, 70, 500)) + [1, 1, 900, 2, 3000, 1]]
utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
S = len(texts)
o_scores, o_labels, ooff, _ = orc.predict_batch(utf8, boff)
o_text, o_toff = orc.write_tokenized_batch(utf8, boff, ooff, o_labels, None, None)
cap = 3 * len(utf8)
pin_text, pin_off = api.PinnedArray((cap,), np.uint8), api.PinnedArray((S + 1,), np.uint64)
L = api._lib.load()
def check(tag):
    text, toff = pred.tokenize_packed(utf8, boff, text_out=pin_text.array, offsets_out=pin_off.array)
    ok_o, ok_t = np.array_equal(toff, o_toff), np.array_equal(text, o_text)
    msg = ""
    if not ok_o:
        bad = np.flatnonzero(toff != o_toff); i = int(bad[0])
        msg += " %d offsets differ: sentences %s; got %s want %s; byte offsets of those sentences %s; sentences whose right offset is the wrong value: %s" % (
            len(bad), bad[:8].tolist(), toff[bad[:8]].tolist(), o_toff[bad[:8]].tolist(), boff[bad[:8]].tolist(), [int(np.flatnonzero(o_toff == v)[0]) if (o_toff == v).any() else None for v in toff[bad[:8]]])
    elif not ok_t:
        k = int(np.flatnonzero(text != o_text)[0]); i = int(np.searchsorted(o_toff, k, side="right") - 1); msg += " first text diff at byte %d (sentence %d)" % (k, i)
    print(tag, ok_o, ok_t, msg, flush=True)
for r in range(3):
    check("call %d" % r)
for out in [pin_text.array, np.zeros(cap, np.uint8)] * 6:
    st = L.vpt_tokenize_batch(pred.handle, utf8.ctypes.data, boff.ctypes.data, S, 0, 0, out.ctypes.data, 64, pin_off.array.ctypes.data)
    print("small capacity ->", st, api._lib.last_error()[:80], flush=True)
    check("after a small-capacity call")
    check("and again")
