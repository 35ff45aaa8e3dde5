"""Randomised GPU-vs-oracle sweep (run on the GPU box from the repo root: `python tools/fuzz_gpu.py [seconds]`;
with VPT_FUZZ_EMULATED=1 the kernels run on the CPU emulator of tests/native/hipemu instead -- slower, no GPU needed).
Random models (alphabet size, pattern counts, windows, word lengths, wide weights, tag models) and random ragged
batches with the label post-filters / fullwidth flag toggled; stops at the first mismatch."""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cbind  # noqa: E402
from tests import randmodel  # noqa: E402
from vaporetto_amd import api  # noqa: E402
from vaporetto_amd.modelfmt import encode_model  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
if os.environ.get("VPT_FUZZ_EMULATED"):   # the same sweep over the kernel sources on the CPU emulator (tests/emu.py)
    from tests import emu  # noqa: E402
    from vaporetto_amd import _lib  # noqa: E402
    _lib._lib = emu.load()
t_end = time.time() + budget
seed = int(os.environ.get("VPT_FUZZ_SEED0", "0"))   # first seed - 1: a later run continues where an earlier one stopped
n_models = n_sent = 0
f = api.KyteaFullwidthFilter()
while time.time() < t_end:
    seed += 1
    rng = random.Random(seed)
    alpha = [chr(c) for c in range(0x3041, 0x3041 + rng.choice([3, 6, 12, 30]))] + rng.sample(list("漢字AZ09az-.、。ｱ￾￿𠮷𩸽🤌 /\\"), rng.randint(0, 6))
    # windows: what the distributed models have (3) half the time, anything up to 8 otherwise (round 4: every window on the packed tables)
    m = randmodel.rand_model(9000 + seed, alphabet=alpha, wc=rng.choice([3, 3, 3, 3, 2, 1, 4, 5, 6, 7, 8]), wt=rng.choice([1, 2, 3, 3, 3, 4, 5, 6, 8]),
                             max_n=rng.choice([3, 3, 4, 6]), n_char=rng.choice([20, 200, 1500]), n_dict=rng.choice([0, 30, 400, 3000]), n_type=rng.choice([0, 10, 80]),
                             max_word=rng.choice([2, 5, 9, 15]), big=(seed % 7 == 0), n_tag_models=rng.choice([0, 0, 10]))
    raw = encode_model(m)
    tags = bool(m.tag_models) and seed % 2 == 0
    if seed % 3 == 1:   # vpt_tokenize_batch in many chunks (the knob is read when the predictor is made)
        os.environ["VPT_TOKENIZE_CHUNK_BYTES"] = str(rng.choice([300, 2000, 20000]))
    else:
        os.environ.pop("VPT_TOKENIZE_CHUNK_BYTES", None)
    pred = api.Predictor(api.Model.read_slice(raw)[0], tags)
    orc = cbind.OraclePredictor(raw, tags)
    texts = randmodel.rand_sentences(seed, m, rng.choice([50, 800]), alphabet=alpha, max_len=rng.choice([5, 60, 400]))
    if seed % 5 == 0:
        texts.append("".join(rng.choice(alpha) for _ in range(rng.choice([1700, 5000]))))   # longer than a tile
    fw = seed % 3 == 0
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    scores, labels, ooff = pred.predict_packed(utf8, boff, fullwidth=fw)
    n_utf8, n_boff = api.pack_texts([(f.filter(t) if fw else t).encode("utf-8") for t in texts])
    o_scores, o_labels, o_ooff, _ = orc.predict_batch(n_utf8, n_boff)
    if not (np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels) and np.array_equal(ooff, o_ooff)):
        print("MISMATCH boundary scores: seed", seed, "info", pred.info())
        sys.exit(1)
    if seed % 4 == 0:   # token emission (vpt_write_tokenized_batch) against the restated writer
        ttext, toff = pred.write_tokenized_packed(utf8, boff, ooff, labels)
        tb = bytes(ttext)
        for i, t in enumerate(texts[:60]):
            lab = labels[int(ooff[i]):int(ooff[i + 1])]
            out, tok = [], []
            for k, c in enumerate(t):
                tok.append("\\" + c if c in " \\/" else c)
                if k == len(t) - 1 or lab[k] == 1:
                    out.append("".join(tok)); tok = []
            if tb[int(toff[i]):int(toff[i + 1])].decode("utf-8") != " ".join(out):
                print("MISMATCH tokenized text: seed", seed, repr(t[:60]))
                sys.exit(1)
    if tags:
        got = pred.fill_tags_packed(utf8, boff, ooff, labels, fullwidth=fw)
        for i, t in enumerate(texts[:40]):
            a, b = int(ooff[i]), int(ooff[i + 1])
            want, _ = orc.predict_tags(f.filter(t) if fw else t, labels=labels[a:b])
            if not np.array_equal(got[a + i:a + i + len(t)], want):
                print("MISMATCH tags: seed", seed, repr(t[:60]))
                sys.exit(1)
    if tags and not fw and seed % 4 == 2:   # the whole tagged writer against the mirror's (fill_tags + write_tokenized_text)
        sents = [api.Sentence.from_raw(t) for t in texts[:40]]
        pred.predict_batch(sents)
        got_t = pred.write_tokenized_batch(sents, tagged=True)
        pred.fill_tags_batch(sents)
        if got_t != [s.write_tokenized_text() for s in sents]:
            print("MISMATCH tagged text: seed", seed)
            sys.exit(1)
    if seed % 4 == 1:   # the one-call pipeline (vpt_tokenize_batch) against predict_batch + the packed writer
        sub = texts[:60]
        sents = [api.Sentence.from_raw(t) for t in sub]
        pred.predict_batch(sents, fullwidth=fw)
        if pred.tokenize(sub, fullwidth=fw) != pred.write_tokenized_batch(sents):
            print("MISMATCH tokenize: seed", seed)
            sys.exit(1)
    if not fw and seed % 2 == 1:   # predict + the writer for the WHOLE batch against the oracle's writer:
        o_text, o_toff = orc.write_tokenized_batch(n_utf8, n_boff, o_ooff, o_labels, None, None)
        t_text, t_toff = pred.tokenize_packed(utf8, boff)       # ... through vpt_tokenize_batch (chunks chained into one text)
        if not (np.array_equal(t_toff, o_toff) and np.array_equal(t_text, o_text)):
            print("MISMATCH tokenize (whole batch): seed", seed, "info", pred.info())
            sys.exit(1)
        if not os.environ.get("VPT_FUZZ_EMULATED"):               # ... and through the device entry point
            import torch
            S, nb = len(texts), int(ooff[-1])
            d = [torch.from_numpy(np.concatenate([utf8, np.zeros(32, np.uint8)])).cuda(), torch.from_numpy(boff.astype(np.int64)).cuda(), torch.from_numpy(ooff.astype(np.int64)).cuda()]
            cap = 3 * len(utf8) + 16
            d_out, d_toff = torch.zeros(cap + 16, dtype=torch.uint8, device="cuda"), torch.zeros(S + 1, dtype=torch.int64, device="cuda")
            d_sc, d_lb = torch.zeros(nb + 1, dtype=torch.int32, device="cuda"), torch.zeros(nb + 1, dtype=torch.uint8, device="cuda")
            batch = api.DeviceBatch(pred)
            batch.predict_write(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), S, nb, int(np.max(np.diff(boff.astype(np.int64)))), d_sc.data_ptr(), d_lb.data_ptr(),
                                d_out.data_ptr(), cap, d_toff.data_ptr(), torch.cuda.current_stream().cuda_stream)
            batch.sync()
            g_toff = d_toff.cpu().numpy().astype(np.uint64)
            if not (np.array_equal(g_toff, o_toff) and np.array_equal(d_out[:int(o_toff[-1])].cpu().numpy(), o_text) and np.array_equal(d_sc[:nb].cpu().numpy(), o_scores)):
                print("MISMATCH predict_write: seed", seed, "info", pred.info())
                sys.exit(1)
    n_models += 1
    n_sent += len(texts)
print("fuzz ok: %d models (seeds up to %d), %d sentences, no mismatch" % (n_models, seed, n_sent))
