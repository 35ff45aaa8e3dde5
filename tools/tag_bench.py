"""fill_tags and the tagged writer on the GPU box, A/B over prebuilt libraries in ONE process (python tools/tag_bench.py [--variants new,name,..]):
BASELINE configs[4]'s batch (tag models on, 8 .. 512 chars) with the oracle's labels; per library the HIP-event medians of vpt_fill_tags_batch_device
(records only / with the dense array), vpt_write_tagged_batch_device and vpt_write_tokenized_batch_device on the same batch, and whether tags and tagged
text equal the oracle's (timing ablations, tools/build_variants.sh -DVPT_TAG_ABLATE / -DVPT_EMIT_ABLATE, are expected to differ).  One JSON line per variant."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lib_path(name):
    return os.path.join(ROOT, "vaporetto_amd", "lib", "libvaporetto_hip.so") if name == "new" else os.path.join(ROOT, "tools", "prebuilt", "libvaporetto_%s.so" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="new")
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--sentences", type=int, default=None)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import bench
    from oracle import cbind
    from vaporetto_amd import _lib, api
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    raw, name = bench.load_model_bytes(cfg["kind"], 1.0)
    ncores = os.cpu_count() or 1
    utf8, boff, ooff, _, S = bench.make_shard(cfg, raw, 0, 1, ncores, args.sentences)
    nb, nbytes = int(ooff[-1]), int(boff[-1])
    orc = cbind.OraclePredictor(raw, True)
    _, o_labels, _, _ = orc.predict_batch(utf8, boff, nthreads=ncores)
    o_tags, _, o_models = orc.fill_tags_batch(utf8, boff, ooff, o_labels, nthreads=ncores, want_scores=False)
    o_text, o_toff = orc.write_tokenized_batch(utf8, boff, ooff, o_labels, o_tags, o_models, nthreads=ncores)
    d_text = torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev)
    d_boff = torch.from_numpy(boff.astype(np.int64)).to(dev)
    d_ooff = torch.from_numpy(ooff.astype(np.int64)).to(dev)
    d_labels = torch.from_numpy(np.concatenate([o_labels, np.zeros(16, np.uint8)])).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for v in args.variants.split(","):
        L = C.CDLL(lib_path(v))
        for fn, (res, a) in _lib.SIGNATURES.items():
            f = getattr(L, fn)
            f.restype, f.argtypes = res, a
        _lib._lib = L
        pred = api.Predictor(api.Model.read_slice(raw)[0], True, device=0)
        nt = pred.n_tags()
        cap = 3 * nbytes + 64 + nbytes * pred.max_tag_suffix()
        d_out = torch.empty(cap + 1, dtype=torch.uint8, device=dev)
        d_toff = torch.empty(S + 1, dtype=torch.int64, device=dev)
        d_tags = torch.empty((nb + S) * nt + 1, dtype=torch.int32, device=dev)
        batch = api.DeviceBatch(pred)
        P = (d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr())

        def timed(fn):
            for _ in range(3):
                fn()
            try:
                batch.sync()
            except api.VaporettoError:   # (an ablation whose output does not add up says so)
                pass
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            try:
                batch.sync()
            except api.VaporettoError:   # (an ablation whose output does not add up says so)
                pass
            return round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4)
        row = {"variant": v, "workload": cfg["name"], "chars": nb + S}
        d_scores = torch.empty(nb + 1, dtype=torch.int32, device=dev)
        d_lab2 = torch.empty(nb + 16, dtype=torch.uint8, device=dev)
        max_bytes = int(np.max(np.diff(boff.astype(np.int64))))
        P2 = (d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, max_bytes, d_scores.data_ptr(), d_lab2.data_ptr(), stream)
        row["predict_ms"] = timed(lambda: batch.predict(*P2))

        def both():   # Sentence::fill_tags behind Predictor::predict: the scoring kernel leaves the chars, no decode launch
            batch.predict(*P2)
            batch.fill_tags(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_lab2.data_ptr(), 0, stream)
        row["predict_then_fill_tags_ms"] = timed(both)
        row["fill_tags_behind_predict_ms"] = round(row["predict_then_fill_tags_ms"] - row["predict_ms"], 4)
        del d_scores, d_lab2
        row["fill_tags_dense_ms"] = timed(lambda: batch.fill_tags(*P, d_tags.data_ptr(), stream))
        row["tags_parity"] = bool(np.array_equal(d_tags[:(nb + S) * nt].cpu().numpy().reshape(nb + S, nt), o_tags))
        row["fill_tags_records_ms"] = timed(lambda: batch.fill_tags(*P, 0, stream))
        row["write_tagged_ms"] = timed(lambda: batch.write_tagged(*P, 0, d_out.data_ptr(), cap, d_toff.data_ptr(), stream))
        toff = d_toff.cpu().numpy().astype(np.uint64)
        n_out = int(min(toff[-1], cap))
        row["tagged_text_parity"] = bool(np.array_equal(toff, o_toff) and np.array_equal(d_out[:n_out].cpu().numpy(), o_text))
        row["tagged_out_bytes"] = int(toff[-1])
        row["write_untagged_ms"] = timed(lambda: batch.write_tokenized(*P, d_out.data_ptr(), cap, d_toff.data_ptr(), stream))
        print(json.dumps(row), flush=True)
        del batch, pred, d_out, d_toff, d_tags


if __name__ == "__main__":
    main()
