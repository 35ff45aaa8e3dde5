"""The untagged writer (vpt_write_tokenized_batch_device) on the GPU box, A/B over prebuilt libraries in ONE process:
    python tools/writer_bench.py [--variants new,name,..] [--configs 1,2] [--steps 20]
Per library and workload: the HIP-event median of the call on the labels the library's own predict left on the device, the bytes it moves (text + labels in,
text + offsets out) against that time, and whether text and offsets equal the oracle's writer on the same labels (oracle/vaporetto_oracle.c, the checker:
sentence.rs:850-886).  One JSON line per (variant, workload)."""
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
    ap.add_argument("--configs", default="1,2")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--per-block", default="0", help="comma list: sentences per run (VPT_EMIT_PER_BLOCK; 0: what the library takes)")
    args = ap.parse_args()
    import torch
    import bench
    from oracle import cbind
    from vaporetto_amd import _lib, api
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    ncores = os.cpu_count() or 1
    stream = torch.cuda.current_stream().cuda_stream
    for cid in [int(c) for c in args.configs.split(",")]:
        cfg = bench.CONFIGS[cid]
        raw, name = bench.load_model_bytes(cfg["kind"], 1.0)
        utf8, boff, ooff, _, S = bench.make_shard(cfg, raw, 0, 1, ncores, 0)
        nb, nbytes = int(ooff[-1]), int(boff[-1])
        d_text = torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev)
        d_boff = torch.from_numpy(boff.astype(np.int64)).to(dev)
        d_ooff = torch.from_numpy(ooff.astype(np.int64)).to(dev)
        max_bytes = int(np.max(np.diff(boff.astype(np.int64))))
        max_chars = int(np.max(np.diff(ooff.astype(np.int64)))) + 1
        want = None
        for v, pb in [(v, pb) for v in args.variants.split(",") for pb in args.per_block.split(",")]:
            os.environ["VPT_EMIT_PER_BLOCK"] = pb
            L = C.CDLL(lib_path(v))
            for fn, (res, a) in _lib.SIGNATURES.items():
                f = getattr(L, fn)
                f.restype, f.argtypes = res, a
            _lib._lib = L
            pred = api.Predictor(api.Model.read_slice(raw)[0], False, device=0)
            batch = api.DeviceBatch(pred)
            batch.set_max_sentence_chars(max_chars)
            d_scores = torch.empty(nb + 1, dtype=torch.int32, device=dev)
            d_labels = torch.empty(nb + 16, dtype=torch.uint8, device=dev)
            batch.predict(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, max_bytes, d_scores.data_ptr(), d_labels.data_ptr(), stream)
            batch.sync()
            cap = 3 * nbytes + 64
            d_out = torch.full((cap + 1,), 0xEE, dtype=torch.uint8, device=dev)
            d_toff = torch.empty(S + 1, dtype=torch.int64, device=dev)

            def emit():
                batch.write_tokenized(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(), d_out.data_ptr(), cap, d_toff.data_ptr(), stream)
            for _ in range(3):
                emit()
            batch.sync()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for a, b in ev:
                a.record(); emit(); b.record()
            torch.cuda.synchronize()
            batch.sync()
            ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
            toff = d_toff.cpu().numpy().astype(np.uint64)
            out_bytes = int(toff[-1])
            moved = nbytes + nb + out_bytes + 16 * S
            row = {"variant": v, "per_block": int(pb), "workload": cfg["name"], "model": name, "ms": round(ms, 4), "GBps": round(moved / ms / 1e6, 1), "frac_of_hbm": round(moved / ms / 1e6 / 8000.0, 4),
                   "bytes_moved": moved, "out_bytes": out_bytes}
            if not args.no_parity:
                if want is None:   # (the labels are the same for every library: the scoring kernel is checked elsewhere)
                    orc = cbind.OraclePredictor(raw, False)
                    labels = d_labels[:nb].cpu().numpy()
                    want = orc.write_tokenized_batch(utf8, boff, ooff, labels, None, None, nthreads=ncores)
                got = d_out[:out_bytes].cpu().numpy()
                row["parity"] = bool(np.array_equal(toff, want[1]) and np.array_equal(got, want[0]))
            print(json.dumps(row), flush=True)
            del d_out, d_toff, d_scores, d_labels, batch, pred
        del d_text, d_boff, d_ooff


if __name__ == "__main__":
    main()
