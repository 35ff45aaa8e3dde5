"""A/B on the GPU box in ONE process (python tools/ab_bench.py [--model-kind K] [--variants a,b,..]): the bench
workload (100 K x 64-char sentences resident in HBM) through the built library and through prebuilt variants
(tools/prebuilt/*.so) or debug knobs, each checked bit for bit against the oracle.  Model, batch and oracle scores are
made once, so a variant costs one table compilation + its timed steps.  One line of JSON per variant."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {   # name -> (library, environment)
    "new": (os.path.join(ROOT, "vaporetto_amd", "lib", "libvaporetto_hip.so"), {}),
    "new_wg5": (os.path.join(ROOT, "vaporetto_amd", "lib", "libvaporetto_hip.so"), {"VPT_DEBUG_LDS_PAD": "5400"}),   # 5 workgroups per CU
    "r01z": (os.path.join(ROOT, "tools", "prebuilt", "libvaporetto_r01z.so"), {}),   # the kernel of profiles/r01_z_final_*
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="new,new_wg5,r01z")
    ap.add_argument("--model-kind", type=int, default=1)
    ap.add_argument("--sentences", type=int, default=100000)
    ap.add_argument("--min-len", type=int, default=64)
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    args = ap.parse_args()
    import torch
    from oracle import cbind
    from vaporetto_amd import _lib, api, synth
    import bench
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    raw, name = bench.load_model_bytes(args.model_kind, 1.0)
    utf8, boff = synth.synth_sentences(raw, args.sentences, args.min_len, args.max_len, seed=synth.SEED_BASE + 2)
    o_scores, o_labels, o_ooff, _ = cbind.OraclePredictor(raw).predict_batch(utf8, boff, nthreads=os.cpu_count() or 1)
    ooff = o_ooff
    S, nb = args.sentences, int(ooff[-1])
    max_bytes = int(np.max(np.diff(boff.astype(np.int64))))
    d_text = torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev)
    d_boff = torch.from_numpy(boff.astype(np.int64)).to(dev)
    d_ooff = torch.from_numpy(ooff.astype(np.int64)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for v in args.variants.split(","):
        path, env = VARIANTS[v]
        for k in ("VPT_DEBUG_LDS_PAD",):
            os.environ.pop(k, None)
        os.environ.update(env)
        L = C.CDLL(path)
        for fn_name, (res, argt) in _lib.SIGNATURES.items():
            fn = getattr(L, fn_name)
            fn.restype = res
            fn.argtypes = argt
        _lib._lib = L
        t = time.perf_counter()
        predictor = api.Predictor(api.Model.read_slice(raw)[0], False, device=0)
        t_create = time.perf_counter() - t
        d_scores = torch.zeros(nb + 1, dtype=torch.int32, device=dev)
        d_labels = torch.zeros(nb + 1, dtype=torch.uint8, device=dev)
        batch = api.DeviceBatch(predictor, timing=True)
        batch.set_max_sentence_chars(int(np.max(np.diff(ooff.astype(np.int64)))) + 1)

        def step():
            batch.predict(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, max_bytes, d_scores.data_ptr(),
                          d_labels.data_ptr(), stream)
        for _ in range(args.warmup):
            step()
        batch.sync()
        batch.kernel_ms()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t) / args.steps
        batch.sync()
        kernel_ms, tiles = batch.kernel_ms()
        ok = bool(np.array_equal(d_scores[:nb].cpu().numpy(), o_scores) and np.array_equal(d_labels[:nb].cpu().numpy(), o_labels))
        print(json.dumps({"variant": v, "model": name, "ms_per_step": round(ms, 4), "kernel_ms": round(kernel_ms, 4), "tiles": tiles,
                          "G_boundaries_per_s": round(nb / ms / 1e6, 2), "parity": ok, "create_s": round(t_create, 1),
                          "sentences": S, "len": [args.min_len, args.max_len]}), flush=True)
        del batch, predictor


if __name__ == "__main__":
    main()
