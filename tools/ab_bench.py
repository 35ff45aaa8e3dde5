"""A/B on the GPU box in ONE process (python tools/ab_bench.py [--model-kind K] [--variants a,b,..]): the bench
workload (100 K x 64-char sentences resident in HBM) through the built library and through prebuilt variants
(tools/prebuilt/*.so) or debug knobs, each checked bit for bit against the oracle.  Model, batch and oracle scores are
made once; a library costs one table compilation, each of its environments only its timed steps.  One line of JSON per
variant.  Variants are interleaved `--rounds` times so that clock drift between them shows up as spread, not as bias.

A variant is NAME or NAME:ENV=VAL[:ENV=VAL..]; NAME = `new` (the built library) or the stem of tools/prebuilt/libvaporetto_<NAME>.so.
`--ablate 0,256,1,..` adds new:VPT_DEBUG_ABLATE=<v> for every v (timing ablations of kernels_fast.hip; parity is
expected to fail for those that skip work)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KNOBS = ("VPT_DEBUG_LDS_PAD", "VPT_DEBUG_ABLATE", "VPT_FORCE_CUT_TILES", "VPT_TILE_FLAT", "VPT_FORCE_WINDOW_TABLE", "VPT_FORCE_GENERIC")


def lib_path(name):
    if name == "new":
        return os.path.join(ROOT, "vaporetto_amd", "lib", "libvaporetto_hip.so")
    return os.path.join(ROOT, "tools", "prebuilt", "libvaporetto_%s.so" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="new,new:VPT_INLINE_ASSIGN=1")
    ap.add_argument("--ablate", default="")
    ap.add_argument("--model-kind", type=int, default=1)
    ap.add_argument("--model-scale", type=float, default=1.0)
    ap.add_argument("--sentences", type=int, default=100000)
    ap.add_argument("--min-len", type=int, default=64)
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--phases", action="store_true", help="VPT_PROFILE_PHASES: wave 0's shader cycles per phase and tile (slows the kernel)")
    args = ap.parse_args()
    import torch
    from oracle import cbind
    from vaporetto_amd import _lib, api, synth
    import bench
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    raw, name = bench.load_model_bytes(args.model_kind, args.model_scale)
    utf8, boff = synth.synth_sentences(raw, args.sentences, args.min_len, args.max_len, seed=synth.SEED_BASE + 2)
    o_scores, o_labels, o_ooff, _ = cbind.OraclePredictor(raw).predict_batch(utf8, boff, nthreads=os.cpu_count() or 1)
    ooff = o_ooff
    S, nb = args.sentences, int(ooff[-1])
    max_bytes = int(np.max(np.diff(boff.astype(np.int64))))
    d_text = torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev)
    d_boff = torch.from_numpy(boff.astype(np.int64)).to(dev)
    d_ooff = torch.from_numpy(ooff.astype(np.int64)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    variants = [v for v in args.variants.split(",") if v]
    variants += ["new:VPT_DEBUG_ABLATE=%s" % a for a in args.ablate.split(",") if a]
    by_lib = {}
    for v in variants:
        parts = v.split(":")
        by_lib.setdefault(parts[0], []).append((v, dict(p.split("=", 1) for p in parts[1:])))
    results = {}
    for libname, envs in by_lib.items():
        path = lib_path(libname)
        if not os.path.exists(path):
            print(json.dumps({"variant": libname, "error": "missing " + path}), flush=True)
            continue
        L = C.CDLL(path)
        for fn_name, (res, argt) in _lib.SIGNATURES.items():
            if not hasattr(L, fn_name):
                continue   # an older prebuilt library
            fn = getattr(L, fn_name)
            fn.restype = res
            fn.argtypes = argt
        _lib._lib = L
        for k in KNOBS:
            os.environ.pop(k, None)
        if args.phases:
            os.environ["VPT_PROFILE_PHASES"] = "1"
        t = time.perf_counter()
        predictor = api.Predictor(api.Model.read_slice(raw)[0], False, device=0)
        t_create = time.perf_counter() - t
        d_scores = torch.zeros(nb + 1, dtype=torch.int32, device=dev)
        d_labels = torch.zeros(nb + 1, dtype=torch.uint8, device=dev)
        for rnd in range(args.rounds):
            for v, env in envs:
                for k in KNOBS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                # occupancy knobs are read when the predictor / workspace is made
                # the library reads its knobs once: table-level ones when a predictor is made, launch-level ones when a workspace is
                pred = api.Predictor(api.Model.read_slice(raw)[0], False, device=0) if ("VPT_DEBUG_LDS_PAD" in env or "VPT_FORCE_WINDOW_TABLE" in env) else predictor
                batch = api.DeviceBatch(pred, timing=True)
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
                if args.phases:
                    ph = batch.phase_cycles()
                    results.setdefault(v + "#phases", {"variant": v, "phase_cycles_per_tile": [round(c / max(tiles, 1) / max(args.steps + args.warmup, 1)) for c in ph]})
                ok = bool(np.array_equal(d_scores[:nb].cpu().numpy(), o_scores) and np.array_equal(d_labels[:nb].cpu().numpy(), o_labels))
                r = results.setdefault(v, {"variant": v, "model": name, "ms_per_step": [], "kernel_ms": [], "tiles": tiles, "parity": ok,
                                           "create_s": round(t_create, 1), "sentences": S, "len": [args.min_len, args.max_len]})
                r["ms_per_step"].append(round(ms, 4))
                r["kernel_ms"].append(round(kernel_ms, 4))
                r["parity"] = r["parity"] and ok
                del batch
        for v, _ in envs:
            r = results[v]
            r["G_boundaries_per_s"] = round(nb / min(r["ms_per_step"]) / 1e6, 2)
            print(json.dumps(r), flush=True)
            if v + "#phases" in results:
                print(json.dumps(results[v + "#phases"]), flush=True)
        # every handle of this library is destroyed BY this library (the next one's structs may differ)
        pred = predictor = batch = None
        import gc
        gc.collect()


if __name__ == "__main__":
    main()
