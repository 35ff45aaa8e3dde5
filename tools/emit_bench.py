"""The writer alone on the GPU box (python tools/emit_bench.py [--config 1|4] [--env K=V ..]): vpt_write_tokenized_batch_device over a bench workload's text with the
predicted labels, HIP-event time per launch, bytes moved against the HBM peak, the output compared byte for byte with the oracle's writer.  One JSON line per
environment (the library reads its launch-level knobs when a workspace is made: VPT_EMIT_WAVE_BLOCKS, VPT_EMIT_RUN_CHARS, VPT_EMIT_PER_BLOCK)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--env", nargs="*", default=["", "VPT_EMIT_WAVE_BLOCKS=1"])
    args = ap.parse_args()
    import torch
    import bench
    from oracle import cbind
    from vaporetto_amd import api
    cfg = bench.CONFIGS[args.config]
    raw, name = bench.load_model_bytes(cfg["kind"], 1.0)
    utf8, boff, ooff, _, S = bench.make_shard(cfg, raw, 0, 1, os.cpu_count() or 1)
    nb, nbytes = int(ooff[-1]), int(boff[-1])
    tags = bool(cfg["tags"])
    orc = cbind.OraclePredictor(raw, tags)
    _, o_labels, _, _ = orc.predict_batch(utf8, boff, nthreads=os.cpu_count() or 1)
    o_tags = o_models = None
    if tags:
        o_tags, _, o_models = orc.fill_tags_batch(utf8, boff, ooff, o_labels, nthreads=os.cpu_count() or 1, want_scores=False)
    o_text, o_toff = orc.write_tokenized_batch(utf8, boff, ooff, o_labels, o_tags, o_models, nthreads=os.cpu_count() or 1)
    dev = torch.device("cuda", 0)
    d_text = torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev)
    d_boff = torch.from_numpy(boff.astype(np.int64)).to(dev)
    d_ooff = torch.from_numpy(ooff.astype(np.int64)).to(dev)
    d_labels = torch.from_numpy(np.concatenate([o_labels, np.zeros(16, np.uint8)])).to(dev)
    pred = api.Predictor(api.Model.read_slice(raw)[0], tags, device=0)
    nt = pred.n_tags() if tags else 0
    cap = 3 * nbytes + 64 + (nbytes * pred.max_tag_suffix() if nt else 0)
    d_out = torch.empty(cap + 1, dtype=torch.uint8, device=dev)
    d_toff = torch.empty(S + 1, dtype=torch.int64, device=dev)
    d_tags = torch.empty((nb + S) * nt + 1, dtype=torch.int32, device=dev) if nt else None
    stream = torch.cuda.current_stream().cuda_stream
    for env in args.env:
        kv = dict(p.split("=", 1) for p in env.split(",") if p)
        for k in ("VPT_EMIT_WAVE_BLOCKS", "VPT_EMIT_WAVE_TAGGED", "VPT_EMIT_RUN_CHARS", "VPT_EMIT_PER_BLOCK"):
            os.environ.pop(k, None)
        os.environ.update(kv)
        batch = api.DeviceBatch(pred)
        if nt:   # the writer takes the tags and the token words fill_tags left on this workspace
            batch.fill_tags(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(), d_tags.data_ptr(), stream)

        def emit():
            if nt:
                batch.write_tagged(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(), d_tags.data_ptr(), d_out.data_ptr(), cap, d_toff.data_ptr(), stream)
            else:
                batch.write_tokenized(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(), d_out.data_ptr(), cap, d_toff.data_ptr(), stream)
        for _ in range(5):
            emit()
        batch.sync()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in ev:
            a.record()
            emit()
            b.record()
        torch.cuda.synchronize()
        batch.sync()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        toff = d_toff.cpu().numpy().astype(np.uint64)
        out_bytes = int(toff[-1])
        ok = bool(np.array_equal(toff, o_toff) and np.array_equal(d_out[:out_bytes].cpu().numpy(), o_text))
        moved = nbytes + nb + out_bytes + 16 * S + (4 * (nb + S) * (nt + 1) if nt else 0)   # text + labels (+ tags, token words) in, text + offsets out
        print(json.dumps({"env": env, "workload": cfg["name"], "model": name, "ms": round(ms, 4), "GBps": round(moved / ms / 1e6, 1), "frac_of_hbm": round(moved / ms / 1e6 / 8000.0, 4),
                          "bytes_moved": moved, "out_bytes": out_bytes, "parity": ok}), flush=True)
        del batch


if __name__ == "__main__":
    main()
