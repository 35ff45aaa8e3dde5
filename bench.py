#!/usr/bin/env python
"""bench.py -- boundary scores/sec of the MI355X hot path (BASELINE.json metric).

One "step" = one pass of Predictor::predict over one batch of synthetic sentences that is already resident in
HBM (configs[1]: bccwj-suw+unidic-shaped model, 100 K sentences x 64 chars per GPU).  With N GPUs every rank
scores its own batch (sentences shard trivially; no data-path collective) after the model file has been
broadcast from rank 0 over RCCL -- weak scaling.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def load_model_bytes(kind: int, scale: float):
    """A real model when $VAPORETTO_MODEL_DIR holds one (zstd, as distributed), else the synthetic stand-in."""
    from vaporetto_amd import synth
    name = {1: "bccwj-suw+unidic", 2: "jp-0.4.7-5", 3: "bccwj-suw+unidic_pos+pron"}[kind]
    d = os.environ.get("VAPORETTO_MODEL_DIR")
    if d:
        path = os.path.join(d, name + ".model.zst")
        if os.path.exists(path):
            env = dict(os.environ, LD_LIBRARY_PATH="/opt/conda/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
            raw = subprocess.check_output(["/opt/conda/bin/zstd", "-d", "-c", path], env=env)
            return raw, name
        path = os.path.join(d, name + ".mod")   # a KyTea model (jp-0.4.7-5.mod): converted like convert_kytea_model does
        if os.path.exists(path):
            from vaporetto_amd import kytea
            with open(path, "rb") as fh:
                return kytea.convert(fh.read()), name
    return synth.synth_model(kind, synth.SEED_BASE + 2, scale), "synthetic-" + {1: "M1", 2: "M2", 3: "M3"}[kind]


def measured_traffic(kernel: str, model: str, sentences: int):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            entries = json.load(fh)["entries"]
    except (OSError, ValueError, KeyError):
        return None
    for e in reversed(entries):
        if e["kernel"] == kernel and e["model"] == model and e["sentences_per_gpu"] == sentences:
            return int(1024 * (e["fetch_size_kib"] * e["fetch_correction"] + e["write_size_kib"]))
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sentences", type=int, default=100000, help="sentences per GPU per step")
    ap.add_argument("--min-len", type=int, default=64)
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--model-kind", type=int, default=1, help="1 bccwj-suw+unidic-like, 2 jp-0.4.7-5-like, 3 = 1 + tag models")
    ap.add_argument("--model-scale", type=float, default=1.0)
    ap.add_argument("--predict-tags", action="store_true", help="Predictor::new(model, true): the reference's BoundaryTag scorers (boundary scores only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--time-tags", action="store_true", help="with --predict-tags: also time vpt_fill_tags_batch_device (extra JSON field `tags`)")
    ap.add_argument("--phases", action="store_true", help="diagnostics: per-phase shader cycles of the scoring kernel (slows it)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vaporetto_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- model: rank 0 loads/synthesises, everyone receives the bytes over RCCL (xGMI) -- vaporetto_amd/dist.py
    from vaporetto_amd import dist as vdist
    model_bytes, model_name = (load_model_bytes(args.model_kind, args.model_scale) if rank == 0 else (None, ""))
    model_bytes = vdist.broadcast_model_bytes(model_bytes, src=0, device=dev)
    predictor = api.Predictor(api.Model.read_slice(model_bytes)[0], args.predict_tags, device=local_rank)
    info = predictor.info()

    # ---- this rank's batch, resident in HBM
    from vaporetto_amd import synth
    utf8, boff = synth.synth_sentences(model_bytes, args.sentences, args.min_len, args.max_len,
                                       seed=synth.SEED_BASE + 2 + 1000 * rank)
    ooff = api.count_boundaries(utf8, boff)
    S, nb, nbytes = args.sentences, int(ooff[-1]), int(boff[-1])
    max_bytes = int(np.max(np.diff(boff.astype(np.int64))))
    d_text = torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev)
    d_boff = torch.from_numpy(boff.astype(np.int64)).to(dev)
    d_ooff = torch.from_numpy(ooff.astype(np.int64)).to(dev)
    d_scores = torch.empty(nb + 1, dtype=torch.int32, device=dev)
    d_labels = torch.empty(nb + 1, dtype=torch.uint8, device=dev)
    if args.phases:
        os.environ["VPT_PROFILE_PHASES"] = "1"
    batch = api.DeviceBatch(predictor, timing=True)
    batch.set_max_sentence_chars(int(np.max(np.diff(ooff.astype(np.int64)))) + 1)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        batch.predict(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, max_bytes,
                      d_scores.data_ptr(), d_labels.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    batch.sync()
    batch.kernel_ms()  # reset the event ring
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    batch.sync()
    kernel_ms, n_tiles = batch.kernel_ms()
    phases = batch.phase_cycles() if args.phases else None
    elapsed, total_boundaries = vdist.reduce_throughput(elapsed, float(nb), device=dev)

    tags_info = None
    if args.time_tags and args.predict_tags and predictor.n_tags() > 0:
        nt = predictor.n_tags()
        d_tags = torch.empty((nb + S) * nt + 1, dtype=torch.int32, device=dev)

        def tag_step():
            batch.fill_tags(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(), d_tags.data_ptr(), stream)

        for _ in range(max(1, args.warmup)):
            tag_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tag_step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        lab = d_labels[:nb].cpu().numpy()
        n_tokens = int((lab == 1).sum()) + S
        tags_info = {"ms_per_step": 1e3 * dt, "tokens_per_s": n_tokens / dt, "chars_per_s": (nb + S) / dt, "n_tags": nt,
                     "note": "decode_chars_kernel + tag_tokens_kernel on the predicted labels (first, untuned mapping)"}
        if not args.no_cpu_baseline and rank == 0:
            from oracle import cbind as _cb
            o = _cb.OraclePredictor(model_bytes, True)
            got = d_tags[:(nb + S) * nt].cpu().numpy().reshape(nb + S, nt)
            text = bytes(utf8)
            ok = True
            for i in range(0, S, max(1, S // 200)):    # a 200-sentence sample against the oracle
                t = text[int(boff[i]):int(boff[i + 1])].decode("utf-8")
                a = int(ooff[i])
                want, _ = o.predict_tags(t, labels=lab[a:a + len(t) - 1])
                ok = ok and bool(np.array_equal(got[a + i:a + i + len(t)], want))
            tags_info["parity_sample"] = ok

    if rank == 0:
        out = {
            "metric": "boundary scores/sec", "value": total_boundaries * args.steps / elapsed, "unit": "boundaries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %s model, %d sentences x %d..%d chars per GPU per step, inputs resident in HBM"
                       % (("configs[1]" if (args.model_kind, args.sentences, args.min_len, args.max_len) == (1, 100000, 64, 64)
                           else "configs[%d]-shaped" % {1: 1, 2: 3, 3: 4}[args.model_kind]), model_name, S, args.min_len, args.max_len),
                       "tokenizer_model": model_name, "sentences_per_gpu": S, "boundaries_per_gpu": nb, "text_bytes_per_gpu": nbytes,
                       "char_ngrams": info["n_char_ngrams"], "dict_words": info["n_dict_words"],
                       "table_bytes": info["device_table_bytes"], "hot_table_bytes": info["hot_table_bytes"],
                       "packed_tables": bool(info["packed"]), "tiles": n_tiles, "sharding": "sentences x%d ranks, no data-path collective" % world},
        }
        if phases is not None:
            tot = float(sum(phases[:5])) or 1.0
            out["phase_share"] = dict(zip(["scan", "decode", "patterns", "barrier", "output"], [round(p / tot, 4) for p in phases[:5]]))
            out["phase_cycles_per_tile"] = [round(p / max(n_tiles, 1) / (args.steps + args.warmup), 1) for p in phases[:5]]
        # ---- roofline of the dominant kernel (score_tiles_kernel): algorithmic bytes per launch / its duration
        a_stream = nbytes + 5 * nb + 16 * S   # text + i32 score + u8 label per boundary + two u64 offsets per sentence
        a_type = 4 * nb                        # one type-window table word per boundary (the reference's cache form)
        a_char = None
        cpu = None
        kernel_name = "score_tiles_fast_kernel" if info["packed"] and info["type_kind"] in (0, 1) else "score_tiles_kernel"
        if not args.no_cpu_baseline:
            from oracle import cbind
            orc = cbind.OraclePredictor(model_bytes, args.predict_tags)
            ncores = os.cpu_count() or 1
            t = time.perf_counter()
            o_scores, o_labels, _, a_char = orc.predict_batch(utf8, boff, nthreads=1)
            t1 = time.perf_counter() - t
            reps = max(1, min(20, int(10.0 / max(t1 / ncores * 1.5, 1e-3))))
            t = time.perf_counter()
            for _ in range(reps):
                orc.predict_batch(utf8, boff, nthreads=ncores)
            tn = (time.perf_counter() - t) / reps
            cpu = {"value": nb / tn, "unit": "boundaries/s", "cores": ncores, "kind": "port",
                   "single_thread_value": nb / t1,
                   "sample": "the same %d-sentence batch: 1 pass on 1 thread, %d passes on %d threads (C restatement of the "
                             "reference algorithm, not the Rust binary)" % (S, reps, ncores)}
            g_scores = d_scores[:nb].cpu().numpy()
            g_labels = d_labels[:nb].cpu().numpy()
            out["parity"] = bool(np.array_equal(g_scores, o_scores) and np.array_equal(g_labels, o_labels))
        if a_char is not None and kernel_ms > 0:
            a = a_stream + a_char + a_type
            achieved = a / (kernel_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS,
                               "traffic": measured_traffic(kernel_name, model_name, S) if args.min_len == 64 and args.max_len == 64 else None,
                               "kernel": kernel_name, "kernel_ms": kernel_ms,
                               "algorithmic_bytes_per_launch": a, "bytes_per_boundary": a / nb,
                               "a_stream": a_stream, "a_char": a_char, "a_type": a_type}
        else:
            out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                               "traffic": None, "kernel": kernel_name, "kernel_ms": kernel_ms}
        out["cpu_baseline"] = cpu
        if tags_info is not None:
            out["tags"] = tags_info
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
