#!/usr/bin/env python
"""bench.py -- boundary scores/sec of the MI355X hot path (BASELINE.json metric).

One "step" = one pass of Predictor::predict over one batch of synthetic sentences that is already resident in HBM.
The workloads are BASELINE.json's configs:

  configs[1]  bccwj-suw+unidic-shaped model (synthetic M1), 100 K sentences x 64 chars
  configs[2]  the same model, 10 M sentences x 64 chars cut into character-balanced shards over the N ranks (strong
              scaling; the predictor is compiled on rank 0 and its tables are broadcast over RCCL)
                                                                   -- the line's `value` at EVERY N, N = 1 included
  configs[3]  jp-0.4.7-5-shaped model (synthetic M2, dictionary heavy), 100 K x 64
  configs[4]  M1 + tag models (synthetic M3), predict_tags on, 1 M sentences of 8..512 chars / N, step = predict + fill_tags
  documents   (not a BASELINE config) M1, 10 K sentences of 2 000..20 000 chars: every sentence is cut across many tiles
  charw2 / charw4  (not BASELINE configs) M1 trained with --charw 2 --typew 2 / --charw 4 --typew 4, 100 K x 64
  nonbmp      (not a BASELINE config) M1 + patterns with kanji outside the BMP, 100 K x 64 with about 1 % of the chars outside the BMP

ONE workload over the whole 1 -> 8 curve: with no --config the `value` is configs[2] at every N (the north-star's "10 M-sentence
synthetic batch"; it fits one GPU), so that a scaling curve built from the per-N values compares like with like.  N = 1 also
folds configs[1], [3], [4] and the documents workload into `workloads`, each with its own parity, kernel time and roofline (configs[1] carries the
PCIe end-to-end figures).  (--quick: the primary workload only.)

    python bench.py --gpus N --steps 20 --warmup 3        # N > 1 without WORLD_SIZE: launches its N ranks itself (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` (N > 1) outside torch.distributed.run re-executes itself under it, one rank per GPU over RCCL; when that
job fails (e.g. RCCL cannot initialise) it falls back to ONE process that clones the predictor to the N devices
(vpt_predictor_clone_to_device: hipMemcpyPeer) and drives one host thread + stream per device -- a measured curve either
way, and the line says which (`config.launch`).  Fewer than N visible HIP devices is an error (exit code 2), never a
silent 1-GPU run.  (Test hook for a 1-GPU box: VPT_BENCH_ONE_DEVICE=1 puts every rank on device 0, VPT_BENCH_BACKEND=gloo.)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PCIE_GBS = 63.0         # MI355X_MICROARCH.md: PCIe Gen5 x16, per direction
PCIE_DUPLEX_GBS = 84.2  # measured on this link with both directions busy at once (one copy each way, 21 MB in + 31 MB out: profiles/r02_c6_pcie_microbench.txt;
                        # one direction alone: 56 GB/s) -- the two directions share: a batch that needs its input before it can send output is priced against THIS

CONFIGS = {
    1: dict(name="configs[1]", kind=1, sentences=100_000, min_len=64, max_len=64, tags=False, blocks=False),
    2: dict(name="configs[2]", kind=1, sentences=10_000_000, min_len=64, max_len=64, tags=False, blocks=True),
    3: dict(name="configs[3]", kind=2, sentences=100_000, min_len=64, max_len=64, tags=False, blocks=False),
    4: dict(name="configs[4]", kind=3, sentences=1_000_000, min_len=8, max_len=512, tags=True, blocks=True),
    # not a BASELINE config: whole documents as ONE sentence each, what the reference's tantivy adapter feeds Predictor::predict
    # (vaporetto_tantivy/src/lib.rs:171-176) -- every sentence spans many tiles
    5: dict(name="documents", kind=1, sentences=10_000, min_len=2_000, max_len=20_000, tags=False, blocks=False),
    # the windows are free parameters of the trainer (train/src/main.rs:33-51): M1 trained with --charw 2 --typew 2 (laid out in the rows
    # of window 3) and with --charw 4 --typew 4 (rows of window 4): the specialised kernel, one instance per row window
    6: dict(name="charw2", kind=4, sentences=100_000, min_len=64, max_len=64, tags=False, blocks=False),
    7: dict(name="charw4", kind=5, sentences=100_000, min_len=64, max_len=64, tags=False, blocks=False),
    # M1 + 40 unigrams, 100 n-grams and 100 dictionary words with kanji outside the BMP (UniDic has such entries); 2.5 % of the text's items
    # (about 1 % of its chars) lie outside the BMP, half of them as one of those patterns: configs[1] with the alphabet a real dictionary has
    8: dict(name="nonbmp", kind=6, sentences=100_000, min_len=64, max_len=64, tags=False, blocks=False, nonbmp_share=0.025),
}
BLOCK = 100_000


def load_model_bytes(kind: int, scale: float):
    """A real model when $VAPORETTO_MODEL_DIR holds one (zstd, as distributed), else the synthetic stand-in."""
    from vaporetto_amd import synth
    name = {1: "bccwj-suw+unidic", 2: "jp-0.4.7-5", 3: "bccwj-suw+unidic_pos+pron", 4: "bccwj-suw+unidic-charw2", 5: "bccwj-suw+unidic-charw4", 6: "bccwj-suw+unidic-nonbmp"}[kind]
    d = os.environ.get("VAPORETTO_MODEL_DIR")
    if d:
        path = os.path.join(d, name + ".model.zst")
        if os.path.exists(path):
            try:    # zstd is outside the reference's API too (README.md:50-63): whatever decoder this image has
                import pyarrow as pa
                with open(path, "rb") as fh:
                    raw = pa.CompressedInputStream(pa.BufferReader(fh.read()), "zstd").read()
            except Exception:
                env = dict(os.environ, LD_LIBRARY_PATH="/opt/conda/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
                raw = subprocess.check_output(["/opt/conda/bin/zstd", "-d", "-c", path], env=env)
            return raw, name
        path = os.path.join(d, name + ".mod")   # a KyTea model (jp-0.4.7-5.mod): converted like convert_kytea_model does
        if os.path.exists(path):
            from vaporetto_amd import kytea
            with open(path, "rb") as fh:
                return kytea.convert(fh.read()), name
    return synth.synth_model(kind, synth.SEED_BASE + 2, scale), "synthetic-" + {1: "M1", 2: "M2", 3: "M3", 4: "M1-charw2-typew2", 5: "M1-charw4-typew4", 6: "M1-nonbmp"}[kind]


def kernel_source_hash() -> str:
    """What the rocprofv3 traffic numbers of profiles/traffic.json are valid for: the sources of the scoring kernels, of the tables
    they read and of the host code that plans their tiles (the tag kernel and the writer are measured by other means)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vaporetto_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cpp", ".h", ".hpp")) and f not in ("kernels_tags.hip", "kernels_emit.hip"):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def measured_entry(kernel: str, model: str, workload: str):
    """The whole entry of profiles/traffic.json for `kernel` on THESE sources (see measured_traffic), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            entries = json.load(fh)["entries"]
    except (OSError, ValueError, KeyError):
        return None
    sha = kernel_source_hash()
    for e in reversed(entries):
        if e.get("kernel") == kernel and e.get("model") == model and e.get("workload") == workload and e.get("source_hash") == sha:
            return e
    return None


def gather_rates():
    """tools/tcp_bench on MI355X (profiles/r06_j_tcp_bench.jsonl): random 16-byte loads per second from a table that fits the vector L1, the
    L2, neither -- what each level of the memory system serves when it is the only one asked."""
    rates = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r06_j_tcp_bench.jsonl")) as fh:
            for l in fh:
                r = json.loads(l)
                if r.get("shape") == "b16":
                    rates[{"L1": "l1", "L2": "l2", "HB": "hbm"}[r["table"][:2]]] = r["G_items_s"]
    except (OSError, ValueError, KeyError):
        return None
    return rates if len(rates) == 3 else None


def gather_block(entry, node_reads, chars, wl, kernel_ms):
    """roofline.gather: what the scoring kernel asks of the memory system per launch, in 16-byte lane loads, against what
    tools/tcp_bench measured each level to serve.  lane_loads_per_launch: counted by a diagnostics launch of the same batch (vpt_batch_node_reads:
    unigram / bigram / trigram nodes, deep entries and rows) + one char-table word per char.  With a PMC pass on these sources (tools/profile.sh):
    the vector L1's lookups, how many of them went to the L2 and how many of those to the fabric; each level's load against its microbench rate --
    the largest is the level that binds, and `frac` is how close the kernel runs to it."""
    if kernel_ms is None or kernel_ms <= 0:
        return None
    out = {"unit": "16-byte lane loads"}
    if node_reads:
        q = {"unigram_nodes": {3: 1, 4: 2, 5: 2, 6: 2, 7: 4, 8: 4}[wl], "bigram_nodes": 2 if wl == 3 else 4, "trigram_nodes": {3: 1, 4: 2, 5: 2, 6: 2, 7: 2, 8: 4}[wl],
             "deep_entries": 1, "deep_rows": 1, "global_type_rows": ((2 * wl + 3) & ~3) // 4}
        loads = int(sum(node_reads.get(k, 0) * q[k] for k in q) + chars)
        out.update({"lane_loads_per_launch": loads, "achieved_Glanes_s": loads / (kernel_ms * 1e-3) / 1e9, "node_reads_per_launch": node_reads})
    rates = gather_rates()
    if rates:
        out["microbench_Glanes_s"] = rates
        out["microbench_source"] = "profiles/r06_j_tcp_bench.jsonl (tools/tcp_bench: random 16-byte loads from a 16 KB / 2 MB / 256 MB table)"
    if entry and rates and entry.get("tcp_lookups"):
        look, l2req, l2miss = float(entry["tcp_lookups"]), float(entry.get("tcp_tcc_read_req") or 0), float(entry.get("tcc_miss") or 0)
        t = kernel_ms * 1e-3
        levels = {"l1": look / t / 1e9 / rates["l1"], "l2": l2req / t / 1e9 / rates["l2"], "hbm": l2miss / t / 1e9 / rates["hbm"]}
        bound = max(levels, key=levels.get)
        out.update({"pmc": {"tcp_lookups_per_launch": int(look), "of_them_to_the_l2": int(l2req), "of_those_to_the_fabric": int(l2miss),
                            "tcp_busy_share": (float(entry["tcp_busy_cycles"]) / float(entry["tcp_cycles"])) if entry.get("tcp_cycles") else None},
                    "load_by_level": {k: round(v, 4) for k, v in levels.items()}, "bound": bound, "frac": round(levels[bound], 4),
                    "reading": "each level's requests per second over what tools/tcp_bench measured that level to serve alone; the levels work in parallel, the "
                               "largest share is the one that binds"})
    return out


def measured_traffic(kernel: str, model: str, workload: str):
    """HBM-side bytes per launch of `kernel` from the rocprofv3 PMC passes tools/profile.sh ran on THESE sources
    (profiles/traffic.json: reads = 128 B x TCC_EA0_RDREQ_128B + 64 B x .._64B + 32 B x .._32B, writes = WRITE_SIZE; both
    checked against known byte counts, profiles/r02_c2_read_request_sizes.txt), or None when the sources have changed since."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            entries = json.load(fh)["entries"]
    except (OSError, ValueError, KeyError):
        return None
    sha = kernel_source_hash()
    for e in reversed(entries):
        if e.get("kernel") == kernel and e.get("model") == model and e.get("workload") == workload and e.get("source_hash") == sha:
            return int(e["read_bytes"] + e["write_bytes"])
    return None


def cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def make_shard(cfg, raw, rank: int, world: int, ncores: int, sentences_override: int = 0):
    """This rank's shard of the workload's batch: (utf8, byte offsets, boundary offsets, index of its first sentence, sentences of the whole batch)."""
    from vaporetto_amd import api, synth, dist as vdist
    S, lo_len, hi_len = cfg["sentences"], cfg["min_len"], cfg["max_len"]
    if sentences_override:
        S = sentences_override
    if not cfg["blocks"]:
        utf8, boff = synth.synth_sentences(raw, S, lo_len, hi_len, seed=synth.SEED_BASE + 2, nonbmp_share=cfg.get("nonbmp_share", 0.0))
        ooff = api.count_boundaries(utf8, boff)
        if world == 1:
            return utf8, boff, ooff, 0, S
        u, b, o, first = vdist.take_shard(utf8, boff, ooff, rank, world)
        return u, b, o, first, S
    n_blocks = (S + BLOCK - 1) // BLOCK
    S = n_blocks * BLOCK
    seed = synth.SEED_BASE + (3 if cfg["name"] == "configs[2]" else 5)
    if lo_len == hi_len:
        # equal lengths: the global offsets are arithmetic, so a rank generates only the blocks its shard touches
        g_ooff = np.arange(S + 1, dtype=np.uint64) * np.uint64(lo_len - 1)
        bounds = vdist.shard_bounds(g_ooff, world)
        a, e = int(bounds[rank]), int(bounds[rank + 1])
        b0, b1 = a // BLOCK, (e + BLOCK - 1) // BLOCK
        utf8, boff = synth.synth_blocks(raw, b0, max(b1 - b0, 1), BLOCK, lo_len, hi_len, seed=seed, nthreads=min(32, ncores))
        i0, i1 = a - b0 * BLOCK, e - b0 * BLOCK
        t0, t1 = int(boff[i0]), int(boff[i1])
        utf8, boff = np.ascontiguousarray(utf8[t0:t1]), (boff[i0:i1 + 1] - boff[i0]).astype(np.uint64)
        ooff = (g_ooff[a:e + 1] - g_ooff[a]).astype(np.uint64)
        return utf8, boff, ooff, a, S
    # ragged lengths: every rank generates the batch and takes its character-balanced shard
    utf8, boff = synth.synth_blocks(raw, 0, n_blocks, BLOCK, lo_len, hi_len, seed=seed, nthreads=min(32, ncores))
    ooff = api.count_boundaries(utf8, boff)
    if world == 1:
        return utf8, boff, ooff, 0, S
    u, b, o, first = vdist.take_shard(utf8, boff, ooff, rank, world)
    return u, b, o, first, S


def die(msg: str, code: int = 2):
    sys.stderr.write("bench.py: error: %s\n" % msg)
    sys.stderr.flush()
    raise SystemExit(code)


class Runner:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:   # never a line whose n_gpus is not what was asked for
            die("--gpus %d but WORLD_SIZE=%d: launch %d ranks (python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d), "
                "or plain `python bench.py --gpus %d`, which launches them itself" % (args.gpus, self.world, args.gpus, args.gpus, args.gpus, args.gpus))
        visible = torch.cuda.device_count()
        # (test hook for a 1-GPU box: VPT_BENCH_ONE_DEVICE=1 VPT_BENCH_BACKEND=gloo runs the N-rank code path with every rank on device 0)
        if os.environ.get("VPT_BENCH_ONE_DEVICE"):
            self.local_rank = 0
        if visible <= self.local_rank:
            die("rank %d of %d wants HIP device %d but %d device(s) are visible: --gpus %d needs %d GPUs on this node"
                % (self.rank, self.world, self.local_rank, visible, args.gpus, args.gpus))
        backend = os.environ.get("VPT_BENCH_BACKEND", "nccl")
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.dist_failed = None     # why the N-rank job cannot run as one (then rank 0 falls back to driving the N devices itself: main())
        if self.world > 1:
            import datetime
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            try:
                if os.environ.get("VPT_BENCH_FAIL_INIT"):   # test hook: as if RCCL could not initialise
                    raise RuntimeError("VPT_BENCH_FAIL_INIT")
                if backend == "nccl":
                    dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev, timeout=datetime.timedelta(seconds=300))   # RCCL
                else:
                    dist.init_process_group(backend, rank=self.rank, world_size=self.world, timeout=datetime.timedelta(seconds=300))
                # the first collective, here: a fabric that does not work shows before any table is compiled
                t = torch.ones(1, dtype=torch.int32, device=self.dev if backend == "nccl" else torch.device("cpu"))
                dist.all_reduce(t)
                if int(t.item()) != self.world:
                    raise RuntimeError("all_reduce over %d ranks returned %d" % (self.world, int(t.item())))
            except Exception as e:   # noqa: BLE001 -- anything: the line must still be printed
                self.dist_failed = "%s: %s" % (type(e).__name__, str(e).replace("\n", " ")[:300])
                sys.stderr.write("bench.py: rank %d: torch.distributed (%s) is not usable: %s\n" % (self.rank, backend, self.dist_failed))
        self.backend = backend if self.world > 1 else None
        self.ncores = max(1, (os.cpu_count() or 1) // self.world)

    # ---- the predictor: compiled on rank 0, its tables broadcast device to device
    def make_predictor(self, kind: int, tags: bool):
        from vaporetto_amd import api, dist as vdist
        raw, name = (load_model_bytes(kind, self.args.model_scale) if self.rank == 0 else (None, ""))
        raw = vdist.broadcast_model_bytes(raw, src=0, device=self.dev)   # every rank draws its text from the model's patterns
        t = time.perf_counter()
        p0 = api.Predictor(api.Model.read_slice(raw)[0], tags, device=self.local_rank) if self.rank == 0 else None
        create_s = time.perf_counter() - t
        t = time.perf_counter()
        pred = vdist.broadcast_predictor(p0, src=0, device=self.dev, model_bytes=raw)
        self.tables_broadcast = getattr(pred, "tables_broadcast", None)   # view / staged / compile: which way the tables took (dist.py)
        if self.world > 1:
            self.torch.cuda.synchronize()
        bcast_s = time.perf_counter() - t
        if self.world > 1:
            names = [None]
            if self.rank == 0:
                names = [name]
            self.dist.broadcast_object_list(names, src=0)
            name = names[0]
        return pred, raw, name, create_s, bcast_s

    # ---- this rank's shard of the batch
    def make_shard(self, cfg, raw):
        return make_shard(cfg, raw, self.rank, self.world, self.ncores, self.args.sentences)

    def all_true(self, flag: bool) -> bool:
        if self.world == 1:
            return flag
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def run(self, cfg_id: int, primary: bool, e2e_leg: bool = False):
        torch, dist = self.torch, self.dist
        from vaporetto_amd import api, dist as vdist
        args = self.args
        cfg = CONFIGS[cfg_id]
        pred, raw, model_name, create_s, bcast_s = self.make_predictor(cfg["kind"], cfg["tags"])
        info = pred.info()
        t = time.perf_counter()
        utf8, boff, ooff, first, S_total = self.make_shard(cfg, raw)
        synth_s = time.perf_counter() - t
        S, nb, nbytes = len(boff) - 1, int(ooff[-1]), int(boff[-1])
        max_bytes = int(np.max(np.diff(boff.astype(np.int64)))) if S else 1
        max_chars = (int(np.max(np.diff(ooff.astype(np.int64)))) + 1) if S else 1
        dev = self.dev
        d_text = torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev)
        d_boff = torch.from_numpy(boff.astype(np.int64)).to(dev)
        d_ooff = torch.from_numpy(ooff.astype(np.int64)).to(dev)
        d_scores = torch.empty(nb + 1, dtype=torch.int32, device=dev)
        d_labels = torch.empty(nb + 1, dtype=torch.uint8, device=dev)
        if args.phases:
            os.environ["VPT_PROFILE_PHASES"] = "1"
        batch = api.DeviceBatch(pred, timing=True)
        batch.set_max_sentence_chars(max_chars)
        stream = torch.cuda.current_stream().cuda_stream
        nt = pred.n_tags() if cfg["tags"] else 0
        d_tags = torch.empty((nb + S) * nt + 1, dtype=torch.int32, device=dev) if nt else None

        def step():
            batch.predict(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, max_bytes,
                          d_scores.data_ptr(), d_labels.data_ptr(), stream)
            if nt:   # Sentence::fill_tags on the labels just predicted: configs[4]'s step is the whole tagging job.  What it leaves is what the tagged
                # writer reads: one record per token that has a tag model (the reference holds None everywhere else, predictor.rs:558-573); the dense
                # (chars x n_tags) array of the C ABI is timed beside it (tags.dense_ms_per_step) and BOTH forms are compared with the oracle below
                batch.fill_tags(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(), 0, stream)

        steps = args.steps if primary else max(10, min(args.steps, 20))
        for _ in range(args.warmup):
            step()
        if not primary:
            # The other workloads' steps are 0.1 .. 3 ms and follow seconds of host work (tables, synthetic text) during which the device idles:
            # W such steps are over before it is back at its clocks, and the K timed ones then carry the ramp (seen as 0.14 -> 0.40 ms per step
            # from one run to the next with the kernel's own time unchanged).  They warm up for at least 50 ms; the primary line keeps exactly W.
            t = time.perf_counter()
            while time.perf_counter() - t < 0.05:
                for _ in range(8):
                    step()
                batch.sync()
        batch.sync()
        batch.kernel_ms()  # reset the event ring
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        batch.sync()
        ktimes = batch.kernel_times()
        kernel_ms_mean, n_tiles = batch.kernel_ms()
        plan = batch.last_plan()
        kernel_ms = float(np.median(ktimes)) if len(ktimes) else kernel_ms_mean
        phases = batch.phase_cycles() if args.phases else None
        # With every rank on ONE device (the dry run of the scaling job on a 1-GPU box) the ranks' kernels share the device and an event
        # pair also times the others' work: the ranks then take turns for a few launches each, and that is the kernel time the line
        # reports per rank (`kernel_ms_solo`): N ranks' solo kernels should add up to the N = 1 kernel.
        kernel_ms_solo = None
        if self.world > 1 and os.environ.get("VPT_BENCH_ONE_DEVICE"):
            for r in range(self.world):
                dist.barrier()
                if r == self.rank:
                    for _ in range(5):
                        step()
                    batch.sync()
                    kt = batch.kernel_times()
                    kernel_ms_solo = float(np.median(kt[-5:])) if len(kt) else None
            dist.barrier()
        elapsed_local = elapsed
        elapsed, total_boundaries = vdist.reduce_throughput(elapsed, float(nb), device=dev)
        per_rank = None
        if self.world > 1:   # what every rank measured, so that a bad curve can be read from the line alone
            mine = {"rank": self.rank, "device": self.local_rank, "sentences": S, "boundaries": nb, "chars": nb + S, "kernel_ms": round(kernel_ms, 4),
                    "tables_broadcast": self.tables_broadcast,
                    "kernel_ms_solo": None if kernel_ms_solo is None else round(kernel_ms_solo, 4), "ms_per_step": round(1e3 * elapsed_local / steps, 4),
                    "tiles": n_tiles, "create_s": round(create_s, 2), "tables_broadcast_s": round(bcast_s, 3), "synth_s": round(synth_s, 2)}
            gathered = [None] * self.world
            dist.all_gather_object(gathered, mine)
            per_rank = gathered

        # ---- tag kernels on their own (event-free wall clock over the same steps)
        tags_info = None
        if nt:
            def timed_fill(dense_ptr):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    batch.fill_tags(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(), dense_ptr, stream)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / steps
            dt_dense = timed_fill(d_tags.data_ptr())       # the dense array with the call (a memset + a scatter of the records): d_tags holds it now
            dt = timed_fill(0)                             # the records alone: what the step above and the tagged writer below use
            d_tags2 = torch.full(((nb + S) * nt + 1,), 7, dtype=torch.int32, device=dev)
            batch.expand_tags(S, nb, d_tags2.data_ptr(), stream)   # ... expanded: must be the same array
            batch.sync()
            records_ok = bool(torch.equal(d_tags2[:(nb + S) * nt], d_tags[:(nb + S) * nt]))
            n_records_lb = int((d_tags2[:(nb + S) * nt].view(nb + S, nt) >= 0).any(dim=1).sum().item())
            del d_tags2
            lab = d_labels[:nb].cpu().numpy()
            n_tokens = int((lab == 1).sum()) + S
            tags_info = {"ms_per_step": 1e3 * dt, "dense_ms_per_step": 1e3 * dt_dense, "records_expand_to_the_dense_array": records_ok,
                         "tokens_per_s": n_tokens / dt, "chars_per_s": (nb + S) / dt, "n_tags": nt, "tokens_with_tags_at_least": n_records_lb,
                         "output": "ms_per_step: one record per token that has a tag model, sorted by position (what write_tagged reads); dense_ms_per_step: "
                                   "the (chars x n_tags) int32 array of vpt_fill_tags_batch_device as well",
                         "kernels": "decode_chars_kernel + tag_filter_summary_kernel + tag_front_flat_kernel (runs of sentences as consecutive chars) + "
                                    "scan_chained_kernel + tag_pass_kernel on the predicted labels"}

        # ---- token emission on the labels just predicted (Sentence::write_tokenized_text, with "/tag" suffixes for tag models)
        emit_info, emit_out = None, None
        if not args.no_emit:
            cap = 3 * nbytes + 64 + ((nbytes * pred.max_tag_suffix()) if nt else 0)
            d_out = torch.empty(cap + 1, dtype=torch.uint8, device=dev)
            d_toff = torch.empty(S + 1, dtype=torch.int64, device=dev)

            def emit():
                if nt:
                    batch.write_tagged(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(), 0,
                                       d_out.data_ptr(), cap, d_toff.data_ptr(), stream)
                else:
                    batch.write_tokenized(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, d_labels.data_ptr(),
                                          d_out.data_ptr(), cap, d_toff.data_ptr(), stream)
            for _ in range(2):
                emit()
            batch.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                emit()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            batch.sync()
            toff = d_toff.cpu().numpy().astype(np.uint64)
            out_bytes = int(toff[-1])
            moved = nbytes + nb + out_bytes + 16 * S + ((16 + 4 * nt) * tags_info["tokens_with_tags_at_least"] if nt else 0)   # text + labels (+ tag records) in, text + offsets out
            emit_info = {"ms_per_step": 1e3 * dt, "out_bytes": out_bytes, "algorithmic_GBps": moved / dt / 1e9, "frac_of_hbm": moved / dt / 1e9 / HBM_PEAK_GBS,
                         "kernels": "emit_flat_kernel: one launch (kernels_emit.hip), %s" % ("tagged, from the tag records" if nt else "boundaries only")}
            # the whole output is compared with the oracle's writer below (sentence.rs:850-886 restated in oracle/vaporetto_oracle.c)
            emit_out = (d_out[:out_bytes].cpu().numpy(), toff) if not args.no_cpu_baseline else None
            del d_out, d_toff

        # ---- parity against the oracle (this rank's whole shard, bit for bit) and the algorithmic bytes it counts
        a_char, parity, cpu, mismatch = None, None, None, None
        if not args.no_cpu_baseline:
            from oracle import cbind
            orc = cbind.OraclePredictor(raw, cfg["tags"])
            t = time.perf_counter()
            if (primary or cfg_id == 1) and self.rank == 0:
                # the CPU baseline, on RANK 0's host cores only (the other ranks run the parity pass alone, so that at N = 8 nobody
                # waits on seven redundant baselines): a bounded sample of the same workload -- one pass over (at most) the
                # first 100 K sentences on ONE thread (the reference is single-threaded), then the whole shard on this rank's share
                # of the host threads (that pass is also the parity check's reference); repeated while it stays within ~10 s
                n1 = min(S, 100_000)
                # (round 5) the TIMED passes walk the char scorer's automaton as a DOUBLE ARRAY -- what the reference's daachorse matcher is
                # (char_scorer/boundary_scorer.rs:76-99); the checker's hash-table automaton, which the parity pass below uses, is timed beside
                # it (`hash_automaton_value`): same scores (tests/test_oracle_c_kat.py), 16 bytes per state instead of 32+ per transition
                sub = (utf8[:int(boff[n1])], boff[:n1 + 1])
                nb1 = int(ooff[n1])
                # the first pass over the shard is the parity check's reference; its outputs are fresh arrays, so it also pays their
                # page faults (3 GB for configs[2], first touched from every thread) -- it is NOT timed.  The timed passes write the same
                # arrays again: what they measure is the algorithm at memory-resident size.
                # ONE thread first (the reference is single-threaded), before the all-core passes take the cores' boost clocks away
                c_out = orc.predict_batch(*sub, nthreads=1)[:3]
                secs_1, _, _ = orc.baseline_timed(*sub, c_out, nthreads=1, reps=3, replicate=False)
                t1 = min(secs_1[1:])
                # ... and on a sample that is NOT cache-resident (the first million sentences, one pass after a warm-up over a tenth of them): the
                # regime the whole batch runs in -- the automaton walk is a chain of dependent loads, and past a few hundred thousand sentences the
                # states they visit no longer fit the caches (by_batch_size below)
                single_big = None
                if S >= 2_000_000:
                    n_big = 1_000_000
                    sub_b = (utf8[:int(boff[n_big])], boff[:n_big + 1])
                    b_out = orc.predict_batch(utf8[:int(boff[n_big // 10])], boff[:n_big // 10 + 1], nthreads=1)[:3]
                    del b_out
                    b_out = (np.zeros(int(ooff[n_big]), np.int32), np.zeros(int(ooff[n_big]), np.uint8), ooff[:n_big + 1].copy())
                    secs_b, _, _ = orc.baseline_timed(*sub_b, b_out, nthreads=1, reps=1, replicate=False)
                    single_big = int(ooff[n_big]) / secs_b[0]
                    del b_out
                o_scores, o_labels, o_ooff, a_char = orc.predict_batch(utf8, boff, nthreads=self.ncores)
                t = time.perf_counter()
                orc.predict_batch(utf8, boff, nthreads=self.ncores, out=(o_scores, o_labels, o_ooff), pin=True)
                t_hash = time.perf_counter() - t
                # (round 6) the timed passes run on a POOL of pinned workers that lives for all of them -- a pass is timed from
                # the first worker's start to the last one's end, thread start-up is in none -- with the data every char walks (automaton, codes,
                # weight records and vectors, type table) REPLICATED per NUMA node; the same pool without the replicas is timed beside it
                reps_n = 3 if S > 2_000_000 else 6
                secs_n, _, nodes = orc.baseline_timed(utf8, boff, (o_scores, o_labels, o_ooff), nthreads=self.ncores, reps=reps_n, replicate=True, huge_pages=True)
                secs_4k, _, _ = orc.baseline_timed(utf8, boff, (o_scores, o_labels, o_ooff), nthreads=self.ncores, reps=2, replicate=True)
                secs_shared, _, _ = orc.baseline_timed(utf8, boff, (o_scores, o_labels, o_ooff), nthreads=self.ncores, reps=2, replicate=False)
                tn = min(secs_n[1:]) if len(secs_n) > 1 else secs_n[0]
                secs_c, _, _ = orc.baseline_timed(*sub, c_out, nthreads=self.ncores, reps=12, replicate=True, huge_pages=True)   # cache-resident: configs[1]'s size on every thread
                tc = min(secs_c[1:])
                del c_out
                # what stops the pool from scaling: the same pool over growing prefixes of the batch -- the walk is a chain of dependent loads, and
                # once the sentences' share of the automaton no longer fits the cores' caches every char waits for DRAM
                by_size = []
                for n_k in (100_000, 300_000, 1_000_000, 3_000_000):
                    if n_k >= S:
                        break
                    sub_k = (utf8[:int(boff[n_k])], boff[:n_k + 1])
                    k_out = (o_scores[:int(ooff[n_k])], o_labels[:int(ooff[n_k])], ooff[:n_k + 1].copy())
                    secs_k, _, _ = orc.baseline_timed(*sub_k, k_out, nthreads=self.ncores, reps=4, replicate=True, huge_pages=True)
                    by_size.append({"sentences": n_k, "value": int(ooff[n_k]) / min(secs_k[1:])})
                by_size.append({"sentences": S, "value": nb / tn})
                cpu = {"value": nb / tn, "unit": "boundaries/s", "cores": self.ncores, "kind": "port", "numa_nodes": nodes,
                       "single_thread_value": nb1 / t1, "scaling_vs_one_thread": (nb / tn) / (nb1 / t1),
                       "single_thread_memory_resident_value": single_big, "scaling_vs_one_thread_memory_resident": ((nb / tn) / single_big) if single_big else None, "cache_resident_value": nb1 / tc,
                       "shared_tables_value": nb / min(secs_shared), "small_pages_value": nb / min(secs_4k), "hash_automaton_value": nb / t_hash, "pass_seconds": [round(x, 4) for x in secs_n], "by_batch_size": by_size, "cpu": cpu_model_name(),
                       "sample": "rank 0's shard of this workload (%d sentences): best of %d passes (the first one apart) on a pool of %d pinned threads that lives for all "
                                 "of them -- a pass is timed from its first worker's start to its last one's end -- into pre-faulted outputs, the char scorer's automaton as a "
                                 "double array (what the reference's matcher is), the tables replicated per NUMA node (%d) on 2 MB pages (madvise); `small_pages_value`: the "
                                 "replicas on 4 KB pages; `shared_tables_value`: the same pool, one copy of the tables; `hash_automaton_value`: one pass with the checker's hash-table automaton, a thread per call; `cache_resident_value`: the "
                                 "first %d sentences, best of 11 passes; `by_batch_size`: the pool over growing prefixes of the batch; `single_thread_value`: the first %d sentences on 1 thread, "
                                 "best of 2, before any all-core pass; `single_thread_memory_resident_value`: the first million sentences once on 1 thread (C restatement of the reference "
                                 "algorithm, not the Rust binary)" % (S, len(secs_n) - 1, self.ncores, nodes, n1, n1)}
            else:
                o_scores, o_labels, _, a_char = orc.predict_batch(utf8, boff, nthreads=self.ncores)
            g_scores = d_scores[:nb].cpu().numpy()
            g_labels = d_labels[:nb].cpu().numpy()
            mismatch = None     # where the first difference is, not only that there is one

            def first_diff(kind, got, want, rows_are_chars=False):
                bad = np.flatnonzero(got != want)
                if not len(bad):
                    return None
                k = int(bad[0])
                if rows_are_chars:   # row of char c of sentence i = ooff[i] + i + c
                    i = int(np.searchsorted(ooff + np.arange(S + 1, dtype=np.uint64), k, side="right") - 1)
                else:
                    i = int(np.searchsorted(ooff, k, side="right") - 1)
                tx = bytes(utf8[int(boff[i]):int(boff[i + 1])]).decode("utf-8", "replace")
                return {"what": kind, "index": k, "sentence": first + i, "text": tx[:80], "gpu": int(got[k]), "oracle": int(want[k]), "differing": int(len(bad))}
            ok = bool(np.array_equal(g_scores, o_scores) and np.array_equal(g_labels, o_labels))
            if not ok:
                mismatch = first_diff("boundary score", g_scores, o_scores) or first_diff("label", g_labels, o_labels)
            o_tags = o_models = None
            if nt:   # EVERY token of the shard: Sentence::fill_tags on the labels the GPU predicted (they are the oracle's: checked above)
                got = d_tags[:(nb + S) * nt].cpu().numpy().reshape(nb + S, nt)
                o_tags, _, o_models = orc.fill_tags_batch(utf8, boff, ooff, g_labels, nthreads=self.ncores, want_scores=False)
                tag_ok = bool(np.array_equal(got, o_tags)) and bool(tags_info["records_expand_to_the_dense_array"])
                if not tag_ok and mismatch is None:
                    mismatch = first_diff("tags of the token ending at this char (1 = some slot differs)", (got != o_tags).any(axis=1).astype(np.int8),
                                          np.zeros(nb + S, np.int8), rows_are_chars=True)
                tags_info["parity"] = tag_ok
                tags_info["tokens_checked"] = int((g_labels == 1).sum()) + S
                tags_info["tokens_with_a_tag_model"] = int((o_models >= 0).sum())
                ok = ok and tag_ok
                del got
            if emit_out is not None:   # the writer's whole output, byte for byte
                o_text, o_toff = orc.write_tokenized_batch(utf8, boff, ooff, g_labels, o_tags, o_models, nthreads=self.ncores)
                e_ok = bool(np.array_equal(emit_out[1], o_toff) and np.array_equal(emit_out[0], o_text))
                if not e_ok and mismatch is None:
                    bad = np.flatnonzero(emit_out[1] != o_toff)
                    i = max(int(bad[0]) - 1, 0) if len(bad) else int(np.searchsorted(o_toff, np.flatnonzero(emit_out[0] != o_text)[0], side="right") - 1)
                    mismatch = {"what": "tokenized text", "sentence": first + i,
                                "gpu": bytes(emit_out[0][int(emit_out[1][i]):int(emit_out[1][min(i + 1, S)])]).decode("utf-8", "replace")[:120],
                                "oracle": bytes(o_text[int(o_toff[i]):int(o_toff[i + 1])]).decode("utf-8", "replace")[:120]}
                emit_info["parity"] = e_ok
                emit_info["bytes_checked"] = int(len(o_text))
                ok = ok and e_ok
                emit_out = None
            del o_tags, o_models
            parity = self.all_true(ok)
            if mismatch is not None:
                sys.stderr.write("bench.py: PARITY MISMATCH on rank %d: %s\n" % (self.rank, json.dumps(mismatch, ensure_ascii=False)))
        del d_tags

        # ---- end to end over PCIe: pinned caller buffers through the pipelined host path (N = 1, primary only)
        e2e = None
        if e2e_leg and self.world == 1 and not args.no_e2e:
            keep = [api.PinnedArray((nbytes,), np.uint8), api.PinnedArray((max(nb, 1),), np.int32), api.PinnedArray((max(nb, 1),), np.uint8)]
            keep[0].array[:] = utf8
            for _ in range(2):
                api.predict_packed_sharded([pred], keep[0].array, boff, out_offsets=ooff, scores=keep[1].array, labels=keep[2].array)
            k = max(5, min(steps, 20))

            def median_time(fn, n):   # per-call wall clock, median: the host's other threads (the CPU baseline just ran on all of them) show up as outliers
                ts = []
                for _ in range(n):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
                return float(np.median(ts))
            dt = median_time(lambda: api.predict_packed_sharded([pred], keep[0].array, boff, out_offsets=ooff, scores=keep[1].array, labels=keep[2].array), k)
            bytes_in, bytes_out = nbytes + 16 * (S + 1), 5 * nb
            e2e = {"boundaries_per_s": nb / dt, "ms_per_batch": 1e3 * dt, "h2d_GBps": bytes_in / dt / 1e9, "d2h_GBps": bytes_out / dt / 1e9,
                   "pcie_peak_GBps_per_direction": PCIE_GBS, "frac_of_pcie": max(bytes_in, bytes_out) / dt / 1e9 / PCIE_GBS,
                   "pcie_measured_both_ways_GBps": PCIE_DUPLEX_GBS, "frac_of_both_ways": (bytes_in + bytes_out) / dt / 1e9 / PCIE_DUPLEX_GBS,
                   "parity": bool(parity is None or (np.array_equal(keep[1].array[:nb], o_scores) and np.array_equal(keep[2].array[:nb], o_labels))),
                   "path": "vpt_predict_batch: pinned host buffers (vpt_host_alloc); batches up to 16 M chars in 512 K-char chunks over 4 "
                           "in-order lanes, larger ones in 4 M-char chunks on three streams with events"}
            # the same batch when only the labels come back (what a tokenizer needs: 1 of the 5 bytes per boundary)
            keep[2].array[:] = 9
            dt = median_time(lambda: api.predict_packed_sharded([pred], keep[0].array, boff, out_offsets=ooff, labels=keep[2].array, want_scores=False), k)
            e2e["labels_only"] = {"boundaries_per_s": nb / dt, "ms_per_batch": 1e3 * dt, "h2d_GBps": bytes_in / dt / 1e9,
                                  "frac_of_pcie": bytes_in / dt / 1e9 / PCIE_GBS,
                                  "parity": bool(parity is None or np.array_equal(keep[2].array[:nb], o_labels))}
            # lines in, tokenized lines out (vpt_tokenize_batch: what the reference's CLI loop does, predict/src/main.rs:122-176):
            # only text crosses the link, in one piece each way
            tk = [api.PinnedArray((3 * nbytes + 64,), np.uint8), api.PinnedArray((S + 1,), np.uint64)]
            for _ in range(2):
                tok_text, tok_off = pred.tokenize_packed(keep[0].array, boff, text_out=tk[0].array, offsets_out=tk[1].array)
            dt = median_time(lambda: pred.tokenize_packed(keep[0].array, boff, text_out=tk[0].array, offsets_out=tk[1].array), k)
            e2e["tokenize"] = {"ms_per_batch": 1e3 * dt, "chars_per_s": (nb + S) / dt, "h2d_GBps": (nbytes + 8 * (S + 1)) / dt / 1e9,
                               "d2h_GBps": (len(tok_text) + 8 * (S + 1)) / dt / 1e9, "out_bytes": int(len(tok_text)),
                               "frac_of_both_ways": (nbytes + len(tok_text) + 16 * (S + 1)) / dt / 1e9 / PCIE_DUPLEX_GBS,
                               "path": "vpt_tokenize_batch: chunks of a sixth of the batch (2 .. 8 MB); per chunk copy in, char count + tile search on a stream of their own, ONE scoring launch that writes the tokenized text (chunks chain into one output), copy out as soon as the chunk's event has fired"}
            del tk
            # ten of these batches as one call: what the pipeline does once its start-up no longer counts
            rep = 10
            big = [api.PinnedArray((nbytes * rep,), np.uint8), api.PinnedArray((nb * rep,), np.int32), api.PinnedArray((nb * rep,), np.uint8)]
            big[0].array[:] = np.tile(utf8, rep)
            boff_big = np.concatenate([boff[:-1] + np.uint64(r * nbytes) for r in range(rep)] + [np.array([rep * nbytes], dtype=np.uint64)])
            ooff_big = np.concatenate([ooff[:-1] + np.uint64(r * nb) for r in range(rep)] + [np.array([rep * nb], dtype=np.uint64)])
            for _ in range(2):
                api.predict_packed_sharded([pred], big[0].array, boff_big, out_offsets=ooff_big, scores=big[1].array, labels=big[2].array)
            dt = median_time(lambda: api.predict_packed_sharded([pred], big[0].array, boff_big, out_offsets=ooff_big, scores=big[1].array, labels=big[2].array), 7)
            e2e["large_batch"] = {"sentences": S * rep, "boundaries_per_s": nb * rep / dt, "ms_per_batch": 1e3 * dt, "d2h_GBps": bytes_out * rep / dt / 1e9,
                                  "h2d_GBps": bytes_in * rep / dt / 1e9, "frac_of_pcie": bytes_out * rep / dt / 1e9 / PCIE_GBS,
                                  "frac_of_both_ways": (bytes_in + bytes_out) * rep / dt / 1e9 / PCIE_DUPLEX_GBS,
                                  "parity": bool(parity is None or (np.array_equal(big[1].array[(rep - 1) * nb:], o_scores) and np.array_equal(big[2].array[:nb], o_labels)))}
            del big
            del keep

        if self.rank != 0:
            return None
        out = {
            "workload": "%s: %s model, %d sentences x %d..%d chars, inputs resident in HBM" % (cfg["name"], model_name, S_total, cfg["min_len"], cfg["max_len"]),
            "value": total_boundaries * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
            "tokenizer_model": model_name, "sentences_per_gpu": S, "boundaries_per_gpu": nb, "text_bytes_per_gpu": nbytes,
            "char_ngrams": info["n_char_ngrams"], "dict_words": info["n_dict_words"], "tag_models": info["n_tag_models"],
            "table_bytes": info["device_table_bytes"], "hot_table_bytes": info["hot_table_bytes"],
            "packed_tables": bool(info["packed"]), "tiles": n_tiles, "tile_plan": "%s of %d flat positions" % (plan["kind"], plan["tile_flat"]),
            "create_s": round(create_s, 2), "tables_broadcast_s": round(bcast_s, 3) if self.world > 1 else None, "synth_s": round(synth_s, 2),
        }
        if cfg.get("nonbmp_share"):
            out["chars_outside_the_bmp"] = float(np.count_nonzero(utf8 >= 0xF0)) / max(nb + S, 1)
        if per_rank is not None:
            out["per_rank"] = per_rank
        if phases is not None:
            tot = float(sum(phases[:7])) or 1.0
            out["phase_share"] = dict(zip(["stage", "sentence_starts", "scan_decode", "classify", "patterns", "barrier", "output"], [round(p / tot, 4) for p in phases[:7]]))
        # roofline of the dominant kernel: algorithmic bytes per launch / its duration
        a_stream = nbytes + 5 * nb + 16 * S   # text + i32 score + u8 label per boundary + two u64 offsets per sentence
        a_type = 4 * nb                        # one type-window table word per boundary (the reference's cache form)
        kernel_name = "score_tiles_kernel" if plan["kind"] == "general kernels" else "score_tiles_fast_kernel"   # which launch the events bracketed
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": kernel_name,
                "kernel_ms": kernel_ms, "kernel_ms_mean": kernel_ms_mean, "kernel_ms_min": float(np.min(ktimes)) if len(ktimes) else None,
                "timed_launches": int(len(ktimes)), "timing": "HIP events on the launch stream around the kernel; median of the timed launches"}
        if a_char is not None and kernel_ms > 0:
            a = a_stream + a_char + a_type
            roof.update({"achieved": a / (kernel_ms * 1e-3) / 1e9, "frac": a / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": measured_traffic(kernel_name, model_name, cfg["name"]) if self.world == 1 and not self.args.sentences else None,
                         "algorithmic_bytes_per_launch": a, "bytes_per_boundary": a / max(nb, 1),
                         "a_stream": a_stream, "a_char": a_char, "a_type": a_type})
        if self.world == 1 and kernel_name == "score_tiles_fast_kernel" and not args.no_cpu_baseline:   # (profiling runs: no second instance of the kernel in the counters)
            node_reads = None
            try:   # a diagnostics launch of the same batch counts the node reads (untimed; the instance that counts is slower)
                os.environ["VPT_PROFILE_PHASES"] = "1"
                counted = api.DeviceBatch(pred)
                os.environ.pop("VPT_PROFILE_PHASES", None)
                counted.set_max_sentence_chars(max_chars)
                counted.predict(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, max_bytes, d_scores.data_ptr(), d_labels.data_ptr(), stream)
                counted.sync()
                node_reads = counted.node_reads()
                del counted
            except Exception as e:   # noqa: BLE001 -- a diagnostic: the line does not depend on it
                sys.stderr.write("bench.py: node reads not counted: %s\n" % e)
            finally:
                os.environ.pop("VPT_PROFILE_PHASES", None)
            wl = max(3, int(info["char_window"]), int(info["type_window"]))
            entry = measured_entry(kernel_name, model_name, cfg["name"]) if not self.args.sentences else None
            roof["gather"] = gather_block(entry, node_reads, nb + S, wl, kernel_ms)
        out["roofline"] = roof
        out["parity"] = parity
        if not args.no_cpu_baseline:
            out["parity_covers"] = "every i32 score and label of the shard%s%s" % (", every token's tags" if nt else "", ", the writer's whole output" if emit_info is not None else "")
            if mismatch is not None:
                out["first_mismatch"] = mismatch
        if tags_info is not None:
            out["tags"] = tags_info
        if e2e is not None:
            out["e2e"] = e2e
        if emit_info is not None:
            out["emit"] = emit_info
        if cpu is not None:
            out["cpu_baseline"] = cpu
        return out


def compact_row(w):
    """One workload as a row of the result line."""
    roof = w.get("roofline") or {}
    a, tr = roof.get("algorithmic_bytes_per_launch"), roof.get("traffic")
    r3 = lambda x, n=3: None if x is None else round(float(x), n)   # noqa: E731
    row = {"name": w["workload"].split(":")[0], "model": w.get("tokenizer_model"), "value_G": r3(w["value"] / 1e9), "ms_per_step": r3(w["ms_per_step"], 4),
           "kernel": roof.get("kernel"), "kernel_ms": r3(roof.get("kernel_ms"), 4), "frac": r3(roof.get("frac"), 4), "bytes_per_boundary": r3(roof.get("bytes_per_boundary"), 1),
           "traffic_ratio": r3(tr / a) if (a and tr) else None, "parity": w.get("parity"), "packed": w.get("packed_tables"), "tile_plan": (w.get("tile_plan") or "").split(" of ")[0]}
    if w.get("tags"):
        row["tags_ms"], row["tags_dense_ms"], row["tags_parity"] = r3(w["tags"]["ms_per_step"], 4), r3(w["tags"].get("dense_ms_per_step"), 4), w["tags"].get("parity")
    if w.get("emit"):
        row["emit_ms"], row["emit_frac"], row["emit_parity"] = r3(w["emit"]["ms_per_step"], 4), r3(w["emit"]["frac_of_hbm"], 4), w["emit"].get("parity")
    if w.get("e2e"):
        row["e2e_ms"], row["e2e_frac_both_ways"] = r3(w["e2e"]["ms_per_batch"], 4), r3(w["e2e"]["frac_of_both_ways"])
        row["tokenize_ms"], row["tokenize_Gchars"] = r3(w["e2e"]["tokenize"]["ms_per_batch"], 4), r3(w["e2e"]["tokenize"]["chars_per_s"] / 1e9)
    if "chars_outside_the_bmp" in w:
        row["chars_outside_the_bmp"] = r3(w["chars_outside_the_bmp"], 4)
    if w.get("first_mismatch"):
        row["first_mismatch"] = w["first_mismatch"]
    return row


def shorten(x):
    """The line as printed: floats to 6 significant digits, prose (`kernels`, `path`, `timing`: explanations that live in --detail-out and
    DESIGN.md) cut to 160 chars."""
    if isinstance(x, float):
        return float("%.6g" % x)
    if isinstance(x, dict):
        return {k: ((v[:157] + "...") if isinstance(v, str) and len(v) > 160 and k in ("kernels", "path", "timing", "sample", "workloads_columns") else shorten(v)) for k, v in x.items()}
    if isinstance(x, list):
        return [shorten(v) for v in x]
    return x


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 7, 8],
                    help="BASELINE.json configs index (0: configs[2] as the value at every N, plus -- at N = 1 -- the others as `workloads`)")
    ap.add_argument("--scale-config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 7, 8],
                    help="the workload of a scaling run other than configs[2]: e.g. 4 = BASELINE configs[4] (tags on, 8..512-char sentences), sharded by chars over the "
                         "N ranks with the per-rank balance in `per_rank` (same as --config K --quick; passes through --scale-sweep / --dry-scale)")
    ap.add_argument("--quick", action="store_true", help="the primary workload only")
    ap.add_argument("--sentences", type=int, default=0, help="override the config's sentence count (diagnostics; traffic is then not reported)")
    ap.add_argument("--model-scale", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle: no parity, no roofline, no cpu_baseline (profiling runs)")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-emit", action="store_true")
    ap.add_argument("--phases", action="store_true", help="diagnostics: per-phase shader cycles of the scoring kernel (slows it)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the per-workload detail goes (the line itself carries one compact row per workload and stays below 12 KB)")
    ap.add_argument("--in-process", action="store_true", help="N > 1: skip torch.distributed, drive the N devices from this process (the fallback path)")
    ap.add_argument("--scale-sweep", default="", help="e.g. 1,2,4,8: run the primary workload at every N of the list (one line each, as --gpus N "
                    "would print it, plus `scaling_efficiency` against the N = 1 value of the same invocation)")
    ap.add_argument("--dry-scale", action="store_true", help="pre-flight of the scaling job: --scale-sweep 1,2,4,8 on whatever this box has -- N ranks on "
                    "N devices over RCCL when it has them, else every rank on device 0 over gloo -- and a check of every line (exit code 1 when one fails)")
    args = ap.parse_args(argv)
    if args.scale_config:
        args.config, args.quick = args.scale_config, True
    if args.gpus < 1:
        die("--gpus must be at least 1")
    return args


def visible_devices() -> int:
    import torch
    return torch.cuda.device_count()


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: check the devices, run the N-rank job, fall back to the
    in-process clones when it fails.  Returns the exit code."""
    import socket
    n_vis = visible_devices()
    one_dev = bool(os.environ.get("VPT_BENCH_ONE_DEVICE"))
    if n_vis < (1 if one_dev else args.gpus):
        die("--gpus %d but only %d HIP device(s) are visible on this node (hipGetDeviceCount): refusing to print a line for fewer GPUs than "
            "asked for (test hook for a 1-GPU box: VPT_BENCH_ONE_DEVICE=1 VPT_BENCH_BACKEND=gloo)" % (args.gpus, n_vis))
    reason = "--in-process"
    if not args.in_process:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, VPT_BENCH_LAUNCH="bench.py launched its %d ranks itself (torch.distributed.run)" % args.gpus)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this host driver
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
        sys.stderr.flush()
        job = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
        out = job.stdout.decode("utf-8", "replace")
        lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
        if job.returncode == 0 and lines:
            sys.stdout.write(out)
            sys.stdout.flush()
            return 0
        sys.stderr.write(out)
        reason = "the %d-rank torch.distributed job exited with code %d%s" % (args.gpus, job.returncode, "" if lines else " and printed no result line")
        sys.stderr.write("bench.py: %s; falling back to one process driving %d devices\n" % (reason, args.gpus))
        sys.stderr.flush()
    return run_in_process(args, reason)


def run_in_process(args, reason: str) -> int:
    """The fallback curve: ONE process, N devices.  The predictor is compiled on device 0 and cloned device to device
    (vpt_predictor_clone_to_device: hipMemcpyPeer over xGMI); every device holds its character-balanced shard of the workload in its own
    HBM and is driven by a host thread + stream of its own.  Timed like the N-rank job: all threads start together, each runs K steps
    and waits for its device; the job's time is first start to last finish."""
    import threading
    import torch
    from vaporetto_amd import api
    N = args.gpus
    one_dev = bool(os.environ.get("VPT_BENCH_ONE_DEVICE"))
    n_vis = visible_devices()
    if n_vis < (1 if one_dev else N):
        die("--gpus %d but only %d HIP device(s) are visible on this node" % (N, n_vis))
    devs = [0] * N if one_dev else list(range(N))
    cfg_id = args.config or 2
    cfg = CONFIGS[cfg_id]
    ncores = max(1, os.cpu_count() or 1)
    raw, model_name = load_model_bytes(cfg["kind"], args.model_scale)
    t = time.perf_counter()
    p0 = api.Predictor(api.Model.read_slice(raw)[0], bool(cfg["tags"]), device=devs[0])
    nt = p0.n_tags() if cfg["tags"] else 0
    create_s = time.perf_counter() - t
    t = time.perf_counter()
    preds = [p0] + [p0.clone_to_device(d) for d in devs[1:]]
    clone_s = time.perf_counter() - t
    info = p0.info()
    shards, S_total, synth_s = [], 0, 0.0
    for r in range(N):
        t = time.perf_counter()
        utf8, boff, ooff, first, S_total = make_shard(cfg, raw, r, N, ncores, args.sentences)
        synth_s += time.perf_counter() - t
        torch.cuda.set_device(devs[r])
        dev = torch.device("cuda", devs[r])
        S, nb = len(boff) - 1, int(ooff[-1])
        sh = dict(utf8=utf8, boff=boff, ooff=ooff, S=S, nb=nb, nbytes=int(boff[-1]), dev=dev,
                  max_bytes=int(np.max(np.diff(boff.astype(np.int64)))) if S else 1,
                  d_text=torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev),
                  d_boff=torch.from_numpy(boff.astype(np.int64)).to(dev), d_ooff=torch.from_numpy(ooff.astype(np.int64)).to(dev),
                  d_scores=torch.empty(nb + 1, dtype=torch.int32, device=dev), d_labels=torch.empty(nb + 1, dtype=torch.uint8, device=dev),
                  d_tags=torch.empty((nb + S) * nt + 1, dtype=torch.int32, device=dev) if nt else None,
                  stream=torch.cuda.Stream(device=dev), batch=api.DeviceBatch(preds[r], timing=True))
        sh["batch"].set_max_sentence_chars((int(np.max(np.diff(ooff.astype(np.int64)))) + 1) if S else 1)
        shards.append(sh)
    start = threading.Barrier(N)
    t_start, t_end, errors = [0.0] * N, [0.0] * N, []

    def work(r):
        try:
            sh = shards[r]
            torch.cuda.set_device(sh["dev"])
            st = sh["stream"].cuda_stream

            def step():
                sh["batch"].predict(sh["d_text"].data_ptr(), sh["d_boff"].data_ptr(), sh["d_ooff"].data_ptr(), sh["S"], sh["nb"], sh["max_bytes"],
                                    sh["d_scores"].data_ptr(), sh["d_labels"].data_ptr(), st)
                if nt:   # configs[4]: the step is the whole tagging job
                    sh["batch"].fill_tags(sh["d_text"].data_ptr(), sh["d_boff"].data_ptr(), sh["d_ooff"].data_ptr(), sh["S"], sh["nb"], sh["d_labels"].data_ptr(),
                                          0, st)   # (the tag records; expanded for the parity check below)
            for _ in range(args.warmup):
                step()
            sh["batch"].sync()
            sh["batch"].kernel_ms()
            sh["stream"].synchronize()
            start.wait()
            t_start[r] = time.perf_counter()
            for _ in range(args.steps):
                step()
            sh["stream"].synchronize()
            t_end[r] = time.perf_counter()
            sh["batch"].sync()
        except BaseException as e:   # noqa: a failed thread must not leave the others at the barrier
            errors.append(e)
            start.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(N)]
    [th.start() for th in threads]
    [th.join() for th in threads]
    if errors:
        die("in-process run failed: %r" % (errors[0],), 1)
    elapsed = max(t_end) - min(t_start)
    total_nb = sum(sh["nb"] for sh in shards)
    ktimes = shards[0]["batch"].kernel_times()
    kernel_ms_mean, n_tiles = shards[0]["batch"].kernel_ms()
    kernel_ms = float(np.median(ktimes)) if len(ktimes) else kernel_ms_mean
    parity, cpu, a_char = None, None, None
    if not args.no_cpu_baseline:
        from oracle import cbind
        orc = cbind.OraclePredictor(raw, bool(cfg["tags"]))
        parity = True
        for r, sh in enumerate(shards):
            t = time.perf_counter()
            o_scores, o_labels, _, ac = orc.predict_batch(sh["utf8"], sh["boff"], nthreads=ncores)
            tn = time.perf_counter() - t
            torch.cuda.set_device(sh["dev"])
            ok = bool(np.array_equal(sh["d_scores"][:sh["nb"]].cpu().numpy(), o_scores) and np.array_equal(sh["d_labels"][:sh["nb"]].cpu().numpy(), o_labels))
            if nt:
                sh["batch"].expand_tags(sh["S"], sh["nb"], sh["d_tags"].data_ptr(), 0)
                sh["batch"].sync()
                o_tags, _, _ = orc.fill_tags_batch(sh["utf8"], sh["boff"], sh["ooff"], o_labels, nthreads=ncores, want_scores=False)
                ok = ok and bool(np.array_equal(sh["d_tags"][:(sh["nb"] + sh["S"]) * nt].cpu().numpy().reshape(sh["nb"] + sh["S"], nt), o_tags))
            parity = parity and ok
            if r == 0:
                a_char = ac
                n1 = min(sh["S"], 100_000)
                t = time.perf_counter()
                orc.predict_batch(sh["utf8"][:int(sh["boff"][n1])], sh["boff"][:n1 + 1], nthreads=1)
                t1 = time.perf_counter() - t
                cpu = {"value": sh["nb"] / tn, "unit": "boundaries/s", "cores": ncores, "kind": "port", "single_thread_value": int(sh["ooff"][n1]) / t1,
                       "cpu": cpu_model_name(),
                       "sample": "device 0's shard of this workload (%d sentences): 1 pass on %d threads; its first %d sentences once on 1 thread (C "
                                 "restatement of the reference algorithm with a hash-table automaton, not the Rust binary: a lower bound for it)" % (sh["S"], ncores, n1)}
    sh0 = shards[0]
    a_stream, a_type = sh0["nbytes"] + 5 * sh0["nb"] + 16 * sh0["S"], 4 * sh0["nb"]
    kernel_name = "score_tiles_kernel" if shards[0]["batch"].last_plan()["kind"] == "general kernels" else "score_tiles_fast_kernel"
    roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": kernel_name,
            "kernel_ms": kernel_ms, "kernel_ms_mean": kernel_ms_mean, "kernel_ms_min": float(np.min(ktimes)) if len(ktimes) else None,
            "timed_launches": int(len(ktimes)), "timing": "HIP events on device 0's launch stream around the kernel; median of the timed launches (device 0's shard)"}
    if a_char is not None and kernel_ms > 0:
        a = a_stream + a_char + a_type
        roof.update({"achieved": a / (kernel_ms * 1e-3) / 1e9, "frac": a / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": a,
                     "bytes_per_boundary": a / max(sh0["nb"], 1), "a_stream": a_stream, "a_char": a_char, "a_type": a_type})
    line = {
        "metric": "boundary scores/sec", "value": total_nb * args.steps / elapsed, "unit": "boundaries/s", "n_gpus": N, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong" if cfg_id == 2 else "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "%s: %s model, %d sentences x %d..%d chars, inputs resident in HBM" % (cfg["name"], model_name, S_total, cfg["min_len"], cfg["max_len"]),
                   "tokenizer_model": model_name, "sentences_per_gpu": sh0["S"], "boundaries_per_gpu": sh0["nb"], "text_bytes_per_gpu": sh0["nbytes"],
                   "char_ngrams": info["n_char_ngrams"], "dict_words": info["n_dict_words"], "tag_models": info["n_tag_models"],
                   "table_bytes": info["device_table_bytes"], "hot_table_bytes": info["hot_table_bytes"], "packed_tables": bool(info["packed"]),
                   "tiles": n_tiles, "create_s": round(create_s, 2), "tables_broadcast_s": round(clone_s, 3), "synth_s": round(synth_s, 2),
                   "sharding": "contiguous sentence ranges balanced by chars over %d device(s), no data-path collective" % N,
                   "chars_per_device": [sh["nb"] + sh["S"] for sh in shards],
                   "hip_devices_visible": n_vis, "world_size": N, "collective_backend": None, "tables_broadcast": "vpt_predictor_clone_to_device (hipMemcpyPeer)",
                   "launch": "in-process fallback: one process, vpt_predictor_clone_to_device (hipMemcpyPeer) to %d device(s), one host thread + stream "
                             "per device (%s)%s" % (N, reason, "; VPT_BENCH_ONE_DEVICE: every shard on device 0" if one_dev else "")},
        "parity": parity, "roofline": roof, "cpu_baseline": cpu,
        "dist_failure": reason,      # top level: this line is NOT the N-rank RCCL job the caller asked for
    }
    print(json.dumps(line))
    sys.stdout.flush()
    return 0


def scale_sweep(args) -> int:
    """--scale-sweep / --dry-scale: the primary workload at every N of the list, each as a job of its own (`python bench.py --gpus N ...`, i.e.
    exactly what the driver runs), the lines relayed with `scaling_efficiency` = value(N) / (N x value(1)).  --dry-scale also checks them."""
    ns = [1, 2, 4, 8] if args.dry_scale and not args.scale_sweep else [int(x) for x in args.scale_sweep.split(",") if x.strip()]
    if not ns or any(n < 1 for n in ns):
        die("--scale-sweep wants a list of GPU counts, e.g. 1,2,4,8")
    n_vis = visible_devices()
    passthrough, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a in ("--scale-sweep", "--gpus"):
            skip = True
            continue
        if a == "--dry-scale" or a.startswith("--scale-sweep=") or a.startswith("--gpus="):
            continue
        passthrough.append(a)
    lines, failures = [], []
    for n in ns:
        env = dict(os.environ)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        if n > n_vis:
            if not args.dry_scale:
                die("--scale-sweep: %d GPUs asked for, %d visible (--dry-scale puts the ranks on device 0)" % (n, n_vis))
            env["VPT_BENCH_ONE_DEVICE"] = "1"      # every rank on device 0: RCCL refuses two ranks on one device, so gloo
            env["VPT_BENCH_BACKEND"] = "gloo"
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n)] + passthrough
        sys.stderr.write("bench.py: sweep N = %d: %s\n" % (n, " ".join(cmd)))
        sys.stderr.flush()
        t0 = time.perf_counter()
        job = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
        wall = time.perf_counter() - t0
        got = [l for l in job.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{") and '"metric"' in l]
        if job.returncode != 0 or len(got) != 1:
            failures.append("N = %d: exit code %d, %d result line(s)" % (n, job.returncode, len(got)))
            continue
        line = json.loads(got[0])
        line["sweep_wall_s"] = round(wall, 1)
        lines.append(line)
    base = next((l for l in lines if l["n_gpus"] == 1), None)
    for l in lines:
        l["scaling_efficiency"] = (l["value"] / (l["n_gpus"] * base["value"])) if base else None
        if args.dry_scale:
            n = l["n_gpus"]
            one_dev = "VPT_BENCH_ONE_DEVICE" in l["config"]["launch"]
            checks = {"n_gpus is what was asked for": n in ns and l["config"]["world_size"] == n,
                      "parity": l["parity"] is True,
                      "the same workload as N = 1": base is None or l["config"]["workload"] == base["config"]["workload"],
                      "strong scaling of one workload": l["scaling"] == "strong" or bool(args.config and args.config != 2),
                      "per-rank figures": n == 1 or (len(l.get("per_rank") or []) == n and sum(r["sentences"] for r in l["per_rank"]) > 0)}
            if base is not None and n > 1 and l.get("per_rank"):
                k1 = base["roofline"]["kernel_ms"]
                ks = [(r["kernel_ms_solo"] if one_dev else r["kernel_ms"]) for r in l["per_rank"]]
                if all(k is not None for k in ks):
                    # N shards of one batch: the ranks' kernels add up to the N = 1 kernel (10 %, plus a tile round per rank on small batches)
                    l["kernel_ms_sum_over_ranks"] = round(sum(ks), 4)
                    checks["the ranks' kernels add up to the N = 1 kernel within 10 %"] = abs(sum(ks) - k1) <= 0.10 * k1 + 0.02 * n
            l["dry_scale_checks"] = checks
            failures += ["N = %d: %s" % (n, k) for k, ok in checks.items() if not ok]
        print(json.dumps(l))
    sys.stdout.flush()
    if args.dry_scale:
        sys.stderr.write("bench.py: dry scale: %d line(s), %s\n" % (len(lines), "all checks passed" if not failures else "FAILED: " + "; ".join(failures)))
        return 1 if failures else 0
    return 1 if failures else 0


def is_dist_error(e: BaseException) -> bool:
    """A failure of the collective layer (RCCL / gloo / the store), as opposed to a failure of the job itself."""
    import torch.distributed as dist
    kinds = tuple(k for k in (getattr(dist, "DistError", None), getattr(dist, "DistBackendError", None), getattr(dist, "DistNetworkError", None),
                              getattr(dist, "DistStoreError", None)) if k is not None) + (TimeoutError, ConnectionError)
    if isinstance(e, kinds):
        return True
    text = str(e)
    return isinstance(e, RuntimeError) and any(w in text for w in ("NCCL", "RCCL", "ProcessGroup", "c10d", "Gloo", "gloo", "collective", "Timed out", "timed out"))


def teardown_process_group(dist) -> None:
    """Leave the process group without waiting for collectives that will never finish (abort where torch has it, destroy otherwise)."""
    try:
        abort = getattr(dist.distributed_c10d, "_abort_process_group", None)
        if abort is not None:
            abort()
        else:
            dist.destroy_process_group()
    except Exception as e2:   # noqa: BLE001 -- already on the way out
        sys.stderr.write("bench.py: leaving the process group: %s\n" % e2)


def main():
    args = parse_args()
    if args.scale_sweep or args.dry_scale:
        raise SystemExit(scale_sweep(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    R = Runner(args)
    if R.dist_failed:   #  the first time RCCL runs must not cost the curve: rank 0 drives the N devices itself, the others leave
        if R.rank != 0:
            raise SystemExit(0)
        raise SystemExit(run_in_process(args, "torch.distributed could not be used by the %d ranks the caller launched (%s)" % (R.world, R.dist_failed)))
    primary_id = args.config or 2          # ONE workload over the whole 1 -> 8 curve (module docstring)
    try:
        prim = R.run(primary_id, primary=True, e2e_leg=(primary_id == 1))
    except Exception as e:   # noqa: BLE001
        # A collective that fails AFTER the process group came up (a fabric error in the tables' broadcast, a rank that died: the others time out
        # after 300 s) must not cost the line either: same way out as a failed init -- but ONLY for a failure of torch.distributed.  Anything
        # else (out of memory, an API error, a bug in this file) is a failure of the job: every rank exits non-zero and no line is printed.
        if R.world == 1 or os.environ.get("VPT_BENCH_NO_FALLBACK") or not is_dist_error(e):
            raise
        import traceback
        sys.stderr.write("bench.py: rank %d: the %d-rank job failed: %s\n%s" % (R.rank, R.world, e, traceback.format_exc()))
        teardown_process_group(R.dist)       # ranks still inside a collective let go of their devices before rank 0 times kernels on them
        if R.rank != 0:
            raise SystemExit(0)
        raise SystemExit(run_in_process(args, "the %d-rank torch.distributed job failed on rank 0: %s: %s" % (R.world, type(e).__name__, str(e).replace("\n", " ")[:200])))
    extra = []
    if not args.config and not args.quick and R.world == 1:
        for cid in (1, 3, 4, 5, 6, 7, 8):
            extra.append(R.run(cid, primary=False, e2e_leg=(cid == 1)))
    if R.rank == 0:
        line = {
            "metric": "boundary scores/sec", "value": prim["value"], "unit": "boundaries/s", "n_gpus": R.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": prim["ms_per_step"], "higher_is_better": True,
            "scaling": "strong" if primary_id == 2 else "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {k: prim[k] for k in ("workload", "tokenizer_model", "sentences_per_gpu", "boundaries_per_gpu", "text_bytes_per_gpu", "char_ngrams",
                                            "dict_words", "tag_models", "table_bytes", "hot_table_bytes", "packed_tables", "tiles", "tile_plan", "create_s",
                                            "tables_broadcast_s", "synth_s")},
            "parity": prim["parity"], "roofline": prim["roofline"], "cpu_baseline": prim.get("cpu_baseline"),
        }
        line["config"]["sharding"] = "contiguous sentence ranges balanced by chars over %d rank(s), no data-path collective" % R.world
        line["config"]["hip_devices_visible"] = R.torch.cuda.device_count()
        line["config"]["world_size"] = R.world
        line["config"]["collective_backend"] = ("RCCL (torch.distributed nccl)" if R.backend == "nccl" else R.backend) if R.world > 1 else None
        line["config"]["tables_broadcast"] = getattr(R, "tables_broadcast", None) if R.world > 1 else None
        line["config"]["launch"] = os.environ.get("VPT_BENCH_LAUNCH") or ("one process, one GPU" if R.world == 1 else "torch.distributed.run started by the caller")
        if os.environ.get("VPT_BENCH_ONE_DEVICE") and R.world > 1:
            line["config"]["launch"] += "; VPT_BENCH_ONE_DEVICE: every rank on device 0"
        for k in ("e2e", "tags", "emit", "phase_share", "per_rank"):
            if k in prim:
                line[k] = prim[k]
        # ONE row per workload inside the line (the driver keeps a 15 KB tail of stdout: round 4's full detail pushed configs[1] and
        # configs[3] out of its record); everything else goes to --detail-out
        if extra:
            line["workloads"] = [compact_row(w) for w in [prim] + extra]
            line["workloads_columns"] = "G boundaries/s; kernel_ms = HIP-event median of the scoring kernel; frac = algorithmic bytes / kernel_ms / 8 TB/s; " \
                                        "traffic_ratio = PMC bytes / algorithmic bytes (null: no PMC pass on these sources); tags_ms / emit_ms = fill_tags, " \
                                        "the writer's launch, wall clock per call"
        detail = dict(line, workloads=[prim] + extra) if extra else line
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail_out)), exist_ok=True)
            with open(args.detail_out, "w") as fh:
                json.dump(detail, fh, indent=1)
            line["config"]["detail_file"] = os.path.relpath(args.detail_out, ROOT)
        except OSError:
            pass
        text = json.dumps(shorten(line))
        if len(text) >= 12000:   # never past the driver's tail: drop the prose first
            for k in ("e2e", "emit", "tags", "phase_share"):
                if len(text) >= 12000 and k in line:
                    line.pop(k)
                    text = json.dumps(shorten(line))
        print(text)
        sys.stdout.flush()
    if R.world > 1:
        R.dist.barrier()
        R.dist.destroy_process_group()


if __name__ == "__main__":
    main()
