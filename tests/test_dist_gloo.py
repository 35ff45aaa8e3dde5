"""world_size-2 `gloo` test of the multi-GPU host logic (vaporetto_amd/dist.py): model broadcast from rank 0, byte-
balanced contiguous shards, whole-job reductions.  Scoring on each rank is done by the CPU oracle here (the GPU
path is covered by the -m gpu tests); the sharded results must concatenate to the unsharded ones."""
import os
import socket
import subprocess
import sys

import numpy as np

from oracle import cbind
from tests import randmodel
from vaporetto_amd import api, dist as vdist
from vaporetto_amd.modelfmt import encode_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, hashlib
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from vaporetto_amd import dist as vdist, api
from oracle import cbind
from tests import randmodel
from vaporetto_amd.modelfmt import encode_model

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
m = randmodel.rand_model(31, alphabet="kana", wc=3, wt=3, n_char=150, n_dict=150, max_word=9)
raw0 = encode_model(m) if rank == 0 else None
raw = vdist.broadcast_model_bytes(raw0, src=0)
texts = randmodel.rand_sentences(9, m, 900, alphabet="kana", max_len=150)   # every rank builds the same batch
utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
s_utf8, s_boff, first = vdist.take_shard(utf8, boff, rank, world)
scores, labels, ooff, _ = cbind.OraclePredictor(raw).predict_batch(s_utf8, s_boff)
el, tot = vdist.reduce_throughput(1.0 + rank, float(len(scores)))
np.savez(os.path.join({out!r}, "rank%d.npz" % rank), scores=scores, labels=labels, first=first, n=len(s_boff) - 1,
         sha=np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8), el=el, tot=tot)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_balance_and_cover():
    rng = np.random.RandomState(0)
    lens = rng.randint(1, 600, size=5000)
    boff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for world in (1, 2, 3, 4, 8):
        b = vdist.shard_bounds(boff, world)
        assert b[0] == 0 and b[-1] == 5000 and np.all(np.diff(b) >= 0)
        sizes = [int(boff[b[r + 1]] - boff[b[r]]) for r in range(world)]
        assert max(sizes) - min(sizes) <= 2 * 600
    # degenerate: fewer sentences than ranks
    b = vdist.shard_bounds(np.array([0, 5, 9], dtype=np.uint64), 4)
    assert b[0] == 0 and b[-1] == 2 and np.all(np.diff(b) >= 0)


def test_two_rank_gloo_broadcast_shard_and_reduce(tmp_path):
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)

    m = randmodel.rand_model(31, alphabet="kana", wc=3, wt=3, n_char=150, n_dict=150, max_word=9)
    raw = encode_model(m)
    texts = randmodel.rand_sentences(9, m, 900, alphabet="kana", max_len=150)
    utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
    full_scores, full_labels, _, _ = cbind.OraclePredictor(raw).predict_batch(utf8, boff)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    import hashlib
    want = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
    assert np.array_equal(r[0]["sha"], want) and np.array_equal(r[1]["sha"], want)      # broadcast delivered the model
    assert int(r[0]["first"]) == 0 and int(r[1]["first"]) == int(r[0]["n"]) and int(r[0]["n"]) + int(r[1]["n"]) == 900
    assert np.array_equal(np.concatenate([r[0]["scores"], r[1]["scores"]]), full_scores)   # shards tile the batch
    assert np.array_equal(np.concatenate([r[0]["labels"], r[1]["labels"]]), full_labels)
    for k in range(2):
        assert float(r[k]["el"]) == 2.0 and float(r[k]["tot"]) == float(len(full_scores))  # MAX / SUM over ranks
