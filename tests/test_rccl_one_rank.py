"""RCCL once, with ONE rank, on a one-GPU box: everything `vaporetto_amd/dist.py` does between ranks pushed through
the `nccl` backend in a process group of one -- `force` takes the world-1 short cut away, `loopback` makes rank 0 its own receiver --
so that the first multi-GPU run is not the first time torch's RCCL meets memory this library allocated:
the zero-copy VIEW of the predictor's arena handed to `dist.broadcast`, `vpt_predictor_adopt_device` from a torch tensor, the STAGED
form through torch-owned memory, the COMPILE path with the header's predict_tags, `all_reduce` (agreement, reduce_throughput),
`broadcast_object_list` / `all_gather_object` as bench.py uses them.  Every predictor that comes back is checked against the oracle.
The worker runs in a process of its own (a process group is process-wide state) and leaves its log in gpurun_out/ when that exists."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, datetime
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from oracle import cbind
from tests import randmodel
from vaporetto_amd import api, dist as vdist
from vaporetto_amd.modelfmt import encode_model

log = {{"torch": torch.__version__, "hip": torch.version.hip}}
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))      # RCCL
log["backend"] = dist.get_backend()
t = torch.ones(1, dtype=torch.int32, device=dev)
dist.all_reduce(t)
assert int(t.item()) == 1

m = randmodel.rand_model(77, alphabet="kana", wc=3, wt=3, n_char=400, n_dict=400, max_word=9, n_tag_models=20)
raw0 = encode_model(m)
raw = vdist.broadcast_model_bytes(raw0, src=0, device=dev, force=True)           # int64 length + uint8 blob through RCCL
assert raw == raw0
texts = randmodel.rand_sentences(5, m, 3000, alphabet="kana", max_len=120)
utf8, boff = api.pack_texts([t.encode("utf-8") for t in texts])
orc = cbind.OraclePredictor(raw, True)
o_scores, o_labels, o_ooff, _ = orc.predict_batch(utf8, boff)
o_tags, _, _ = orc.fill_tags_batch(utf8, boff, o_ooff, o_labels, want_scores=False)

p0 = api.Predictor(api.Model.read_slice(raw)[0], True, device=0)
log["paths"] = {{}}
for path in vdist.TABLES_BROADCAST_PATHS:
    os.environ["VPT_TABLES_BROADCAST"] = path
    p = vdist.broadcast_predictor(p0, src=0, device=dev, model_bytes=raw, force=True, loopback=True)
    assert p is not p0 and p.tables_broadcast == path, (path, p.tables_broadcast)      # the RECEIVER's predictor, by the path asked for
    assert p.info()["predict_tags"] == 1
    scores, labels, ooff = p.predict_packed(utf8, boff)
    assert np.array_equal(scores, o_scores) and np.array_equal(labels, o_labels) and np.array_equal(ooff, o_ooff), path
    tags = p.fill_tags_packed(utf8, boff, ooff, labels)
    assert np.array_equal(tags, o_tags), path
    log["paths"][path] = {{"tables_broadcast": p.tables_broadcast, "table_bytes": int(p.info()["device_table_bytes"]), "parity": True}}
    del p
os.environ.pop("VPT_TABLES_BROADCAST")
p = vdist.broadcast_predictor(p0, src=0, device=dev, force=True)                  # src's own way out of the default path
assert p is p0 and p.tables_broadcast == "view"
el, tot = vdist.reduce_throughput(1.5, 42.0, device=dev, force=True)              # all_reduce MAX / SUM of float64 on the device
assert (el, tot) == (1.5, 42.0)
names = ["x"]
dist.broadcast_object_list(names, src=0)
got = [None]
dist.all_gather_object(got, {{"rank": 0, "kernel_ms": 0.1}})
assert names == ["x"] and got == [{{"rank": 0, "kernel_ms": 0.1}}]
dist.barrier()
torch.cuda.synchronize()
# which HIP runtimes and RCCL this process mapped: the library links /opt/rocm's libamdhip64, torch ships its own
maps = sorted({{l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("libamdhip64", "librccl", "libvaporetto_hip", "libhsa-runtime"))}})
log["mapped"] = maps
dist.destroy_process_group()
print("RCCL_ONE_RANK " + json.dumps(log))
'''


@pytest.mark.gpu
def test_every_collective_of_the_multi_gpu_path_through_rccl_with_one_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VPT_TABLES_BROADCAST", "VPT_DIST_FORCE_COLLECTIVES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode("utf-8", "replace")
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "rccl_one_rank.log"), "w") as fh:
            fh.write(out)
    assert r.returncode == 0, out[-4000:]
    line = [l for l in out.splitlines() if l.startswith("RCCL_ONE_RANK ")]
    assert len(line) == 1, out[-2000:]
    import json
    log = json.loads(line[0][len("RCCL_ONE_RANK "):])
    assert log["backend"] == "nccl" and sorted(log["paths"]) == ["compile", "staged", "view"]
    assert any("librccl" in p for p in log["mapped"]) and any("libvaporetto_hip" in p for p in log["mapped"])
