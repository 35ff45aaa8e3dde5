"""bench.py's launch contract: `python bench.py --gpus N` must never print a line for fewer GPUs than it
was asked for.  Runs wherever fewer than two HIP devices are visible (this container: none; a 1-GPU box: one)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _visible() -> int:
    import torch
    return torch.cuda.device_count()


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VPT_BENCH_ONE_DEVICE", "VPT_BENCH_BACKEND"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_more_gpus_than_visible_is_a_loud_error():
    if _visible() >= 2:
        pytest.skip("needs a machine with fewer than 2 HIP devices")
    r = _run(["--gpus", "2", "--quick"])
    assert r.returncode == 2 and b"HIP device(s) are visible" in r.stderr
    assert b'"metric"' not in r.stdout                       # no result line at all, in particular none with n_gpus = 1


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--quick"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and b"WORLD_SIZE=1" in r.stderr and b'"metric"' not in r.stdout
    r = _run(["--gpus", "1", "--quick"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and b"WORLD_SIZE=2" in r.stderr and b'"metric"' not in r.stdout


@pytest.mark.gpu
def test_two_ranks_on_one_device_through_the_self_launch():
    """The N-rank code path on a 1-GPU box (both ranks on device 0, gloo): bench.py launches its ranks itself, the line says
    n_gpus = 2 and names the same workload as the 1-rank line; the in-process fallback gives the same shape of line."""
    env = {"VPT_BENCH_ONE_DEVICE": "1", "VPT_BENCH_BACKEND": "gloo"}
    common = ["--config", "2", "--sentences", "200000", "--model-scale", "0.05", "--steps", "3", "--warmup", "1", "--no-emit", "--no-e2e"]
    one = _run(["--gpus", "1"] + common)
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = json.loads([l for l in one.stdout.decode().splitlines() if l.startswith("{")][-1])
    two = _run(["--gpus", "2"] + common, env)
    assert two.returncode == 0, two.stderr[-2000:]
    l2 = json.loads([l for l in two.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["config"]["world_size"] == 2
    assert l1["config"]["workload"] == l2["config"]["workload"] and l1["scaling"] == l2["scaling"] == "strong"
    assert l1["parity"] is True and l2["parity"] is True and l2["cpu_baseline"] is not None
    assert "launched its 2 ranks itself" in l2["config"]["launch"]
    fb = _run(["--gpus", "2", "--in-process"] + common, env)
    assert fb.returncode == 0, fb.stderr[-2000:]
    l3 = json.loads([l for l in fb.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert l3["n_gpus"] == 2 and l3["parity"] is True and l3["config"]["workload"] == l1["config"]["workload"]
    assert "in-process fallback" in l3["config"]["launch"]


@pytest.mark.gpu
def test_dry_scale_sweep_up_to_four_ranks():
    """`bench.py --dry-scale`: the scaling job's pre-flight -- N = 1, 2, 4 as jobs of their own (on a 1-GPU box every rank on
    device 0 over gloo), one line per N with the same workload, parity, per-rank figures, and the ranks' solo kernels adding up to the N = 1 kernel."""
    r = _run(["--dry-scale", "--scale-sweep", "1,2,4", "--config", "2", "--sentences", "400000", "--model-scale", "0.05", "--steps", "3", "--warmup", "1",
              "--no-emit", "--no-e2e"])
    lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert r.returncode == 0, (r.stderr[-3000:], [l.get("dry_scale_checks") for l in lines])
    assert [l["n_gpus"] for l in lines] == [1, 2, 4]
    assert len({l["config"]["workload"] for l in lines}) == 1
    for l in lines:
        assert l["parity"] is True and all(l["dry_scale_checks"].values()), l["dry_scale_checks"]
        assert l["scaling_efficiency"] is not None
        if l["n_gpus"] > 1:
            assert len(l["per_rank"]) == l["n_gpus"] and all(pr["tables_broadcast_s"] is not None for pr in l["per_rank"])
            assert sum(pr["sentences"] for pr in l["per_rank"]) == 400000


def test_the_result_line_stays_below_the_drivers_tail():
    """the driver keeps a 15 KB tail of stdout and round 4's line (25 KB: every workload's full detail) lost configs[1]
    and configs[3] to it.  The line now carries ONE compact row per workload; checked here on round 4's own detail (eight workloads)."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r04_zz_bench_all_workloads.json")) as fh:
        old = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])
    prim = dict(old["config"], value=old["value"], ms_per_step=old["ms_per_step"], roofline=old["roofline"], parity=old["parity"])
    rows = [bench.compact_row(w) for w in [prim] + old["workloads"] + [old["workloads"][0]]]
    line = dict(old, workloads=rows, workloads_columns="x" * 400)
    text = json.dumps(bench.shorten(line))
    assert len(text) < 12000, len(text)
    back = json.loads(text)
    names = [r["name"] for r in back["workloads"]]
    assert names[:5] == ["configs[2]", "configs[1]", "configs[3]", "configs[4]", "documents"]
    for r in back["workloads"]:
        assert r["parity"] is True and r["kernel_ms"] > 0 and 0 < r["frac"] < 1 and r["kernel"].startswith("score_tiles")
    assert back["workloads"][1]["e2e_ms"] > 0 and back["workloads"][3]["tags_ms"] > 0
    assert "traffic_ratio" in back["workloads"][0]
    assert back["roofline"]["frac"] == pytest.approx(old["roofline"]["frac"], rel=1e-5) and back["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_table_broadcast_paths_and_a_failed_init_on_one_device():
    """the 2-rank job (both ranks on device 0 over gloo) with the tables travelling each of the three ways (dist.py), and with
    an init that fails (VPT_BENCH_FAIL_INIT: as if RCCL could not come up): rank 0 then drives the devices itself and the line says so -- exit code 0."""
    env = {"VPT_BENCH_ONE_DEVICE": "1", "VPT_BENCH_BACKEND": "gloo"}
    common = ["--config", "2", "--sentences", "200000", "--model-scale", "0.05", "--steps", "3", "--warmup", "1", "--no-emit", "--no-e2e"]
    for path in ("view", "staged", "compile"):
        r = _run(["--gpus", "2"] + common, dict(env, VPT_TABLES_BROADCAST=path))
        assert r.returncode == 0, r.stderr[-2000:]
        l = json.loads([x for x in r.stdout.decode().splitlines() if x.startswith("{")][-1])
        assert l["n_gpus"] == 2 and l["parity"] is True and l["config"]["tables_broadcast"] == path, l["config"]
        assert [pr["tables_broadcast"] for pr in l["per_rank"]] == [path, path]
    r = _run(["--gpus", "2"] + common, dict(env, VPT_BENCH_FAIL_INIT="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    l = json.loads([x for x in r.stdout.decode().splitlines() if x.startswith("{")][-1])
    assert l["n_gpus"] == 2 and l["parity"] is True and "in-process fallback" in l["config"]["launch"] and "could not be used" in l["config"]["launch"]


@pytest.mark.gpu
def test_dry_scale_of_the_tagged_ragged_workload_at_four_ranks():
    """BASELINE configs[4] ("8 x MI355X": tags on, 8..512-char sentences) as a sharded job: `--scale-config 4` through the dry run at N = 1 and 4 --
    parity of scores, labels and tags on every rank, and the ranks' shares balanced by CHARS (within one longest sentence of each other)."""
    r = _run(["--dry-scale", "--scale-sweep", "1,4", "--scale-config", "4", "--sentences", "100000", "--model-scale", "0.05", "--steps", "3", "--warmup", "1",
              "--no-emit", "--no-e2e"])
    lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert r.returncode == 0, (r.stderr[-3000:], [l.get("dry_scale_checks") for l in lines])
    assert [l["n_gpus"] for l in lines] == [1, 4] and all(l["parity"] is True for l in lines)
    assert all("configs[4]" in l["config"]["workload"] for l in lines) and lines[1]["tags"]["parity"] is True
    chars = [pr["chars"] for pr in lines[1]["per_rank"]]
    assert len(chars) == 4 and max(chars) - min(chars) <= 2 * 512, chars


def test_the_gather_block_of_the_line():
    """roofline.gather: lane loads from the node reads of a diagnostics launch, tools/tcp_bench's rates from profiles/, and -- with a
    PMC entry for the sources -- every level's requests per second over the rate the microbenchmark measured for it; the largest share binds."""
    sys.path.insert(0, ROOT)
    import bench
    rates = bench.gather_rates()
    assert rates and set(rates) == {"l1", "l2", "hbm"} and rates["l1"] > rates["l2"] > rates["hbm"] > 0      # profiles/r06_j_tcp_bench.jsonl
    reads = {"unigram_nodes": 1000, "bigram_nodes": 900, "trigram_nodes": 400, "deep_entries": 200, "deep_rows": 50, "global_type_rows": 0}
    g = bench.gather_block(None, reads, chars=1100, wl=3, kernel_ms=0.001)
    assert g["lane_loads_per_launch"] == 1000 + 2 * 900 + 400 + 200 + 50 + 1100 and "frac" not in g and g["microbench_Glanes_s"] == rates
    assert abs(g["achieved_Glanes_s"] - g["lane_loads_per_launch"] / 1e-6 / 1e9) < 1e-9
    wide = bench.gather_block(None, reads, chars=0, wl=4, kernel_ms=0.001)
    assert wide["lane_loads_per_launch"] == 2 * 1000 + 4 * 900 + 2 * 400 + 200 + 50                               # 32- / 64- / 32-byte nodes at row window 4
    entry = {"tcp_lookups": 3.0e9, "tcp_tcc_read_req": 1.1e9, "tcc_miss": 3.3e8, "tcp_busy_cycles": 9.0e8, "tcp_cycles": 1.0e9}
    g = bench.gather_block(entry, reads, chars=0, wl=3, kernel_ms=9.0)
    t = 9.0e-3
    want = {"l1": 3.0e9 / t / 1e9 / rates["l1"], "l2": 1.1e9 / t / 1e9 / rates["l2"], "hbm": 3.3e8 / t / 1e9 / rates["hbm"]}
    assert g["bound"] == max(want, key=want.get) and abs(g["frac"] - max(want.values())) < 1e-3 and abs(g["pmc"]["tcp_busy_share"] - 0.9) < 1e-9
    assert bench.gather_block(entry, reads, 0, 3, None) is None
