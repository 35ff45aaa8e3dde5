"""How the headline number depends on the text: the synthetic sentences are drawn `hit_share` from the model's own
patterns (0.7) and the algorithmic bytes -- hence `frac` -- scale with how often patterns hit.  For every share: G
boundaries/s, kernel ms, algorithmic bytes per boundary, frac, parity, and -- from a second, counted run of the diagnostics build of the
kernel (VPT_PROFILE_PHASES) -- the node reads the lanes issued per trie level: "useful node bytes" = reads x node size, the figure to set
beside the 128-byte-line traffic the L2 fetches for them (tools/profile.sh with VPT_HIT_SHARE gives that per share).

    python tools/hit_share_sweep.py [--shares 0.3,0.5,0.7] [--model-kind 1] [--sentences 100000]      -> one JSON line per share
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shares", default="0.1,0.3,0.5,0.7,0.9")
    ap.add_argument("--model-kind", type=int, default=1)
    ap.add_argument("--model-scale", type=float, default=1.0)
    ap.add_argument("--sentences", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import torch
    import bench
    from oracle import cbind
    from vaporetto_amd import api, synth
    dev = torch.device("cuda", 0)
    raw, name = bench.load_model_bytes(args.model_kind, args.model_scale)
    orc = cbind.OraclePredictor(raw)
    os.environ.pop("VPT_PROFILE_PHASES", None)
    pred = api.Predictor(api.Model.read_slice(raw)[0], False, device=0)
    wl = max(3, pred.info()["char_window"], pred.info()["type_window"])
    node_bytes = {"unigram_nodes": 4 * {3: 4, 4: 8, 5: 8, 6: 8, 7: 16, 8: 16}[wl], "bigram_nodes": 32 if wl == 3 else 64,
                  "trigram_nodes": 4 * {3: 4, 4: 8, 5: 8, 6: 8, 7: 8, 8: 16}[wl], "deep_entries": 16, "deep_rows": 16, "global_type_rows": 4 * ((2 * wl + 3) & ~3)}
    for share in [float(x) for x in args.shares.split(",")]:
        utf8, boff = synth.synth_sentences(raw, args.sentences, 64, 64, seed=synth.SEED_BASE + 2, hit_share=share)
        o_scores, o_labels, ooff, a_char = orc.predict_batch(utf8, boff, nthreads=os.cpu_count() or 1)
        S, nb, nbytes = args.sentences, int(ooff[-1]), int(boff[-1])
        d = [torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev), torch.from_numpy(boff.astype(np.int64)).to(dev),
             torch.from_numpy(ooff.astype(np.int64)).to(dev), torch.empty(nb + 1, dtype=torch.int32, device=dev), torch.empty(nb + 1, dtype=torch.uint8, device=dev)]
        stream = torch.cuda.current_stream().cuda_stream
        out = {"hit_share": share, "model": name, "sentences": S}
        for counted in (False, True):
            if counted:
                os.environ["VPT_PROFILE_PHASES"] = "1"
            batch = api.DeviceBatch(pred, timing=True)
            os.environ.pop("VPT_PROFILE_PHASES", None)
            batch.set_max_sentence_chars(64)
            n = 3 if counted else args.steps + 5
            for i in range(n):
                if i == 5 and not counted:
                    batch.sync()
                    batch.kernel_ms()
                batch.predict(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), S, nb, int(np.max(np.diff(boff.astype(np.int64)))), d[3].data_ptr(), d[4].data_ptr(), stream)
            batch.sync()
            if counted:
                reads = {k: v // n for k, v in batch.node_reads().items()}
                useful = sum(reads[k] * node_bytes[k] for k in reads)
                out["node_reads_per_launch"] = reads
                out["useful_node_bytes"] = useful
                out["useful_node_bytes_per_boundary"] = useful / nb
                out["lines_if_one_128B_line_per_read"] = 128 * sum(reads.values())
            else:
                kt = batch.kernel_times()
                ms = float(np.median(kt))
                a = nbytes + 5 * nb + 16 * S + 4 * nb + a_char
                ok = bool(np.array_equal(d[3][:nb].cpu().numpy(), o_scores) and np.array_equal(d[4][:nb].cpu().numpy(), o_labels))
                out.update({"kernel_ms": ms, "G_boundaries_per_s": nb / ms / 1e6, "bytes_per_boundary": a / nb, "a_char_per_boundary": a_char / nb,
                            "frac": a / (ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, "parity": ok, "label_1_share": float((o_labels == 1).mean())})
            del batch
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
