"""Model-shape sweep on the GPU box: alphabet {4 K, 8 K, 16 K} x patterns {1.7 M, 4 M} x share of dictionary
words that repeat an n-gram {0, 5 %} (rows outside the packed fields) -> table bytes, vpt_predictor_create seconds, kernel time and
roofline fraction of the configs[1] batch (100 K x 64 chars, HBM resident), parity against the oracle.  One JSON line per model:
    python tools/model_sweep.py > gpurun_out/model_sweep.jsonl"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from oracle import cbind
    from vaporetto_amd import api, synth
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    S = 100_000
    for vocab in (4000, 8000, 16000):
        for scale in (1.0, 2.35):
            for dup in (0.0, 0.05):
                raw = synth.synth_model(1, synth.SEED_BASE + 2, scale, vocab=vocab, dup_share=dup)
                t = time.perf_counter()
                pred = api.Predictor(api.Model.read_slice(raw)[0], False)
                create_s = time.perf_counter() - t
                info = pred.info()
                utf8, boff = synth.synth_sentences(raw, S, 64, 64, seed=synth.SEED_BASE + 2)
                o_scores, o_labels, ooff, a_char = cbind.OraclePredictor(raw).predict_batch(utf8, boff, nthreads=os.cpu_count() or 1)
                nb, nbytes = int(ooff[-1]), int(boff[-1])
                d_text = torch.from_numpy(np.concatenate([utf8, np.zeros(64, np.uint8)])).to(dev)
                d_boff, d_ooff = torch.from_numpy(boff.astype(np.int64)).to(dev), torch.from_numpy(ooff.astype(np.int64)).to(dev)
                d_scores, d_labels = torch.empty(nb + 1, dtype=torch.int32, device=dev), torch.empty(nb + 1, dtype=torch.uint8, device=dev)
                batch = api.DeviceBatch(pred, timing=True)
                batch.set_max_sentence_chars(64)
                st = torch.cuda.current_stream().cuda_stream
                for k in range(25):
                    if k == 5:
                        batch.sync(); batch.kernel_ms()
                    batch.predict(d_text.data_ptr(), d_boff.data_ptr(), d_ooff.data_ptr(), S, nb, 192, d_scores.data_ptr(), d_labels.data_ptr(), st)
                batch.sync()
                kernel_ms = float(np.median(batch.kernel_times()))
                ok = bool(np.array_equal(d_scores[:nb].cpu().numpy(), o_scores) and np.array_equal(d_labels[:nb].cpu().numpy(), o_labels))
                a = nbytes + 5 * nb + 16 * S + a_char + 4 * nb
                print(json.dumps({"alphabet": vocab, "char_ngrams": info["n_char_ngrams"], "dict_words": info["n_dict_words"], "repeat_share": dup,
                                  "table_bytes": info["device_table_bytes"], "hot_table_bytes": info["hot_table_bytes"], "packed": bool(info["packed"]),
                                  "create_s": round(create_s, 2), "kernel_ms": round(kernel_ms, 4), "G_boundaries_per_s": round(nb / kernel_ms / 1e6, 2),
                                  "bytes_per_boundary": round(a / nb, 1), "frac_of_hbm_roofline": round(a / (kernel_ms * 1e-3) / 8e12, 3), "parity": ok}), flush=True)
                del batch, pred, d_text, d_scores, d_labels


if __name__ == "__main__":
    main()
