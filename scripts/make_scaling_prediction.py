#!/usr/bin/env python3
"""profiles/scaling_prediction.json: the strong-scaling curve the first real `bench.py --gpus N` run should show, from the
single-GPU `rank_batch_probe` of a default bench line (train-step time at the per-rank batches 128 / 64 / 32, whole LSTM on the
rank, no exchange) - machine-readable form of DESIGN.md section 7 (review item 9c).  NOTHING here is measured on N > 1 GPUs.

    python scripts/make_scaling_prediction.py profiles/r06_c3_bench_line.log

Model: per-rank step(N) = probe[256 / N] - whh_ms (N - 1) / N  (the row-sharded label LSTM divides the W_hh passes and its NAdam pass)
                        + exchange(N) + small_collectives (ten dependent ones of the sharded LSTM) ; exchange = ring all-reduce of
                        72 MB at ~100 GB/s per direction and link, of which only what the convolution backward cannot hide is exposed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_c3_bench_line.log")
line = json.loads([l for l in open(src) if l.startswith("{")][-1])
probe = {int(k): float(v) for k, v in line["rank_batch_probe"]["ms_per_step_at_batch"].items()}
probe[256] = float(line["ms_per_step"])
WHH_MS = 4 * 0.96 + 3 * 0.94 + 5.7            # seven 5.4 GB passes over W_hh + the NAdam pass (profiles/r06_c3_step_summary.md)
EXCH_BYTES = 72e6                  # every gradient except W_hh
LINK = 100e9                       # achieved per direction on one xGMI link (assumption; the spec is ~153 GB/s)
rows = []
for n in (1, 2, 4, 8):
    b = 256 // n
    comp = probe[b] - WHH_MS * (n - 1) / n
    ring = 0.0 if n == 1 else 2 * (n - 1) / n * EXCH_BYTES / LINK * 1e3
    exposed = 0.0 if n == 1 else 0.15 * ring + 0.2       # the 9 MB tail bucket + launch / wait overhead
    small = 0.0 if n == 1 else 0.4
    ms = comp + exposed + small
    rows.append({"n_gpus": n, "per_rank_batch": b, "single_gpu_step_ms_at_that_batch": round(probe[b], 2),
                 "whh_part_removed_by_sharding_ms": round(WHH_MS * (n - 1) / n, 2), "ring_allreduce_ms": round(ring, 2),
                 "exchange_exposed_ms": round(exposed, 2), "sharded_lstm_collectives_ms": small,
                 "predicted_ms_per_step": round(ms, 1), "predicted_mel_frames_per_s": round(256 / ms * 1e3),
                 "predicted_efficiency": round((256 / ms) / (n * 256 / probe[256]), 3)})
out = {"status": "PREDICTION - never run on more than one GPU", "from": os.path.relpath(src, ROOT),
       "model": __doc__.split("Model:")[1].strip(), "whh_ms_single_gpu": round(WHH_MS, 2),
       "fallback_if_the_small_collectives_cost_milliseconds": "--lstm-shard auto keeps the whole LSTM per rank: predicted step = "
       "single_gpu_step_ms_at_that_batch + exchange + one 9.4 MB all-reduce of the factor rows",
       "curve": rows}
with open(os.path.join(ROOT, "profiles", "scaling_prediction.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(rows, indent=1))
