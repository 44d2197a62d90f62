"""Round-4 golden from the REAL reference (build container only; see make_golden.py for the rules: the reference is
imported from /root/reference, nothing of it is copied, only input checksums and outputs are stored).

  G11b  THREE NAdam steps of the reference's ``SynthesisModelCNN(80, 128, 400, dropout=0.0)`` - the timed shape
        (H = 18 432, 1 376 768 720 parameters, the 73 728-row LSTM, the low-rank W_hh update, the split-K Linear) - at
        B = 2 on three distinct seeded batches, through the body of the reference's batch loop
        (models/synthesis_trainer.py:198-236: zero_grad, forward, integer-truncated targets, L1Loss, backward,
        NAdam(lr 5e-4, betas (0.9, 0.999), eps 1e-8, weight_decay 0.004), compute_mcd).  Stored per step: L1 loss, MCD,
        the mel MSE mean((out - target)^2), the output; after the third step every parameter as a prime-strided sample +
        (sum, abs-sum), and the same sample of the initial parameters (so a test can form the three-step UPDATE vector).
        G11 pins one step at this shape, G14 thirty steps at 16 x 200; this one carries the timed geometry past step 1.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r4.py
Needs ~10 min and ~35 GB of host memory.
"""
from __future__ import annotations

import argparse
import gc
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from oracle.make_golden import _import_reference, maxrel  # noqa: E402
from oracle.make_golden_r2 import pack_sampled, _prime_at_least  # noqa: E402
import golden_inputs as gi  # noqa: E402

STEPS, B, C, T, D = 3, 2, 128, 400, 80
SEED, DATA_SEED = 0, 8642
MAX_SAMPLES = 20000


def _strided(a: np.ndarray) -> np.ndarray:
    """The same flat sample pack_sampled keeps (full tensor up to MAX_SAMPLES elements)."""
    if a.size <= MAX_SAMPLES:
        return a.copy()
    return a.reshape(-1)[::_prime_at_least(-(-a.size // MAX_SAMPLES))].copy()


def golden_c3_traj(out_dir, report):
    rsm, rst, _rsc, rdu, _rdl, _rff = _import_reference()
    from oracle import synthesis_oracle as so
    xs, tones, syls, labs, tg = gi.train_batches(STEPS, B, C, T, seed=DATA_SEED)
    for t_, s_, l_ in zip(tones, syls, labs):
        assert np.array_equal(rdu.prepare_tone_dynamics(gi.TONE_MAP, t_.numpy(), s_.numpy()), l_.numpy())
    torch.manual_seed(SEED)
    net = rsm.SynthesisModelCNN(D, C, T, dropout=0.0)
    assert net.get_nparams() == 1376768720
    keep = {"dims": np.array([D, C, T, B, STEPS]), "seed": SEED, "data_seed": DATA_SEED,
            "in_checksum": gi.checksum(*xs, *labs, *tg)}
    for k, v in net.named_parameters():
        keep["init." + k] = _strided(v.detach().numpy())
    opt = torch.optim.NAdam(net.parameters(), lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.004)
    crit = torch.nn.L1Loss()
    net.train()
    losses, mcds, mses, outs = [], [], [], []
    for s in range(STEPS):
        opt.zero_grad()
        o = net(xs[s], labs[s])
        t = tg[s].long()
        loss = crit(o, t)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        mcds.append(float(rst.compute_mcd(t, o)))
        mses.append(float(((o.detach().double() - tg[s].double()) ** 2).mean()))
        outs.append(o.detach().clone().numpy())
        print(f"reference step {s}: loss {losses[-1]:.6f} mcd {mcds[-1]:.4f} mse {mses[-1]:.6f}", flush=True)
    keep.update(losses=np.array(losses), mcds=np.array(mcds), mses=np.array(mses), outs=np.stack(outs))
    keep.update(pack_sampled({"final." + k: v for k, v in net.named_parameters()}, MAX_SAMPLES))
    del opt, net, o, loss
    gc.collect()

    # ---- the oracle (CPU restatement) over the same three steps ----
    torch.manual_seed(SEED)
    p = so.init_cnn_params(D, C, T)
    st = so.NAdamState(p)
    ol, om = [], []
    for s in range(STEPS):
        l_, m_, _g, o_ = so.train_step("cnn", p, None, st, xs[s], labs[s], tg[s], return_grads=True)
        ol.append(l_)
        om.append(float(((o_.double() - tg[s].double()) ** 2).mean()))
        del _g
        gc.collect()
    report["g11b_c3_traj_loss_oracle_vs_reference"] = maxrel(ol, losses)
    report["g11b_c3_traj_mse_oracle_vs_reference"] = max(abs(a - b) / b for a, b in zip(om, mses))
    worst = 0.0
    for k, v in p.items():
        fin = _strided(v.detach().numpy()).reshape(-1)
        key = "final." + k if "final." + k in keep else next(
            q for q in keep if q.startswith("final." + k + "@s") and not q.endswith("@sum"))
        worst = max(worst, gi.update_rel_l2(fin, keep[key].reshape(-1), keep["init." + k].reshape(-1)))
    report["g11b_c3_traj_update_l2_oracle_vs_reference"] = worst
    np.savez_compressed(os.path.join(out_dir, "g11b_c3_trajectory.npz"), **keep)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    report = {}
    golden_c3_traj(args.out, report)
    with open(os.path.join(args.out, "PINNING.txt"), "a") as f:
        f.write("\n# round 4 (oracle/make_golden_r4.py): max relative deviation oracle vs imported reference\n")
        for k, v in sorted(report.items()):
            f.write(f"{k} {v:.3e}\n")
            print(k, f"{v:.3e}")


if __name__ == "__main__":
    main()
