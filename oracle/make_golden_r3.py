"""Round-3 goldens from the REAL reference (build container only; see make_golden.py for the rules: the reference is
imported from /root/reference, nothing of it is copied, only inputs' checksums and outputs are stored).

  G14  a training TRAJECTORY: 30 NAdam steps of the reference's ``SynthesisModelCNN(80, 16, 200, dropout=0.0)`` on 30
       distinct seeded batches of 8 windows, through the body of the reference's batch loop
       (models/synthesis_trainer.py:198-236: zero_grad, forward, integer-truncated targets, L1Loss, backward,
       NAdam(lr 5e-4, betas (0.9, 0.999), eps 1e-8, weight_decay 0.004), compute_mcd).  Stored per step: L1 loss, MCD,
       the mel MSE mean((out - target)^2) and the output itself.  This is the form of parity BASELINE.json's north_star
       states ("within 1e-3 rel on the mel-spectrogram MSE for identical seeds").
  G15  ``hilbert_filter`` (preprocess/signal/frequency_filter.py:80-184) away from the 400 Hz / high-gamma case: a low
       band (1-4 Hz) and the high-gamma band at a raw-recording rate of 3 kHz, envelope and real part - the low band's
       Gaussian kernels are thousands of samples long there (the DFT-domain path of the product).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_r3.py [--only traj,hilbert]
"""
from __future__ import annotations

import argparse
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
import golden_inputs as gi  # noqa: E402

TRAJ_STEPS, TRAJ_B, TRAJ_C, TRAJ_T = 30, 8, 16, 200


def golden_traj(out_dir, report):
    rsm, rst, _rsc, rdu, _rdl, _rff = _import_reference()
    from oracle import synthesis_oracle as so
    xs, tones, syls, labs, tg = gi.train_batches(TRAJ_STEPS, TRAJ_B, TRAJ_C, TRAJ_T, seed=4321)
    for t_, s_, l_ in zip(tones, syls, labs):   # the reference's own label builder agrees with the test helper
        assert np.array_equal(rdu.prepare_tone_dynamics(gi.TONE_MAP, t_.numpy(), s_.numpy()), l_.numpy())
    torch.manual_seed(0)
    net = rsm.SynthesisModelCNN(80, TRAJ_C, TRAJ_T, dropout=0.0)
    opt = torch.optim.NAdam(net.parameters(), lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.004)
    crit = torch.nn.L1Loss()
    net.train()
    losses, mcds, mses, outs = [], [], [], []
    for s in range(TRAJ_STEPS):
        opt.zero_grad()
        o = net(xs[s], labs[s])
        t = tg[s].long()
        loss = crit(o, t)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        mcds.append(rst.compute_mcd(t, o))
        mses.append(float(((o.detach().double() - tg[s].double()) ** 2).mean()))
        outs.append(o.detach().clone().numpy())
    # the oracle (CPU restatement) over the same trajectory
    torch.manual_seed(0)
    p = so.init_cnn_params(80, TRAJ_C, TRAJ_T)
    st = so.NAdamState(p)
    ol, om = [], []
    for s in range(TRAJ_STEPS):
        l_, m_, _g, o_ = so.train_step("cnn", p, None, st, xs[s], labs[s], tg[s], return_grads=True)
        ol.append(l_)
        om.append(float(((o_.double() - tg[s].double()) ** 2).mean()))
    report["g14_traj_loss_oracle_vs_reference"] = maxrel(ol, losses)
    report["g14_traj_mse_oracle_vs_reference"] = max(abs(a - b) / b for a, b in zip(om, mses))
    np.savez_compressed(os.path.join(out_dir, "g14_cnn_trajectory.npz"),
                        losses=np.array(losses), mcds=np.array(mcds), mses=np.array(mses), outs=np.stack(outs),
                        dims=np.array([80, TRAJ_C, TRAJ_T, TRAJ_B, TRAJ_STEPS]), seed=0, data_seed=4321,
                        in_checksum=gi.checksum(*xs, *labs, *tg))


def golden_hilbert(out_dir, report):
    _rsm, _rst, _rsc, _rdu, _rdl, rff = _import_reference()
    from oracle import signal_oracle as sg
    fs, T = 3000, 9000
    x = np.random.default_rng(15).standard_normal((2, T))
    keep = {"fs": fs, "x_checksum": float(np.abs(x).sum())}
    for name, fr, env in (("low_env", [1.0, 4.0], True), ("low_real", [1.0, 4.0], False), ("hg_env", [70.0, 150.0], True)):
        ref = rff.hilbert_filter(x, fs, fr, envelope=env)
        mine = sg.hilbert_filter(x, fs, fr, envelope=env)
        report[f"g15_hilbert_{name}_oracle_vs_reference"] = maxrel(mine, ref)
        keep[name] = ref
    x32 = x.astype(np.float32)
    keep["low_env_f32"] = rff.hilbert_filter(x32, fs, [1.0, 4.0], envelope=True)
    np.savez_compressed(os.path.join(out_dir, "g15_hilbert_low_band.npz"), **keep)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="traj,hilbert")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    report = {}
    only = args.only.split(",")
    if "traj" in only:
        golden_traj(args.out, report)
    if "hilbert" in only:
        golden_hilbert(args.out, report)
    with open(os.path.join(args.out, "PINNING.txt"), "a") as f:
        f.write("\n# round 3 (oracle/make_golden_r3.py): max relative deviation oracle vs imported reference\n")
        for k, v in sorted(report.items()):
            f.write(f"{k} {v:.3e}\n")
            print(k, f"{v:.3e}")


if __name__ == "__main__":
    main()
