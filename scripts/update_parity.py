"""3-step NAdam update of golden G4 under the three conv forms: relative L2 of the update vs the reference golden."""
import os, sys
os.environ.setdefault("TONAL_AB", "1")      # timing / A/B script: the per-switch variables are honoured (_kernels.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import golden_inputs as gi
from tests.test_gpu_parity import _trainer, GOLD
g = np.load(os.path.join(GOLD, "g4_cnn_train.npz"))
xs, _t, _s, labs, tg = gi.train_batches(3, 8, 16, 200)
dev = torch.device("cuda:0")
for mode in ("0", "1", "4"):
    os.environ["TONAL_WINO"] = mode
    from decode_tonal_langauge_amd.models.synthesis_models import SynthesisModelCNN
    torch.manual_seed(0)
    model = SynthesisModelCNN(80, 16, 200, dropout=0.0)
    init = {k: v.detach().clone().numpy() for k, v in model.named_parameters()}
    tr = _trainer(model, dev, 200)
    model.train()
    for s in range(3):
        tr._fused_step(xs[s].to(dev), labs[s].to(dev), tg[s].to(dev))
    res = {}
    for k, p in model.named_parameters():
        fin = p.detach().cpu().numpy()
        if "final." + k in g:
            res[k] = gi.update_rel_l2(fin, g["final." + k], init[k])
        else:
            res[k] = gi.update_rel_l2(fin.reshape(-1)[::97], g["final." + k + "@s97"], init[k].reshape(-1)[::97])
    worst = sorted(res.items(), key=lambda kv: -kv[1])[:6]
    print("TONAL_WINO=" + mode, {k: f"{v:.2e}" for k, v in worst}, flush=True)
