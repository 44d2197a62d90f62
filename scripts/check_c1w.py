"""Stage-2 input gradient: the fused first-stage weight gradient (epilogue 4 of the V-form NT kernel) against the
un-fused pair (MASK epilogue writing G1, then tl_conv1_wgrad) on the same random buffers.

    python scripts/check_c1w.py [--batch 6] [--channels 8] [--timepoints 200] [--c1 512]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from decode_tonal_langauge_amd._cnn_engine import CnnEngine
from decode_tonal_langauge_amd._lib import check, ptr

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--channels", type=int, default=8)
ap.add_argument("--timepoints", type=int, default=200)
ap.add_argument("--c1", type=int, default=512)
args = ap.parse_args()
dev = torch.device("cuda:0")
stages_def = [(args.c1, 3, True), (512, 3, True), (512, 3, True), (256, 1, True), (64, 1, False)]


def run(fuse):
    eng = CnnEngine(80, args.channels, args.timepoints, 6, 64, 0.0, 0.01, stages_def, [128, 128, 128, 128, 64])
    eng.wino_vout = False
    eng.fuse_c1 = fuse
    eng._alloc(args.batch, dev)
    eng._alloc_bwd()
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(eng.S, args.timepoints, device=dev, generator=g)
    eng._x = x
    for k in sorted(eng.G):
        eng.G[k].normal_(generator=g)
    for k in sorted(eng.bits):
        eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)
        eng.sbits[k].random_(-2**31, 2**31 - 1, generator=g)
    st = eng.stages[0]
    w = torch.randn(st.cout, st.cin, 3, 1, device=dev, generator=g) * 0.02
    gw, gb = torch.zeros_like(w), torch.zeros(st.cout, device=dev)
    if 1 in eng.P:
        eng.P[1].normal_(generator=g)
    eng.generation += 1
    eng._v_ready = {}
    if 1 not in eng.P:                      # V1 from random rows (the fused path keeps no P1)
        P1 = torch.randn(eng.S * eng.tp1, eng.c1, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
        eng.P[1] = P1
    else:
        eng.P[1].copy_(torch.randn(eng.S * eng.tp1, eng.c1, device=dev, generator=torch.Generator(device=dev).manual_seed(9)))
    eng.stage_wgrad(st, gw, gb)
    part = eng.stage_dgrad(st, w)
    if part is None:
        nblk = int(min(2048, eng.S))
        part = torch.empty(nblk, (eng.k1 + 1) * eng.c1, device=dev)
        check(eng.lib.tl_conv1_wgrad(ptr(x), ptr(eng.G[1]), ptr(eng.bits[1]), ptr(part), nblk, eng.S, eng.T, eng.k1, eng.c1,
                                     eng.tp1, eng.tout1, torch.cuda.current_stream().cuda_stream), "tl_conv1_wgrad")
    torch.cuda.synchronize()
    return part.double().sum(0).view(eng.k1 + 1, eng.c1), part


ref, _ = run(False)
got, part = run(True)
for j in range(ref.shape[0]):
    d = (got[j] - ref[j]).norm() / ref[j].norm()
    print(f"tap {j}: rel {float(d):.3e}   ref[:4] {ref[j][:4].tolist()}   got[:4] {got[j][:4].tolist()}")
print("partial tiles", part.shape, "finite", bool(torch.isfinite(part).all()))

# ---- per-tile reference in torch from the un-fused G1 (run again, keep the engine) ----
def ref_tiles():
    eng = CnnEngine(80, args.channels, args.timepoints, 6, 64, 0.0, 0.01, stages_def, [128, 128, 128, 128, 64])
    eng.wino_vout = False
    eng.fuse_c1 = False
    eng._alloc(args.batch, dev)
    eng._alloc_bwd()
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(eng.S, args.timepoints, device=dev, generator=g)
    eng._x = x
    for k in sorted(eng.G):
        eng.G[k].normal_(generator=g)
    for k in sorted(eng.bits):
        eng.bits[k].random_(-2**31, 2**31 - 1, generator=g)
        eng.sbits[k].random_(-2**31, 2**31 - 1, generator=g)
    st = eng.stages[0]
    w = torch.randn(st.cout, st.cin, 3, 1, device=dev, generator=g) * 0.02
    gw, gb = torch.zeros_like(w), torch.zeros(st.cout, device=dev)
    eng.P[1].normal_(generator=g)
    eng.generation += 1
    eng._v_ready = {}
    eng.P[1].copy_(torch.randn(eng.S * eng.tp1, eng.c1, device=dev, generator=torch.Generator(device=dev).manual_seed(9)))
    eng.stage_wgrad(st, gw, gb)
    eng.stage_dgrad(st, w)
    G1 = eng.G[1].double().view(eng.S, eng.tp1, eng.c1)
    bits = eng.bits[1].view(eng.S, eng.tp1, eng.c1 // 32)
    sh = torch.arange(32, device=dev)
    am = ((bits.unsqueeze(-1) >> sh) & 1).reshape(eng.S, eng.tp1, eng.c1).double()          # arg-max bit per element
    t = torch.arange(eng.tp1, device=dev)
    valid = (t < eng.tout1).double().view(1, -1, 1)
    xs = x.double()
    idx = (2 * t).clamp(max=args.timepoints - 4)
    out = []
    for j in range(3):
        x0 = xs[:, idx + j].unsqueeze(-1)            # (S, tp1, 1)
        x1 = xs[:, idx + j + 1].unsqueeze(-1)
        out.append(G1 * valid * (x0 * (1 - am) + x1 * am))
    out.append(G1 * valid)
    contrib = torch.stack(out, 0).reshape(4, eng.S * eng.tp1, eng.c1)     # per row
    ntm = (eng.S * eng.tp1 + 511) // 512
    pad = ntm * 512 - eng.S * eng.tp1
    contrib = torch.nn.functional.pad(contrib, (0, 0, 0, pad))
    return contrib.view(4, ntm, 512, eng.c1).sum(2).permute(1, 0, 2)       # (ntm, 4, c1)

rt = ref_tiles()
pt = part.double().view(part.shape[0], 4, -1)
for tmi in range(min(pt.shape[0], 4)):
    for j in range(4):
        d = (pt[tmi, j] - rt[tmi, j])
        print(f"tile {tmi} tap {j}: rel {float(d.norm() / rt[tmi, j].norm()):.3e}  worst col {int(d.abs().argmax())}")
# rows: contribution of 64-row blocks of tile 0, tap 3 (bias: sum of G1) to find which rows are wrong
