"""The discrete branches of the HIP backward pass, read back in the oracle's layout - TEST INFRASTRUCTURE.

LeakyReLU' (1 or the slope) and the max-pool arg-max are decided on pre-activations; two correct fp32 implementations whose
pre-activations agree to 2e-6 still take a handful of those branches differently (near-ties), and each flipped branch moves a
conv-stack gradient by O(1) of one element: 1e-3 relative L2 at batch 2.  ``hip_decisions`` reads the branches the HIP path
took (1-bit planes of the pooling epilogues, the sign of the stored stage-5 / 1x1-stack outputs); handed to
``oracle.synthesis_oracle.cnn_forward(decisions=...)`` they make the oracle's backward take the same ones, so that what is
left between the two gradients is arithmetic (held to a few 1e-6)."""
import torch


def bit_plane(words: torch.Tensor, S: int, tp: int, tout: int, B: int, Cn: int) -> torch.Tensor:
    """(S * tp, ch / 32) words of a pooling epilogue (bit c % 32 of word c / 32 = channel c) -> bool (B, ch, tout, C), the
    layout of the reference's activation (models/synthesis_models.py:157-159)."""
    sh = torch.arange(32, device=words.device, dtype=torch.int32)
    b = ((words.view(S, tp, -1)[:, :tout, :, None] >> sh) & 1).bool().reshape(S, tout, -1)
    return b.view(B, Cn, tout, -1).permute(0, 3, 2, 1).contiguous().cpu()


def hip_decisions(eng, B: int, Cn: int) -> dict:
    """Every discrete branch the HIP backward pass takes after a forward pass of ``eng`` (a CnnEngine): arg-max and "pooled
    output > 0" planes of the pooled stages, and the LeakyReLU' masks it derives from the stored stage-5 and 1x1-stack
    outputs."""
    S = B * Cn
    dec = {}
    for i in range(1, len(eng.stages) + 1):
        st = None if i == 1 else eng.stages[i - 2]
        if st is not None and not st.pool:
            continue
        tp = eng.tp1 if i == 1 else st.tp_out
        tout = eng.tout1 if i == 1 else st.tout
        dec[f"ecog{i}.odd"] = bit_plane(eng.bits[i], S, tp, tout, B, Cn)
        dec[f"ecog{i}.pos"] = bit_plane(eng.sbits[i], S, tp, tout, B, Cn)
    v = lambda t, ld, n: (t.view(B, Cn, eng.tp5, ld)[:, :, :eng.lat, :n] > 0).permute(0, 3, 2, 1).contiguous().cpu()
    dec[f"ecog{len(eng.stages) + 1}.pos"] = v(eng.P[len(eng.stages) + 1], eng.ld5, eng.Cc)
    for i, (_cin, _cld, cout, cout_ld) in enumerate(eng.concat_dims):
        dec[f"concat{i + 1}.pos"] = v(eng.Y[i], cout_ld, cout)
    return dec


def count_differing(dec: dict, own: dict):
    """({plane: branches that differ}, {plane: branches})"""
    return ({k: int((dec[k] != own[k]).sum()) for k in sorted(dec)}, {k: int(dec[k].numel()) for k in sorted(dec)})
