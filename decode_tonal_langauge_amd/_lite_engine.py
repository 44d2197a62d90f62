"""Host-side plan of the SynthesisLite forward / backward on MI355X (reference
models/synthesis_models.py:201-296): two Conv1d+BatchNorm1d+LeakyReLU+MaxPool1d blocks
(``tl_lite_*`` kernels), the small label LSTM (one launch per direction), concat + dropout, and the
two Linear layers on the fp32-MFMA GEMM kernels.  Layout is the reference's own (B, C, T)."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib
from ._lib import EPI_LRELU, EPI_MASK, EPI_STORE, LOAD_DIRECT, NtParams, TnParams, check, ptr


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


class LiteEngine:
    def __init__(self, output_dim, n_channels, n_timepoints, label_dim, conv_channels, lstm_hidden, dropout,
                 negative_slope):
        if negative_slope < 0:
            raise ValueError("the MI355X path needs negative_slope >= 0")
        if label_dim > 8:
            raise ValueError("label_dim > 8 is not supported by the LSTM input-gradient kernel")
        self.seed_dev = None          # set by the trainer's HIP-graph mode: dropout seed (int64 tensor) in device memory
        self.lib = _lib.load()
        self.out_dim, self.C, self.T, self.in_dim = output_dim, n_channels, n_timepoints, label_dim
        self.CC, self.H = conv_channels, lstm_hidden
        self.p_drop, self.slope = float(dropout), float(negative_slope)
        self.T1, self.T2 = n_timepoints // 2, n_timepoints // 2 // 2
        self.F = conv_channels * self.T2
        self.ldf = _r4(self.F + lstm_hidden)
        self.ldd = _r4(output_dim)
        self.hid = 512
        self.generation = 0
        self._saved_generation = -1
        self.timers = None

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def _permute(self, src, dst, dims, strides, lims=None, nz=1, zs=0, bias=None):
        d = (C.c_int64 * 4)(*dims)
        s = (C.c_int64 * 4)(*strides)
        l = (C.c_int64 * 4)(*(lims if lims is not None else dims))
        check(self.lib.tl_permute_reduce(ptr(src), ptr(dst), d, s, l, nz, zs, ptr(bias), self._stream()),
              "tl_permute_reduce")

    def _nt(self, **kw):
        p = NtParams()
        p.splitk, p.bm, p.J, p.Tp, p.slope = 1, 128, 1, 1, 0.0
        for k, v in kw.items():
            setattr(p, k, v)
        check(self.lib.tl_gemm_nt_window(C.byref(p), self._stream()), "tl_gemm_nt_window")

    def _tn(self, **kw):
        p = TnParams()
        p.splitk, p.J, p.Tp, p.Tvalid = 1, 1, 1, 1
        for k, v in kw.items():
            setattr(p, k, v)
        check(self.lib.tl_gemm_tn_window(C.byref(p), self._stream()), "tl_gemm_tn_window")

    def _colsum(self, G, rows, ncols, ld, dst):
        nc4 = _r4(ncols)
        rpb = max(1, 256 // (nc4 // 4))
        nblk = int(min(256, max(1, rows // (rpb * 4))))
        part = torch.empty(nblk, nc4, dtype=torch.float32, device=G.device)
        check(self.lib.tl_colsum(ptr(G), ptr(part), nblk, rows, nc4, ld, 1, 1, self._stream()), "tl_colsum")
        self._permute(part, dst, (1, 1, 1, ncols), (0, 0, 0, 1), nz=nblk, zs=nc4)

    # ------------------------------------------------------------------ forward
    def forward(self, t: Dict[str, torch.Tensor], x, labels, training: bool, save: bool, seed: int = 0, row0: int = 0):
        B, Cn, T = x.shape
        if Cn != self.C or T != self.T:
            raise ValueError(f"expected ECoG input (B, {self.C}, {self.T}), got {tuple(x.shape)}")
        if labels.dim() != 3 or labels.shape[0] != B or labels.shape[1] != self.in_dim:
            raise ValueError(f"expected labels (B, {self.in_dim}, L), got {tuple(labels.shape)}")
        lib, st = self.lib, self._stream()
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        x = x.contiguous().float()
        self.generation += 1
        self._B, self._x, self._training = B, x, training
        p_drop = self.p_drop if training else 0.0
        seed = (seed + 0x9E3779B97F4A7C15 * int(row0)) & 0xFFFFFFFFFFFFFFFF      # data-parallel shards draw different masks
        self._p_used, self._seed = p_drop, seed
        CC, T1, T2, H = self.CC, self.T1, self.T2, self.H
        nt1, nt2 = (T + 63) // 64, (T1 + 63) // 64
        # block 1
        self.z1 = torch.empty(B, CC, T, **f32)
        part = torch.empty(B * nt1, CC, 2, **f32)
        check(lib.tl_lite_conv_fwd(ptr(x), ptr(t["ecog_conv.0.weight"]), ptr(t["ecog_conv.0.bias"]), ptr(self.z1),
                                   ptr(part), B, Cn, CC, T, 5, 2, st), "tl_lite_conv_fwd")
        self.m1, self.r1 = torch.empty(CC, **f32), torch.empty(CC, **f32)
        check(lib.tl_lite_bn_finalize(ptr(part), ptr(self.m1), ptr(self.r1), ptr(t["ecog_conv.1.running_mean"]),
                                      ptr(t["ecog_conv.1.running_var"]), B * nt1, CC, B * T, 0.1, 1e-5, int(training),
                                      ptr(t["ecog_conv.1.num_batches_tracked"]), st), "tl_lite_bn_finalize")
        self.y1 = torch.empty(B, CC, T1, **f32)
        check(lib.tl_lite_bn_act_pool_fwd(ptr(self.z1), ptr(self.m1), ptr(self.r1), ptr(t["ecog_conv.1.weight"]),
                                          ptr(t["ecog_conv.1.bias"]), ptr(self.y1), B, CC, T, self.slope, st),
              "tl_lite_bn_act_pool_fwd")
        # block 2
        self.z2 = torch.empty(B, CC, T1, **f32)
        part2 = torch.empty(B * nt2, CC, 2, **f32)
        check(lib.tl_lite_conv_fwd(ptr(self.y1), ptr(t["ecog_conv.4.weight"]), ptr(t["ecog_conv.4.bias"]),
                                   ptr(self.z2), ptr(part2), B, CC, CC, T1, 3, 1, st), "tl_lite_conv_fwd")
        self.m2, self.r2 = torch.empty(CC, **f32), torch.empty(CC, **f32)
        check(lib.tl_lite_bn_finalize(ptr(part2), ptr(self.m2), ptr(self.r2), ptr(t["ecog_conv.5.running_mean"]),
                                      ptr(t["ecog_conv.5.running_var"]), B * nt2, CC, B * T1, 0.1, 1e-5,
                                      int(training), ptr(t["ecog_conv.5.num_batches_tracked"]), st), "tl_lite_bn_finalize")
        y2 = torch.empty(B, CC, T2, **f32)
        check(lib.tl_lite_bn_act_pool_fwd(ptr(self.z2), ptr(self.m2), ptr(self.r2), ptr(t["ecog_conv.5.weight"]),
                                          ptr(t["ecog_conv.5.bias"]), ptr(y2), B, CC, T1, self.slope, st),
              "tl_lite_bn_act_pool_fwd")
        # label LSTM
        L = labels.shape[2]
        self._L = L
        self.xl = labels.float().permute(0, 2, 1).contiguous()
        self.act = torch.empty(B, L, 4 * H, **f32)
        self.cs = torch.empty(B, L, H, **f32)
        self.hs = torch.empty(B, L, H, **f32)
        check(lib.tl_lite_lstm_fwd(ptr(self.xl), ptr(t["label_lstm.weight_ih_l0"]), ptr(t["label_lstm.weight_hh_l0"]),
                                   ptr(t["label_lstm.bias_ih_l0"]), ptr(t["label_lstm.bias_hh_l0"]), ptr(self.act),
                                   ptr(self.cs), ptr(self.hs), B, L, H, self.in_dim, st), "tl_lite_lstm_fwd")
        # concat + dropout, fc.1 + LeakyReLU, fc.3
        self.feat = torch.empty(B, self.ldf, **f32)
        if getattr(self, "seed_dev", None) is not None and p_drop > 0:     # HIP-graph mode: the seed lives in device memory
            check(lib.tl_lite_cat_dev(ptr(y2), ptr(self.hs), ptr(self.feat), B, self.F, H, L, self.ldf, p_drop,
                                      ptr(self.seed_dev), st), "tl_lite_cat_dev")
        else:
            check(lib.tl_lite_cat(ptr(y2), ptr(self.hs), ptr(self.feat), B, self.F, H, L, self.ldf, p_drop, seed, st),
                  "tl_lite_cat")
        fh = self.F + H
        w1 = t["fc.1.weight"]
        if self.ldf != fh:
            w1p = torch.empty(self.hid, self.ldf, **f32)
            self._permute(w1, w1p, (1, 1, self.hid, self.ldf), (0, 0, fh, 1), (1, 1, self.hid, fh))
            w1 = w1p
        # fc.1: M = B is small, so the grid comes from split-K (32-row tiles x 4 column tiles x splits)
        bm = 32 if B <= 64 else 128
        tiles = ((B + bm - 1) // bm) * ((self.hid + 127) // 128)
        sk = int(max(1, min((self.ldf + 31) // 32, 256 // tiles)))
        slab = torch.empty(sk, B, self.hid, **f32)
        self._nt(A=ptr(self.feat), Bw=ptr(w1), out=ptr(slab), M=B, A_rows=B, N=self.hid, K=self.ldf, lda=self.ldf,
                 ldb=self.ldf, ldo=self.hid, loader=LOAD_DIRECT, epilogue=EPI_STORE, bm=bm, splitk=sk,
                 slab_stride=B * self.hid)
        self.a1 = torch.empty(B, self.hid, **f32)
        check(lib.tl_splitk_bias_lrelu(ptr(slab), ptr(t["fc.1.bias"]), ptr(self.a1), sk, B * self.hid, self.hid, self.slope, st),
              "tl_splitk_bias_lrelu")
        out = torch.empty(B, self.out_dim, **f32)
        tiles3 = ((B + bm - 1) // bm) * ((self.out_dim + 127) // 128)
        sk3 = int(max(1, min((self.hid + 31) // 32, 64 // tiles3)))
        if sk3 > 1:           # one or two tiles of a 16-step K loop are latency-bound (48 us): split K, reduce + bias after
            slab3 = torch.empty(sk3, B, self.out_dim, **f32)
            self._nt(A=ptr(self.a1), Bw=ptr(t["fc.3.weight"]), out=ptr(slab3), M=B, A_rows=B, N=self.out_dim, K=self.hid,
                     lda=self.hid, ldb=self.hid, ldo=self.out_dim, loader=LOAD_DIRECT, epilogue=EPI_STORE, bm=bm,
                     splitk=sk3, slab_stride=B * self.out_dim)
            self._permute(slab3, out, (1, 1, B, self.out_dim), (0, 0, self.out_dim, 1), nz=sk3, zs=B * self.out_dim,
                          bias=t["fc.3.bias"])
        else:
            self._nt(A=ptr(self.a1), Bw=ptr(t["fc.3.weight"]), bias=ptr(t["fc.3.bias"]), out=ptr(out), M=B, A_rows=B,
                     N=self.out_dim, K=self.hid, lda=self.hid, ldb=self.hid, ldo=self.out_dim, loader=LOAD_DIRECT,
                     epilogue=EPI_STORE, bm=bm)
        if save:
            self._saved_generation = self.generation
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, t: Dict[str, torch.Tensor], dout: torch.Tensor, grads: Dict[str, torch.Tensor],
                 gather_whh=None, whh_factors: bool = False, reduce_rows=None, on_factors=None) -> None:
        if self._saved_generation != self.generation:
            raise RuntimeError("SynthesisLite backward: the forward intermediates were overwritten by a later forward")
        lib, st = self.lib, self._stream()
        B, dev = self._B, dout.device
        f32 = dict(dtype=torch.float32, device=dev)
        CC, T, T1, T2, H, L = self.CC, self.T, self.T1, self.T2, self.H, self._L
        fh = self.F + H
        hid, ldd, ldf = self.hid, self.ldd, self.ldf
        # fc.3
        # the weight-gradient GEMMs write straight into the gradient tensors when their padded extents equal the true
        # ones (out_dim and F + H multiples of 4 - the reference's sizes are): no slab, no copy
        direct3 = ldd == self.out_dim and grads["fc.3.weight"].is_contiguous()
        slab = grads["fc.3.weight"] if direct3 else torch.empty(ldd, hid, **f32)
        # (short reductions: the bias gradient = the column sums of the same operand comes out of the weight-gradient kernel)
        cs3 = ldd == self.out_dim and B <= 512 and ldd * hid <= (1 << 20)
        self._tn(A=ptr(dout), B=ptr(self.a1), slab=ptr(slab), Krows=B, A_rows=B, B_rows=B, Mdim=ldd, Ndim=hid, lda=ldd,
                 ldb=hid, ldc=hid, loader=LOAD_DIRECT, colsum=ptr(grads["fc.3.bias"]) if cs3 else None)
        if not direct3:
            grads["fc.3.weight"].copy_(slab[:self.out_dim])
        if not cs3:
            self._colsum(dout, B, self.out_dim, ldd, grads["fc.3.bias"])
        w2t = torch.empty(hid, ldd, **f32)
        self._permute(t["fc.3.weight"], w2t, (1, 1, hid, ldd), (0, 0, 1, hid), (1, 1, hid, self.out_dim))
        g1 = torch.empty(B, hid, **f32)
        bm = 32 if B <= 64 else 128            # small batches: 32-row tiles, the grid comes from columns / split-K
        self._nt(A=ptr(dout), Bw=ptr(w2t), aux=ptr(self.a1), out=ptr(g1), M=B, A_rows=B, N=hid, K=ldd, lda=ldd, ldb=ldd,
                 ldo=hid, ldaux=hid, loader=LOAD_DIRECT, epilogue=EPI_MASK, slope=self.slope, bm=bm)
        # fc.1
        direct1 = ldf == fh and grads["fc.1.weight"].is_contiguous()
        slab1 = grads["fc.1.weight"] if direct1 else torch.empty(hid, ldf, **f32)
        cs1 = B <= 512 and hid * ldf <= (1 << 20)
        self._tn(A=ptr(g1), B=ptr(self.feat), slab=ptr(slab1), Krows=B, A_rows=B, B_rows=B, Mdim=hid, Ndim=ldf, lda=hid,
                 ldb=ldf, ldc=ldf, loader=LOAD_DIRECT, colsum=ptr(grads["fc.1.bias"]) if cs1 else None)
        if not direct1:
            grads["fc.1.weight"].copy_(slab1[:, :fh])
        if not cs1:
            self._colsum(g1, B, hid, hid, grads["fc.1.bias"])
        w1t = torch.empty(ldf, hid, **f32)
        self._permute(t["fc.1.weight"], w1t, (1, 1, ldf, hid), (0, 0, 1, fh), (1, 1, fh, hid))
        dfeat = torch.empty(B, ldf, **f32)
        tiles = ((B + bm - 1) // bm) * ((ldf + 127) // 128)
        sk = int(max(1, min((hid + 31) // 32, 256 // tiles)))
        if sk > 1:
            slab_d = torch.empty(sk, B, ldf, **f32)
            self._nt(A=ptr(g1), Bw=ptr(w1t), out=ptr(slab_d), M=B, A_rows=B, N=ldf, K=hid, lda=hid, ldb=hid, ldo=ldf,
                     loader=LOAD_DIRECT, epilogue=EPI_STORE, bm=bm, splitk=sk, slab_stride=B * ldf)
            self._permute(slab_d, dfeat, (1, 1, 1, B * ldf), (0, 0, 0, 1), nz=sk, zs=B * ldf)
        else:
            self._nt(A=ptr(g1), Bw=ptr(w1t), out=ptr(dfeat), M=B, A_rows=B, N=ldf, K=hid, lda=hid, ldb=hid, ldo=ldf,
                     loader=LOAD_DIRECT, epilogue=EPI_STORE, bm=bm)
        dy2 = torch.empty(B, CC, T2, **f32)
        dh = torch.empty(B, H, **f32)
        if getattr(self, "seed_dev", None) is not None and self._p_used > 0:
            check(lib.tl_lite_uncat_dev(ptr(dfeat), ptr(dy2), ptr(dh), B, self.F, H, ldf, self._p_used, ptr(self.seed_dev), st),
                  "tl_lite_uncat_dev")
        else:
            check(lib.tl_lite_uncat(ptr(dfeat), ptr(dy2), ptr(dh), B, self.F, H, ldf, self._p_used, self._seed, st),
                  "tl_lite_uncat")
        # LSTM
        dg = torch.empty(B, L, 4 * H, **f32)
        check(lib.tl_lite_lstm_bwd(ptr(dh), ptr(t["label_lstm.weight_hh_l0"]), ptr(self.act), ptr(self.cs), ptr(dg), B, L,
                                   H, H, st), "tl_lite_lstm_bwd")
        gwhh = grads["label_lstm.weight_hh_l0"]
        if L > 1:
            kr = B * L - 1          # pairs (dg[b,t], h[b,t-1]): A row R+1, B row R, rows with R%L == L-1 masked
            h4 = _r4(H)
            if h4 == H:
                # a few hundred rows: the reduction is split over 64-row chunks (tn_short_kernel), the slabs summed in order
                sk = max(1, min(8, (kr + 63) // 64))
                slab = gwhh if sk == 1 else torch.empty(sk, 4 * H, H, **f32)
                self._tn(A=dg.data_ptr() + 4 * 4 * H, B=ptr(self.hs), slab=ptr(slab), Krows=kr, A_rows=kr, B_rows=kr,
                         Mdim=4 * H, Ndim=H, lda=4 * H, ldb=H, ldc=H, loader=LOAD_DIRECT, Tp=L, Tvalid=L - 1, splitk=sk,
                         slab_stride=4 * H * H)
                if sk > 1:
                    self._permute(slab, gwhh, (1, 1, 1, 4 * H * H), (0, 0, 0, 1), nz=sk, zs=4 * H * H)
            else:
                raise ValueError("lstm_hidden must be a multiple of 4 on the MI355X path")
        else:
            gwhh.zero_()
        gb = grads["label_lstm.bias_ih_l0"]
        check(lib.tl_lstm_ih_grad(ptr(dg), ptr(self.xl), ptr(grads["label_lstm.weight_ih_l0"]), ptr(gb),
                                  ptr(grads["label_lstm.bias_hh_l0"]), 1, B * L, H, self.in_dim, st), "tl_lstm_ih_grad")
        # block 2
        work = torch.empty(B * CC * 2 + CC * 2, **f32)
        dz2 = torch.empty(B, CC, T1, **f32)
        check(lib.tl_lite_bn_act_pool_bwd(ptr(dy2), ptr(self.z2), ptr(self.m2), ptr(self.r2), ptr(t["ecog_conv.5.weight"]),
                                          ptr(t["ecog_conv.5.bias"]), ptr(dz2), ptr(grads["ecog_conv.5.weight"]),
                                          ptr(grads["ecog_conv.5.bias"]), ptr(work), B, CC, T1, self.slope,
                                          int(self._training), st), "tl_lite_bn_act_pool_bwd")
        dy1 = torch.empty(B, CC, T1, **f32)
        n2 = CC * CC * 3
        dwp = torch.empty(B, n2, **f32)
        dbp = torch.empty(B, CC, **f32)
        check(lib.tl_lite_conv_bwd(ptr(dz2), ptr(self.y1), ptr(t["ecog_conv.4.weight"]), ptr(dy1), ptr(dwp), ptr(dbp), B,
                                   CC, CC, T1, 3, 1, st), "tl_lite_conv_bwd")
        check(lib.tl_sum_slabs2(ptr(dwp), ptr(grads["ecog_conv.4.weight"]), n2, ptr(dbp), ptr(grads["ecog_conv.4.bias"]), CC, B, st),
              "tl_sum_slabs2")
        # block 1
        dz1 = torch.empty(B, CC, T, **f32)
        check(lib.tl_lite_bn_act_pool_bwd(ptr(dy1), ptr(self.z1), ptr(self.m1), ptr(self.r1), ptr(t["ecog_conv.1.weight"]),
                                          ptr(t["ecog_conv.1.bias"]), ptr(dz1), ptr(grads["ecog_conv.1.weight"]),
                                          ptr(grads["ecog_conv.1.bias"]), ptr(work), B, CC, T, self.slope,
                                          int(self._training), st), "tl_lite_bn_act_pool_bwd")
        n1 = CC * self.C * 5
        dwp1 = torch.empty(B, n1, **f32)
        check(lib.tl_lite_conv_bwd(ptr(dz1), ptr(self._x), ptr(t["ecog_conv.0.weight"]), None, ptr(dwp1), ptr(dbp), B,
                                   self.C, CC, T, 5, 2, st), "tl_lite_conv_bwd")
        check(lib.tl_sum_slabs2(ptr(dwp1), ptr(grads["ecog_conv.0.weight"]), n1, ptr(dbp), ptr(grads["ecog_conv.0.bias"]), CC, B, st),
              "tl_sum_slabs2")
