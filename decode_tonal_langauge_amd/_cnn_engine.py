"""Host-side plan of the SynthesisModelCNN forward / backward on MI355X.

Owns the geometry (padded row strides per stage), the HBM workspaces and the order in which the
HIP entry points of ``libtonal_hip.so`` are enqueued.  Everything numerical happens inside those
kernels; torch is used for buffer allocation, the label de-duplication (``torch.unique``) and the
current stream only.

Data layout: activations are channels-last, sequence-major: one *sequence* is one
(batch element, ECoG channel) pair, ``seq = b*C + c``; a stage's tensor is a row-major matrix
``[seq*Tp + t][channel]`` with ``Tp`` (rows per sequence) padded so every max-pool pair is
row-aligned.  In that layout the (k,1) convolution of the reference
(models/synthesis_models.py:86-105) is a GEMM whose A rows are overlapping windows of the
activation matrix - no im2col copy exists anywhere.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, List, Optional

import torch

from . import _kernels, _lib
from ._lib import (EPI_C1WGRAD, EPI_GY, EPI_LRELU, EPI_MASK, EPI_MASKY, EPI_POOL, EPI_POOLV, EPI_STORE, LOAD_DIRECT, LOAD_UNPOOL, LOAD_V,
                   LOAD_Y, NtParams, TnParams, check, ptr)


def _r4(n: int) -> int:
    return (n + 3) // 4 * 4


class _Stage:
    """One ecog_conv_block stage (conv (k,1) + LeakyReLU [+ MaxPool (2,1)])."""

    def __init__(self, idx, cin, cout, k, pool, tin, tp_in):
        self.idx, self.cin, self.cout, self.k, self.pool = idx, cin, cout, k, pool
        self.tin, self.tp_in = tin, tp_in
        self.tc = tin - k + 1
        self.tout = self.tc // 2 if pool else self.tc
        self.tp_out = tp_in // 2 if pool else tp_in


class CnnEngine:
    #: the forward-only classifier engine (a subclass that walks the stages itself) keeps the F(4,3) geometry
    F63_CAPABLE = True

    def __init__(self, output_dim: int, n_channels: int, n_timepoints: int, lstm_channels: int,
                 conv_channels: int, dropout: float, negative_slope: float, stage_defs, concat_widths):
        if negative_slope < 0:
            raise ValueError("the MI355X path needs negative_slope >= 0 (max-pool / LeakyReLU are fused)")
        self.lib = _lib.load()
        self.out_dim = output_dim
        self.C = n_channels
        self.T = n_timepoints
        self.Lc = lstm_channels
        self.Cc = conv_channels
        self.p_drop = float(dropout)
        self.slope = float(negative_slope)
        self.cslope = 0.1                      # concat block slope, models/synthesis_models.py:118-130
        # ---- geometry ----
        k1 = stage_defs[0][1]
        npool_after = sum(1 for s in stage_defs[1:] if s[2])
        tc1 = n_timepoints - k1 + 1
        self.k1 = k1
        self.c1 = stage_defs[0][0]
        self.tout1 = tc1 // 2
        if not stage_defs[0][2]:
            raise ValueError("first stage must pool")
        align = 1 << npool_after
        self.tp1 = max(align, (self.tout1 + align - 1) // align * align)
        # TONAL_WINO=6: Winograd F(6,3) on pre-transformed operands for stages 2 and 3 (csrc/tonal_wino63.hip; 8 products per
        # 6 conv rows).  A sequence of stage 2 holds a multiple of 12 rows (hexes of 6 rows, pooled into hexes of stage 3);
        # the pooled output of stage 3 keeps the row stride of the default geometry (tl_nt_params.out_tp), so everything from
        # stage 4 on is unchanged.  Shapes the form does not cover fall back to TONAL_WINO=4 as a whole.
        _kernels.validate()                    # TONAL_KERNELS: unknown keys / values raise here, not on the hot path
        self.wino63 = (self.F63_CAPABLE and _kernels.get("wino") == "6"
                       and self._f63_covers(stage_defs, n_timepoints))
        tp1_default = self.tp1
        if self.wino63:
            self.tp1 = (self.tout1 + 11) // 12 * 12
        # the input gradient of stage 3 writes the operands of stage 2's backward - Y2 = A dz and Vd2 - instead of the gradient
        # rows G2 (epilogue 6 of tl_conv3_wino63v_nt): the weight gradient of stage 2 then runs without a transform
        # (tl_conv3_wino63v_tn, loader 3; needs C_in of stage 2 % 256 == 0).  TONAL_F63_YPROD=0: off (G2 is stored, the
        # weight-gradient kernel un-pools and transforms it itself, as stage 3's does)
        self.f63_yprod = (self.wino63 and _kernels.get("f63_yprod") != "0" and stage_defs[0][0] % 256 == 0
                          and self.tp1 >= 12)
        # ... and stage 3's (whose gradient rows no Winograd epilogue produces) from a kernel of its own, tl_wino63_unpool_yvd
        # (TONAL_F63_YPROD3=0: its weight-gradient kernel un-pools and transforms G3 itself and writes Vd3)
        self.f63_yprod3 = (self.wino63 and _kernels.get("f63_yprod") != "0" and stage_defs[1][0] % 256 == 0)
        self.stages: List[_Stage] = []
        cin, tin, tp = self.c1, self.tout1, self.tp1
        for i, (cout, k, pool) in enumerate(stage_defs[1:], start=2):
            st = _Stage(i, cin, cout, k, pool, tin, tp)
            if st.tout < 1:
                raise ValueError("n_timepoints too small for the conv stack")
            if self.wino63 and i == 3:
                st.tp_out = tp1_default // 4               # the default geometry's rows per sequence behind stage 3
            self.stages.append(st)
            cin, tin, tp = cout, st.tout, st.tp_out
        self.gy4 = self._gy_applies()          # (fixed here: _alloc_bwd leaves out the gradient rows this path never stores)
        self.lat = tin
        self.tp5 = tp
        self.H = self.lat * n_channels * lstm_channels
        self.ld5 = _r4(conv_channels)
        self.ldx = _r4(conv_channels + lstm_channels)
        self.concat_dims = []                  # (cin_true, cin_ld, cout_true, cout_ld)
        cin_t, cin_ld = conv_channels + lstm_channels, self.ldx
        for w in concat_widths:
            self.concat_dims.append((cin_t, cin_ld, w, _r4(w)))
            cin_t, cin_ld = w, _r4(w)
        self.ldy5 = self.concat_dims[-1][3]
        self.kflat = n_channels * self.tp5 * self.ldy5
        self.ldd = _r4(output_dim)
        self.lowrank_param = "label_lstm.weight_hh_l0"   # reduced via gathered factors under DP
        # data-parallel row shard of the label LSTM (rank, world) - set by the trainer (parallel.py docstring);
        # used for a step when the caller hands the label table (identical distinct rows on every rank)
        self.lstm_shard = None
        self._sh = None
        self.timers = None
        # Kernels for the pooled 3-tap stages (TONAL_KERNELS wino):
        #   6  default: Winograd F(6,3) on pre-transformed operands for all three passes of stages 2 and 3 where the stack
        #      allows it (_f63_covers; 4/9 of the direct-form MFMA work); the F(4,3) V form below, stage by stage, elsewhere
        #   4  Winograd F(4,3) on pre-transformed operands (tonal_wino43v.hip; 1/2 of the MFMA work): the A/B partner
        #   0  direct-form MFMA kernels (the parity partner, and the fallback for every shape neither V form covers)
        # (the in-loop-transform F(2,3) / F(4,3) kernels of rounds 1-2 were retired in round 6)
        mode = _kernels.get("wino")
        self.wino43 = mode != "0"
        # with V written by the first stage the raw pooled rows P1 (13.4 GB at the north-star shape) have no reader
        # left (the LeakyReLU' mask of the backward pass comes from the 1-bit sign array); store_p1 keeps them anyway
        self.store_p1 = _kernels.get("store_p1") == "1"
        # fold the first stage's weight gradient into the stage-2 input-gradient epilogue (Winograd kernels)
        self.fuse_c1 = True                    # (tests clear it to reach the stand-alone tl_conv1_wgrad)
        # test hooks of the F(4,3) V form (no TONAL_KERNELS keys): wino_vout False - forward epilogues write raw rows, every
        # stage transforms its own input; tn_bm 64 / 127 / 128 - force a C_in tile of the weight-gradient kernel (0: auto)
        self.wino_vout = True
        self.tn_bm = 0
        self._side = None
        self._B = None
        self.generation = 0
        self._saved_generation = -1

    # ------------------------------------------------------------------ buffers
    def _alloc(self, B: int, dev):
        if self._B == B and self._dev == dev:
            return
        self._B, self._dev = B, dev
        S = B * self.C
        self.S = S
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda *s: torch.zeros(*s, **f32)
        zi = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
        self._v_ready = {}     # V tensors already written by the producing kernel in this forward
        self._gy_A = None      # (per-batch / per-device scratch of the NT63 input-gradient paths: re-created on demand)
        self._vhalo = {}
        self.P = {}
        if not (self.wino63 or self._conv1_writes_v()) or self.store_p1:
            self.P[1] = z(S * self.tp1, self.c1)
        self.bits = {1: zi(S * self.tp1, self.c1 // 32)}
        self.sbits = {1: zi(S * self.tp1, self.c1 // 32)}      # "pooled output > 0": the LeakyReLU' mask of backward
        for st in self.stages:
            rows = S * st.tp_out
            ld = st.cout if st.pool else self.ld5
            # raw rows are not stored where the forward epilogue hands the next stage V instead: F(6,3) stage 2 (POOLV), or
            # an F(4,3) stage whose successor reads V - but never F(6,3) stage 3, whose POOL epilogue always writes rows
            no_rows = (self.wino63 and st.idx == 2) or (self._writes_v(st) and not self._f63(st))
            if not no_rows or self.store_p1:
                self.P[st.idx] = z(rows, ld)
            if st.pool:
                self.bits[st.idx] = zi(rows, st.cout // 32)
                self.sbits[st.idx] = zi(rows, st.cout // 32)
        rows5 = S * self.tp5
        self.rows5 = rows5
        self.Xc = z(rows5, self.ldx)
        self.Y = [z(rows5, d[3]) for d in self.concat_dims]
        self.out_slab = None
        self.Yt = {}           # F(6,3): Y = A dz of stage idx, written by the input gradient of the stage above (f63_yprod)
        self._y_ready = {}
        self.V = {}            # F(4,3) input transforms of P[idx] (quads, 6, channels) for the stages that read them
        self.Vd = {}           # ... and of the un-pooled dZ of stage idx (the operand of its input-gradient pass)
        self._vd_ready = {}
        self.G = None          # gradient workspaces are allocated lazily on the first backward

    def _alloc_bwd(self):
        if self.G is not None:
            return
        dev = self._dev
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda *s: torch.zeros(*s, **f32)
        S = self.S
        self.G = {}
        if not self.wino63 and not (self.fuse_c1 and self._c1_fusable() and self._v43(self.stages[0])):
            self.G[1] = z(S * self.tp1, self.c1)      # otherwise G1 never leaves the stage-2 epilogue
        for st in self.stages:
            # (with f63_yprod G2 is never stored; with the NT63 form of stage 4's input gradient neither is G3)
            if not (self.f63_yprod and st.idx == 2) and not (self.gy4 and st.idx == 3):
                self.G[st.idx] = z(S * st.tp_out, st.cout if st.pool else self.ld5)
        self.GY = [z(self.rows5, d[3]) for d in self.concat_dims]
        self.dXc = z(self.rows5, self.ldx)

    def _side_stream(self, dev):
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    # ------------------------------------------------------------------ ABI helpers
    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def enable_timers(self, on: bool = True):
        """Per-launch HIP-event timing of the GEMM kernels (bench.py roofline leg).  Events are
        recorded on the stream the kernels are launched on (torch's current stream)."""
        self.timers = {} if on else None

    def _tick(self, name):
        if getattr(self, "timers", None) is None or name is None:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.timers.setdefault(name, []).append(ev)
        ev[0].record()
        return ev

    def timer_summary(self):
        """{name: (launches, mean ms)} - synchronises."""
        torch.cuda.synchronize()
        out = {}
        for k, evs in (self.timers or {}).items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[k] = (len(ms), sum(ms) / max(len(ms), 1))
        return out

    def _permute(self, src, dst, dims, strides, lims=None, nz=1, zs=0, src_off=0, bias=None):
        d = (C.c_int64 * 4)(*dims)
        s = (C.c_int64 * 4)(*strides)
        l = (C.c_int64 * 4)(*(lims if lims is not None else dims))
        check(self.lib.tl_permute_reduce(src.data_ptr() + 4 * src_off, dst.data_ptr(), d, s, l, nz, zs, ptr(bias),
                                         self._stream()), "tl_permute_reduce")

    def _nt(self, tag=None, fn="tl_gemm_nt_window", **kw):
        p = NtParams()
        p.splitk, p.bm, p.J, p.Tp, p.slope = 1, 128, 1, 1, 0.0
        for k, v in kw.items():
            setattr(p, k, v)
        # bm = 256 (8-wave workgroups) exists but measured slower than two independent 4-wave
        # workgroups per CU (conv2 fwd 127 vs 129, dgrad 114 vs 124 TFLOP/s): not selected.
        ev = self._tick(tag)
        check(getattr(self.lib, fn)(C.byref(p), self._stream()), fn)
        if ev:
            ev[1].record()

    def _tn(self, tag=None, fn="tl_gemm_tn_window", **kw):
        p = TnParams()
        p.splitk, p.J, p.Tp, p.Tvalid = 1, 1, 1, 1
        for k, v in kw.items():
            setattr(p, k, v)
        ev = self._tick(tag)
        check(getattr(self.lib, fn)(C.byref(p), self._stream()), fn)
        if ev:
            ev[1].record()

    @staticmethod
    def _splitk(tiles: int, ksteps: int, target: int = 2048) -> int:
        return int(max(1, min(ksteps, (target + tiles - 1) // tiles, 1024)))

    @staticmethod
    def _splitk_rounds(tiles: int, ksteps: int, slots: int = 512, max_sk: int = 32, min_rounds: float = 1.9) -> int:
        """Split-K factor for the HBM-streaming GEMMs over W_hh: the workgroup count should fill whole
        rounds of the 512 resident workgroup slots (256 CUs x 2) - a partial last round streams at a
        fraction of the bandwidth.  Smallest factor whose last round is >= 95 % full with ``min_rounds``
        rounds or more; failing that the fullest."""
        cands = []
        for sk in range(1, max(1, min(max_sk, ksteps)) + 1):
            n = tiles * sk
            cands.append((sk, n, n / (-(-n // slots) * slots)))
        full = [c for c in cands if c[2] >= 0.95]
        for sk, n, _ in full:
            if n >= min_rounds * slots:
                return sk
        if full:
            return full[-1][0]
        return max(cands, key=lambda c: c[2])[0]

    # ------------------------------------------------------------------ weight packing
    def _pack_conv(self, w, cin_ld, flip_for_dgrad):
        """torch (O, I, J, 1) -> forward pack [J][O][cin_ld] or dgrad pack [J'][I_ld][O_ld] (J flipped)."""
        O, I, J, _ = w.shape
        if not flip_for_dgrad:
            dst = torch.empty(J, O, cin_ld, dtype=torch.float32, device=w.device)
            self._permute(w, dst, (1, J, O, cin_ld), (0, 1, I * J, J), (1, J, O, I))
            return dst
        old = _r4(O)
        dst = torch.empty(J, cin_ld, old, dtype=torch.float32, device=w.device)
        # dst[j'][i][o] = w[o][i][J-1-j']
        self._permute(w, dst, (1, J, cin_ld, old), (0, -1, J, I * J), (1, J, I, O), src_off=J - 1)
        return dst


    @staticmethod
    def _f63_covers(stage_defs, T) -> bool:
        """The F(6,3) kernels cover the stack: stages 2 and 3 are pooled 3-tap convolutions with C_in % 128 == 0 and
        C_out % 64 == 0, the first stage has 1..3 taps, one input channel and a width tl_conv1_fwd_v6 takes, and the fused
        first-stage weight gradient can read its sample windows."""
        if len(stage_defs) < 4:
            return False
        (c1, k1, p1), (c2, k2, p2), (c3, k3, p3) = stage_defs[0], stage_defs[1], stage_defs[2]
        tout1 = (T - k1 + 1) // 2
        tout2 = (tout1 - 2) // 2
        tout3 = (tout2 - 2) // 2
        return (p1 and p2 and p3 and k2 == 3 and k3 == 3 and 1 <= k1 <= 3 and c1 in (128, 256, 512, 1024)
                and c2 % 128 == 0 and c3 % 64 == 0 and tout3 >= 1 and T >= 2 * tout1 + 2)

    def _f63(self, st) -> bool:
        return self.wino63 and st.idx in (2, 3)

    def _nt63_rows(self) -> int:
        """conv rows of a row tile of tl_conv3_wino63v_nt: its halo arrays and the fused conv1 gradient's partial sums hold one
        entry per row tile"""
        return int(self.lib.tl_wino63_nt_tile_rows())

    def _v_hex_buffer(self, store, idx, rows, cin):
        """V / Vd of a stage input in hex form: rows / 6 hexes, padded with zero hexes to whole 128-hex tiles (and by at
        least 24: the weight-gradient kernel prefetches three 6-hex K-steps past the last one it uses)."""
        nh = rows // 6
        nh_pad = (nh + 24 + 127) // 128 * 128
        V = store.get(idx)
        if V is None or V.shape[0] != nh_pad or V.shape[1] != 8 or V.shape[2] != cin:
            V = store[idx] = torch.zeros(nh_pad, 8, cin, dtype=torch.float32, device=self._dev)
        return V

    def _pack_wino63(self, w, forward: bool):
        O, I = w.shape[0], w.shape[1]
        dst = torch.empty(8, O, I, dtype=torch.float32, device=w.device) if forward else \
            torch.empty(8, I, O, dtype=torch.float32, device=w.device)
        check(self.lib.tl_wino63_weights(ptr(w), ptr(dst) if forward else None, None if forward else ptr(dst), O, I, I, O,
                                         self._stream()), "tl_wino63_weights")
        return dst

    def f63_issue_factor(self, st) -> float:
        """MFMA FLOPs the F(6,3) kernels issue per direct-convolution FLOP of the stage (8 products per hex, hexes padded
        to whole sequences, against 3 MACs per valid conv row)."""
        return (st.tp_in // 6) * 8.0 / (st.tc * 3.0)

    def _stage_forward63(self, st, w, bia) -> None:
        S = self.S
        wp = self._pack_wino63(w, True)
        V = self._v_ready[st.idx - 1]
        if st.idx == 2 and self.store_p1 and 2 not in self.P:       # (tests: the raw pooled rows of stage 2 as well)
            self.P[2] = torch.zeros(S * st.tp_out, st.cout, dtype=torch.float32, device=self._dev)
        Pout = self.P.get(st.idx) if (st.idx != 2 or self.store_p1) else None
        kw = dict(A=ptr(V), A_rows=V.shape[0], lda=V.shape[2], loader=LOAD_V, Bw=ptr(wp), bias=ptr(bia), out=ptr(Pout),
                  M=S * st.tp_in, N=st.cout, K=st.cin, ldb=st.cin, ldo=Pout.shape[1] if Pout is not None else st.cout, J=3,
                  row_shift=0, Tp=st.tp_in, slope=self.slope, obits=ptr(self.bits[st.idx]), osign=ptr(self.sbits[st.idx]),
                  ld_obits=st.cout // 32, Tvalid=2 * st.tout)
        if st.idx == 2:
            rows_out = S * st.tp_out
            Vn = self._v_hex_buffer(self.V, 2, rows_out, st.cout)
            ntm = -(-(S * st.tp_in) // self._nt63_rows())
            if not hasattr(self, "_vhalo"):
                self._vhalo = {}
            halo = self._vhalo.get(2)
            if halo is None or halo.shape[0] != ntm or halo.shape[2] != st.cout:
                halo = self._vhalo[2] = torch.zeros(ntm, 2, st.cout, dtype=torch.float32, device=self._dev)
            kw.update(epilogue=EPI_POOLV, vout=ptr(Vn), vhalo=ptr(halo), vout_quads=Vn.shape[0], ld_vout=Vn.shape[2])
            self._nt(tag="conv2_fwd", fn="tl_conv3_wino63v_nt", **kw)
            check(self.lib.tl_wino63_v_fixup(ptr(Vn), ptr(halo), rows_out // 6, ntm, st.tp_out, st.cout, Vn.shape[2],
                                             self._stream()), "tl_wino63_v_fixup")
            self._v_ready[2] = Vn
        else:
            if Pout is None:
                raise RuntimeError("F(6,3) stage 3 writes pooled rows: its output buffer was not allocated")
            kw.update(epilogue=EPI_POOL, out_tp=st.tp_out)
            self._nt(tag="conv3_fwd", fn="tl_conv3_wino63v_nt", **kw)

    def _stage_wgrad63(self, st, gw, gb) -> None:
        S = self.S
        f32 = dict(dtype=torch.float32, device=self._dev)
        rows_in = S * st.tp_in
        nd = st.cout
        ldg = nd
        V = self._v_ready[st.idx - 1]
        tiles = (st.cin // 64) * (nd // 64)
        sk = self._splitk(tiles, (rows_in + 35) // 36, 4096)
        slab = torch.empty(sk, 8 * st.cin, ldg, **f32)
        bias_part = torch.empty(sk, nd, **f32)
        if (st.idx == 3 and self.f63_yprod3 and self._y_ready.get(3) != self.generation):
            # stage 3's gradient rows come out of stage 4's one-tap GEMM: a kernel of its own un-pools and transforms them
            # (Y3, Vd3), and the weight gradient below runs without a transform like stage 2's
            Gs = self.G[3]
            Y3 = self._v_hex_buffer(self.Yt, 3, rows_in, nd)
            Vd3 = self._v_hex_buffer(self.Vd, 3, rows_in, nd)
            ev = self._tick("conv3_yvd")
            check(self.lib.tl_wino63_unpool_yvd(ptr(Gs), ptr(self.bits[3]), ptr(Y3), ptr(Vd3), rows_in, Gs.shape[0], st.tp_in,
                                                st.tp_out, 2 * st.tout, nd, Gs.shape[1], st.cout // 32, nd, self._stream()),
                  "tl_wino63_unpool_yvd")
            if ev:
                ev[1].record()
            self._y_ready[3] = self.generation
            self._vd_ready[3] = self.generation
        if self._y_ready.get(st.idx) == self.generation:
            # both operands pre-transformed: Y (and Vd) of this stage were written by the input gradient of the stage above
            Y = self.Yt[st.idx]
            self._y_ready[st.idx] = -1
            self._tn(tag=f"conv{st.idx}_wgrad", fn="tl_conv3_wino63v_tn", A=ptr(V), B=ptr(Y), slab=ptr(slab), Krows=rows_in,
                     A_rows=V.shape[0], B_rows=Y.shape[0], Mdim=st.cin, Ndim=nd, lda=V.shape[2], ldb=Y.shape[2], ldc=ldg, J=3,
                     Tp=st.tp_in, splitk=sk, slab_stride=8 * st.cin * ldg, loader=LOAD_Y, Tvalid=2 * st.tout,
                     colsum=ptr(bias_part))
        else:
            Gs = self.G[st.idx]
            Vd = self._v_hex_buffer(self.Vd, st.idx, rows_in, nd)
            self._tn(tag=f"conv{st.idx}_wgrad", fn="tl_conv3_wino63v_tn", A=ptr(V), B=ptr(Gs), slab=ptr(slab), Krows=rows_in,
                     A_rows=V.shape[0], B_rows=Gs.shape[0], Mdim=st.cin, Ndim=nd, lda=V.shape[2], ldb=Gs.shape[1], ldc=ldg, J=3,
                     Tp=st.tp_in, splitk=sk, slab_stride=8 * st.cin * ldg, loader=LOAD_UNPOOL, bbits=ptr(self.bits[st.idx]),
                     ld_bbits=st.cout // 32, Tvalid=2 * st.tout, colsum=ptr(bias_part), vd=ptr(Vd), ld_vd=nd, g_tp=st.tp_out)
            self._vd_ready[st.idx] = self.generation
        if sk > 1:
            red = torch.empty(8 * st.cin, ldg, **f32)
            n = 8 * st.cin * ldg
            self._permute(slab, red, (1, 1, 1, n), (0, 0, 0, 1), nz=sk, zs=n)
        else:
            red = slab
        check(self.lib.tl_wino63_wgrad_finalize(ptr(red), ptr(gw), st.cout, st.cin, ldg, self._stream()),
              "tl_wino63_wgrad_finalize")
        self._permute(bias_part, gb, (1, 1, 1, st.cout), (0, 0, 0, 1), nz=sk, zs=nd)

    def _gy_stage(self, st) -> bool:
        """The one-tap pooled stage right behind F(6,3) stage 3 (conv4 of the reference stack) whose input gradient runs on the
        NT63 kernel and writes stage 3's backward operands Y3 / Vd3 itself (tl_conv1_wino63v_dgrad_nt; TONAL_KERNELS
        conv4_dgrad=gemm: the one-tap GEMM + tl_wino63_unpool_yvd of round 4)."""
        return self.gy4 and st.idx == 4

    def _gy_applies(self) -> bool:
        if not (self.wino63 and self.f63_yprod3 and len(self.stages) >= 3):
            return False
        below, st = self.stages[1], self.stages[2]
        return (_kernels.get("conv4_dgrad") == "nt63" and st.k == 1 and st.pool and self._f63(below)
                and st.cout % 32 == 0 and st.cout >= 40 and st.cin % 32 == 0 and (below.tp_in // 2) % 3 == 0)

    def _stage_dgrad_gy(self, st, w):
        S = self.S
        below = self.stages[1]                                   # the 3-tap stage whose pooled output this stage reads
        tpg = below.tp_in // 2                                   # gradient rows per sequence in ITS hex geometry (3 per hex)
        rows = S * tpg
        f32 = dict(dtype=torch.float32, device=self._dev)
        nh = -(-rows // 6)
        nh_pad = (nh + 127) // 128 * 128
        A = getattr(self, "_gy_A", None)
        if A is None or A.shape[0] != nh_pad or A.shape[2] != st.cout:
            A = self._gy_A = torch.zeros(nh_pad, 8, st.cout, **f32)   # slots 6, 7 and the pad hexes stay zero
        Gs = self.G[st.idx]
        ev = self._tick(f"conv{st.idx}_dgrad")
        check(self.lib.tl_wino63_unpool_rows6(ptr(Gs), ptr(self.bits[st.idx]), ptr(A), rows, Gs.shape[0], tpg, st.tp_out,
                                              2 * st.tout, st.cout, Gs.shape[1], st.cout // 32, st.cout, 0, self._stream()),
              "tl_wino63_unpool_rows6")
        taps = torch.empty(st.cout // 8, 8, st.cin, 8, **f32)
        check(self.lib.tl_wino63_weights1(ptr(w), ptr(taps), st.cout, st.cin, st.cout, self._stream()), "tl_wino63_weights1")
        rows3 = S * below.tp_in                                  # conv rows of the stage below: six per hex of ITS geometry
        Y3 = self._v_hex_buffer(self.Yt, below.idx, rows3, below.cout)
        Vd3 = self._v_hex_buffer(self.Vd, below.idx, rows3, below.cout)
        ntm = -(-rows // self._nt63_rows())
        if not hasattr(self, "_vhalo"):
            self._vhalo = {}
        halo = self._vhalo.get("d3")
        if halo is None or halo.shape[0] != ntm or halo.shape[2] != below.cout:
            halo = self._vhalo["d3"] = torch.zeros(ntm, 2, below.cout, **f32)
        self._nt(tag=None, fn="tl_conv1_wino63v_dgrad_nt", A=ptr(A), A_rows=A.shape[0], lda=A.shape[2], loader=LOAD_V,
                 Bw=ptr(taps), M=rows, N=st.cin, K=st.cout, ldb=st.cout, ldo=st.cin, J=1, row_shift=0, Tp=tpg, slope=self.slope,
                 auxbits=ptr(self.sbits[below.idx]), ld_auxbits=self.sbits[below.idx].shape[1], abits=ptr(self.bits[below.idx]),
                 ld_abits=below.cout // 32, out_tp=below.tp_out, Tvalid_in=2 * below.tout, epilogue=EPI_GY, out=None,
                 vout=ptr(Y3), vout2=ptr(Vd3), vhalo=ptr(halo), vout_quads=Y3.shape[0], ld_vout=Y3.shape[2])
        check(self.lib.tl_wino63_vd_fixup(ptr(Vd3), ptr(halo), rows // 3, ntm, below.tp_in // 6, below.cout, Vd3.shape[2],
                                          self._stream()), "tl_wino63_vd_fixup")
        if ev:
            ev[1].record()
        self._y_ready[below.idx] = self.generation
        self._vd_ready[below.idx] = self.generation
        return None

    def _stage_dgrad63(self, st, w):
        S = self.S
        rows_in = S * st.tp_in
        if self._vd_ready.get(st.idx) != self.generation:
            raise RuntimeError("F(6,3) input gradient: the stage's weight-gradient pass (which writes Vd) must run first")
        self._vd_ready[st.idx] = -1
        Vd = self.Vd[st.idx]
        wd = self._pack_wino63(w, False)                   # [8][cin][cout]
        kw = dict(A=ptr(Vd), A_rows=Vd.shape[0], lda=Vd.shape[2], loader=LOAD_V, Bw=ptr(wd), M=rows_in, N=st.cin,
                  K=st.cout, ldb=st.cout, ldo=st.cin, J=3, row_shift=-2, Tp=st.tp_in, slope=self.slope,
                  auxbits=ptr(self.sbits[st.idx - 1]), ld_auxbits=self.sbits[st.idx - 1].shape[1])
        if st.idx == 3 and self.f63_yprod:
            below = self.stages[0]
            rows2 = S * below.tp_in                              # conv rows of stage 2: six per hex = three of this GEMM's rows
            Y2 = self._v_hex_buffer(self.Yt, 2, rows2, below.cout)
            Vd2 = self._v_hex_buffer(self.Vd, 2, rows2, below.cout)
            ntm = -(-rows_in // self._nt63_rows())
            if not hasattr(self, "_vhalo"):
                self._vhalo = {}
            halo = self._vhalo.get("d2")
            if halo is None or halo.shape[0] != ntm or halo.shape[2] != below.cout:
                halo = self._vhalo["d2"] = torch.zeros(ntm, 2, below.cout, dtype=torch.float32, device=self._dev)
            self._nt(tag="conv3_dgrad", fn="tl_conv3_wino63v_nt", epilogue=EPI_MASKY, out=None, vout=ptr(Y2), vout2=ptr(Vd2),
                     vhalo=ptr(halo), vout_quads=Y2.shape[0], ld_vout=Y2.shape[2], abits=ptr(self.bits[2]),
                     ld_abits=below.cout // 32, Tvalid_in=2 * below.tout, **kw)
            check(self.lib.tl_wino63_vd_fixup(ptr(Vd2), ptr(halo), rows_in // 3, ntm, below.tp_in // 6, below.cout, Vd2.shape[2],
                                              self._stream()), "tl_wino63_vd_fixup")
            self._y_ready[2] = self.generation
            self._vd_ready[2] = self.generation
            return None
        if st.idx == 3:
            self._nt(tag="conv3_dgrad", fn="tl_conv3_wino63v_nt", epilogue=EPI_MASK, out=ptr(self.G[2]), **kw)
            return None
        ntm = -(-rows_in // self._nt63_rows())
        part = torch.empty(ntm, (self.k1 + 1) * self.c1, dtype=torch.float32, device=self._dev)
        self._nt(tag="conv2_dgrad", fn="tl_conv3_wino63v_nt", epilogue=EPI_C1WGRAD, out=None, c1x=ptr(self._x),
                 c1bits=ptr(self.bits[1]), c1partial=ptr(part), c1T=self.T, c1kt=self.k1, Tvalid=self.tout1, **kw)
        return part

    def _c1_fusable(self) -> bool:
        # the epilogue reads x[2t + a + j] for j < 3 unconditionally (4 floats from 2t)
        return self.k1 <= 3 and self.T >= 2 * self.tout1 + 2

    def _v43(self, st) -> bool:
        """The stage runs on the F(4,3) V-form kernels, all three passes: forward and weight gradient on V (the input transform
        its producer wrote), input gradient on Vd (the transformed un-pooled dZ its weight-gradient launch writes).  Every
        other shape runs on the direct MFMA kernels."""
        return (self.wino43 and st.k == 3 and st.pool and st.cin % 64 == 0 and st.cout % 32 == 0 and st.tp_in % 4 == 0
                and _r4(st.cout) % 16 == 0)

    def _tn_bm(self, st) -> int:
        """C_in tile of the V-form weight-gradient kernel: 128 (8 waves, the Y side by LDS-DMA: C_in % 128 == 0,
        C_out % 64 == 0) where the shape allows it, else 64 (4 waves)."""
        wide = st.cin % 128 == 0
        dma8 = wide and _r4(st.cout) % 64 == 0 and (st.cout // 32) % 2 == 0
        if self.tn_bm == 127:                  # (the 8-wave kernel that stages Y through registers: bit-identical A/B partner)
            return 127 if wide else 64
        if self.tn_bm in (0, 128):
            return 128 if dma8 else 64
        return 64

    def _conv1_writes_v(self) -> bool:
        """The first stage hands its output to stage 2 as V (tl_conv1_fwd_v) - nothing else reads P1 then."""
        return (self._v43(self.stages[0]) and self.tp1 % 4 == 0 and self.c1 in (128, 256, 512, 1024)
                and (self.fuse_c1 and self._c1_fusable()))

    def _writes_v(self, st) -> bool:
        """The forward pass of this stage writes V of its own output for the next stage (nothing else reads the raw rows:
        the next stage's forward and weight gradient read V, its input gradient's LeakyReLU' mask the 1-bit sign array)."""
        if not (self.wino_vout and st.pool and st.idx - 1 < len(self.stages)):
            return False
        nxt = self.stages[st.idx - 1]                      # stages[k] has idx k + 2
        return self._v43(st) and self._v43(nxt) and st.tp_in % 8 == 0 and nxt.cin == st.cout

    def _pin(self, st):
        """Input activation of a stage, or None when only its V form exists (stage 2 behind tl_conv1_fwd_v)."""
        return self.P.get(st.idx - 1)

    def _v_buffer(self, idx, rows, cin):
        """V of P[idx]: rows / 4 quads, padded with zero quads to whole 128-quad tiles (the weight-gradient kernel
        reads whole 8-quad K-steps, the forward kernel 128-quad tiles)."""
        nq = rows // 4
        nq_pad = (nq + 127) // 128 * 128
        V = self.V.get(idx)
        if V is None or V.shape[0] != nq_pad or V.shape[2] != cin:
            V = self.V[idx] = torch.zeros(nq_pad, 6, cin, dtype=torch.float32, device=self._dev)
        return V

    def _input_transform(self, st):
        """V of the stage's input P[idx-1] (stand-alone transform kernel; stage 2 gets it from tl_conv1_fwd)."""
        src = self.P[st.idx - 1]
        V = self._v_buffer(st.idx - 1, src.shape[0], st.cin)
        ev = self._tick(f"conv{st.idx}_xform")
        check(self.lib.tl_wino43_input_transform(ptr(src), ptr(V), src.shape[0], st.tp_in, st.cin, src.shape[1], st.cin,
                                                 self._stream()), "tl_wino43_input_transform")
        if ev:
            ev[1].record()
        return V

    def wgrad_issue_factor(self, st) -> float:
        """MFMA FLOPs the weight-gradient kernel of a stage issues per direct-convolution FLOP."""
        if self._f63(st):
            return self.f63_issue_factor(st)
        return 0.5 if self._v43(st) else 1.0

    def kernel_families(self):
        """({rocprofv3 kernel family: [timer tags]}, {family: MFMA FLOPs issued per algorithmic FLOP})
        for the conv stages - bench.py prices the HIP-event timers of ``enable_timers`` with it."""
        if self.wino63:
            self._fam_share = {}
            st2, st3 = self.stages[0], self.stages[1]
            f6 = "Winograd F(6,3) on pre-transformed operands, LDS-DMA"
            tn = "wino63v_tn4_kernel<true>" if st3.cin % 256 == 0 and st3.tp_in >= 12 else "wino63v_tn_kernel<true>"
            tn2 = "wino63v_tn4_kernel<true>" if st2.cin % 256 == 0 else "wino63v_tn_kernel<true>"
            fams = {f"wino63v_nt_kernel<POOLV> (conv2 forward, {f6}; writes V of its pooled output for conv3)": ["conv2_fwd"],
                    f"wino63v_nt_kernel<POOL> (conv3 forward, {f6})": ["conv3_fwd"],
                    f"wino63v_nt_kernel<C1WGRAD> (conv2 input gradient + conv1 weight gradient, {f6})": ["conv2_dgrad"],
                    }
            if self.f63_yprod3:
                gy = self.gy4
                src = "the epilogue of conv4's input gradient" if gy else "wino63_unpool_yvd_kernel"
                fams[f"wino63v_tn4y_kernel (conv3 weight gradient, {f6}: both operands by LDS-DMA, no transform in the kernel; Y3 / Vd3 "
                     f"from {src})"] = ["conv3_wgrad"]
            else:
                fams[f"{tn} (conv3 weight gradient, {f6}; also writes Vd)"] = ["conv3_wgrad"]
            if self.f63_yprod:
                fams[f"wino63v_nt_kernel<MASKY> (conv3 input gradient, {f6}; writes Y and Vd of conv2 instead of the gradient rows)"] = ["conv3_dgrad"]
                fams[f"wino63v_tn4y_kernel (conv2 weight gradient, {f6}: both operands by LDS-DMA, no transform in the kernel)"] = ["conv2_wgrad"]
            else:
                fams[f"wino63v_nt_kernel<MASK> (conv3 input gradient, {f6})"] = ["conv3_dgrad"]
                fams[f"{tn2} (conv2 weight gradient, {f6}; also writes Vd)"] = ["conv2_wgrad"]
            issued = {k: self.f63_issue_factor(st2 if "conv2" in k else st3) for k in fams}
            return fams, issued
        self._fam_share = {}         # family -> share of its stages' algorithmic FLOPs it computes (default 1)
        if not all(self._v43(st) for st in self.stages[:2]):
            fams = {"nt_window_kernel<128,UNPOOL,MASK> (conv input-gradient)": ["conv2_dgrad", "conv3_dgrad", "conv4_dgrad"],
                    "nt_window_kernel<128,DIRECT,POOL> (conv forward)": ["conv2_fwd", "conv3_fwd", "conv4_fwd"],
                    "tn3_kernel<UNPOOL> (conv weight-gradient)": ["conv2_wgrad", "conv3_wgrad"]}
            return fams, {k: 1.0 for k in fams}      # (mixed shapes: a stage the V form does not cover runs direct)
        form = "Winograd F(4,3) on pre-transformed operands, LDS-DMA"
        fused = self.fuse_c1 and self._c1_fusable()
        bm = self._tn_bm(self.stages[0])
        extra = {}
        if bm == 128:
            # one launch: its workgroups take turns at writing Vd, the operand of the input gradient
            tn = f"wino43v_tn8_kernel<true> (conv2/conv3 weight gradient, {form}; also writes Vd for the input gradient)"
        else:
            # the op is two launches, named apart by rocprofv3: the first C_in tile (1 / ntm of the MFMA work) also writes Vd
            ntm = (self.stages[0].cin + 63) // 64
            tn = f"wino43v_tn_kernel<false, 2> (conv2/conv3 weight gradient, C_in tiles 1..{ntm - 1} of {ntm}, {form})"
            vdn = f"wino43v_tn_kernel<true, 2> (conv2/conv3 weight gradient, C_in tile 0 of {ntm}, {form}, + writes Vd for the input gradient)"
            extra[vdn] = ["conv2_wgrad_vd", "conv3_wgrad_vd"]
            self._fam_share = {tn: (ntm - 1) / ntm, vdn: 1.0 / ntm}
        fams = {tn: ["conv2_wgrad", "conv3_wgrad"]}
        if self._writes_v(self.stages[0]):
            # stage 2's forward launch also writes V of its output for stage 3 (epilogue 5): its own kernel name
            fams[f"wino43v_nt_kernel<POOLV> (conv2 forward, {form}; writes V of its pooled output for conv3 instead of the raw rows)"] = ["conv2_fwd"]
            fams[f"wino43v_nt_kernel<POOL> (conv3 forward, {form})"] = ["conv3_fwd"]
        else:
            fams[f"wino43v_nt_kernel<POOL> (conv2/conv3 forward, {form})"] = ["conv2_fwd", "conv3_fwd"]
        fams.update(extra)
        if fused:       # the stage-2 launch carries the fused conv1 weight-gradient epilogue: its own kernel name
            fams[f"wino43v_nt_kernel<UNPOOL,C1WGRAD> (conv2 input gradient + conv1 weight gradient, {form})"] = ["conv2_dgrad"]
            fams[f"wino43v_nt_kernel<UNPOOL,MASK> (conv3 input gradient, {form})"] = ["conv3_dgrad"]
        else:
            fams[f"wino43v_nt_kernel<UNPOOL,MASK> (conv2/conv3 input gradient, {form})"] = ["conv2_dgrad", "conv3_dgrad"]
        return fams, {k: 0.5 for k in fams}

    def _pack_wino43(self, w, forward: bool):
        """torch (O, I, 3, 1) -> the 6 F(4,3) taps: forward [6][O][I] or input-gradient [6][I][O]."""
        O, I = w.shape[0], w.shape[1]
        dst = torch.empty(6, O, I, dtype=torch.float32, device=w.device) if forward else \
            torch.empty(6, I, O, dtype=torch.float32, device=w.device)
        check(self.lib.tl_wino43_weights(ptr(w), ptr(dst) if forward else None, None if forward else ptr(dst), O, I, I, O,
                                         self._stream()), "tl_wino43_weights")
        return dst

    # ------------------------------------------------------------------ one ecog stage (2..5)
    STAGE_NAMES = {2: "ecog_conv_block.3", 3: "ecog_conv_block.6", 4: "ecog_conv_block.9", 5: "ecog_conv_block.12"}

    def stage_forward(self, st: _Stage, w: torch.Tensor, bia: torch.Tensor) -> None:
        """conv (k,1) + bias + LeakyReLU (+ max-pool, arg-max bits): P[idx-1] -> P[idx]."""
        if self._f63(st):
            return self._stage_forward63(st, w, bia)
        S = self.S
        v43 = self._v43(st)
        wp = self._pack_wino43(w, True) if v43 else self._pack_conv(w, st.cin, False)
        src = self._pin(st)
        vout = self._writes_v(st)
        if vout and self.store_p1 and st.idx not in self.P:
            self.P[st.idx] = torch.zeros(S * st.tp_out, st.cout, dtype=torch.float32, device=self._dev)
        Pout = self.P.get(st.idx) if (not vout or self.store_p1) else None
        kw = dict(A=ptr(src), Bw=ptr(wp), bias=ptr(bia), out=ptr(Pout), M=S * st.tp_in,
                  A_rows=S * st.tp_in, N=st.cout, K=st.cin, lda=st.cin, ldb=st.cin,
                  ldo=Pout.shape[1] if Pout is not None else st.cout, J=st.k, row_shift=0, Tp=st.tp_in, slope=self.slope,
                  loader=LOAD_DIRECT)
        if st.pool:
            kw.update(epilogue=EPI_POOL, obits=ptr(self.bits[st.idx]), osign=ptr(self.sbits[st.idx]),
                      ld_obits=st.cout // 32, Tvalid=2 * st.tout)
        else:
            kw.update(epilogue=EPI_LRELU, Tvalid=st.tout)
        if not v43:
            self._nt(tag=f"conv{st.idx}_fwd", fn="tl_gemm_nt_window", **kw)
            return
        V = self._v_ready.get(st.idx - 1)
        if V is None:
            V = self._v_ready[st.idx - 1] = self._input_transform(st)
        kw.update(A=ptr(V), A_rows=V.shape[0], lda=V.shape[2], loader=LOAD_V)
        if vout:
            rows_out = S * st.tp_out
            Vn = self._v_buffer(st.idx, rows_out, st.cout)
            ntm = (S * st.tp_in + 511) // 512
            halo = self._vhalo.get(st.idx)
            if halo is None or halo.shape[0] != ntm or halo.shape[2] != st.cout:
                halo = self._vhalo[st.idx] = torch.zeros(ntm, 2, st.cout, dtype=torch.float32, device=self._dev)
            kw.update(epilogue=EPI_POOLV, vout=ptr(Vn), vhalo=ptr(halo), vout_quads=Vn.shape[0], ld_vout=Vn.shape[2])
            self._nt(tag=f"conv{st.idx}_fwd", fn="tl_conv3_wino43v_nt", **kw)
            check(self.lib.tl_wino43_v_fixup(ptr(Vn), ptr(halo), rows_out // 4, ntm, st.tp_out, st.cout, Vn.shape[2],
                                             self._stream()), "tl_wino43_v_fixup")
            self._v_ready[st.idx] = Vn
            return
        self._nt(tag=f"conv{st.idx}_fwd", fn="tl_conv3_wino43v_nt", **kw)

    def _colsum(self, Gm, rows, ncols, ld, Tp, Tvalid, dst):
        nc4 = _r4(ncols)                       # pad columns of G are zero by construction
        rpb = max(1, 256 // (nc4 // 4))
        nblk = int(min(2048, max(1, rows // (rpb * 16))))
        part = torch.empty(nblk, nc4, dtype=torch.float32, device=Gm.device)
        check(self.lib.tl_colsum(ptr(Gm), ptr(part), nblk, rows, nc4, ld, Tp, Tvalid, self._stream()), "tl_colsum")
        self._permute(part, dst, (1, 1, 1, ncols), (0, 0, 0, 1), nz=nblk, zs=nc4)

    def _reduce_c1_partials(self, part: torch.Tensor, gw: torch.Tensor, gb: torch.Tensor) -> None:
        """Partial sums of the first stage's weight / bias gradient, ``part[tile][(k1 + 1) * c1]`` in the layout of
        ``tl_conv1_wgrad`` (tap-major weight sums, then the bias sums), -> torch's (c1, 1, k1, 1) weight and (c1,) bias."""
        f32 = dict(dtype=torch.float32, device=part.device)
        nblk = part.shape[0]
        zs = (self.k1 + 1) * self.c1
        if nblk >= 256 and zs % 4 == 0:
            # many partial rows (one per row tile of the fused epilogue: 8 704 x 2 048 floats at the north-star shape): sum them
            # with the column-sum kernel (reads run along the row) and permute the 2 048 results - the slab-parallel permute
            # reads such a matrix one element per cache line (94 + 78 us for 71 MB)
            red = torch.empty(zs, **f32)
            flat = part.view(-1)
            for c0 in range(0, zs, 1024):
                nc = min(1024, zs - c0)
                self._colsum(flat[c0:], nblk, nc, zs, 1, 1, red[c0:c0 + nc])
            self._permute(red, gw, (1, 1, self.c1, self.k1), (0, 0, 1, self.c1))
            self._permute(red, gb, (1, 1, 1, self.c1), (0, 0, 0, 1), src_off=self.k1 * self.c1)
        else:
            self._permute(part, gw, (1, 1, self.c1, self.k1), (0, 0, 1, self.c1), nz=nblk, zs=zs)
            self._permute(part, gb, (1, 1, 1, self.c1), (0, 0, 0, 1), nz=nblk, zs=zs, src_off=self.k1 * self.c1)

    def stage_wgrad(self, st: _Stage, gw: torch.Tensor, gb: torch.Tensor) -> None:
        """dW, db of one stage from its input P[idx-1] and G[idx] (pooled gradient + arg-max bits)."""
        if self._f63(st):
            return self._stage_wgrad63(st, gw, gb)
        S = self.S
        f32 = dict(dtype=torch.float32, device=self._dev)
        Xin = self._pin(st)
        Gs = self.G[st.idx]
        rows_in = S * st.tp_in
        ldg = Gs.shape[1]
        nd = _r4(st.cout)
        if self._v43(st):
            # F(4,3) on V: 6 transform accumulators; split-K slabs summed afterwards.  (Measured at conv2: the 8-wave kernel,
            # one workgroup per CU, runs 0.3 ms better with 8 rounds of 256 workgroups than with 16 - 42.2 / 42.55 ms)
            tiles = ((st.cin + 63) // 64) * ((nd + 63) // 64)
            bm = self._tn_bm(st)
            sk = self._splitk(tiles, (rows_in + 31) // 32, 4096 if bm in (127, 128) else 8192)
            slab = torch.empty(sk, 6 * st.cin, ldg, **f32)
            bias_part = torch.empty(sk, nd, **f32)     # the kernel's Y1 = sum of the quad's dZ rows doubles as the bias gradient
            V = self._v_ready.get(st.idx - 1)          # normally written in the forward pass
            if V is None:
                V = self._v_ready[st.idx - 1] = self._input_transform(st)
            nq_pad = (rows_in // 4 + 127) // 128 * 128
            Vd = self.Vd.get(st.idx)
            if Vd is None or Vd.shape[0] != nq_pad or Vd.shape[2] != nd:
                Vd = self.Vd[st.idx] = torch.zeros(nq_pad, 6, nd, **f32)
            kw = dict(A=ptr(V), B=ptr(Gs), slab=ptr(slab), Krows=rows_in, A_rows=V.shape[0], B_rows=Gs.shape[0],
                      Mdim=st.cin, Ndim=nd, lda=V.shape[2], ldb=ldg, ldc=ldg, J=3, Tp=st.tp_in, splitk=sk,
                      slab_stride=6 * st.cin * ldg, loader=LOAD_UNPOOL, bbits=ptr(self.bits[st.idx]),
                      ld_bbits=st.cout // 32, Tvalid=2 * st.tout, colsum=ptr(bias_part), bm=bm, vd=ptr(Vd), ld_vd=nd)
            self._vd_ready[st.idx] = self.generation
            if self.timers is not None and bm != 128:
                # two calls so that the two launches of the op get their own HIP-event timers (rocprofv3 names them apart too)
                self._tn(tag=f"conv{st.idx}_wgrad_vd", fn="tl_conv3_wino43v_tn", part=1, **kw)
                self._tn(tag=f"conv{st.idx}_wgrad", fn="tl_conv3_wino43v_tn", part=2, **kw)
            else:
                self._tn(tag=f"conv{st.idx}_wgrad", fn="tl_conv3_wino43v_tn", **kw)
            if sk > 1:
                red = torch.empty(6 * st.cin, ldg, **f32)
                n = 6 * st.cin * ldg
                self._permute(slab, red, (1, 1, 1, n), (0, 0, 0, 1), nz=sk, zs=n)
            else:
                red = slab
            check(self.lib.tl_wino43_wgrad_finalize(ptr(red), ptr(gw), st.cout, st.cin, ldg, self._stream()),
                  "tl_wino43_wgrad_finalize")
            self._permute(bias_part, gb, (1, 1, 1, st.cout), (0, 0, 0, 1), nz=sk, zs=nd)
            return
        if st.k == 3:      # all-taps kernel: 128 x 64 tiles
            tiles = ((st.cin + 127) // 128) * ((nd + 63) // 64)
        else:
            tiles = st.k * ((st.cin + 127) // 128) * ((nd + 127) // 128)
        sk = self._splitk(tiles, (rows_in + 31) // 32, 1024)     # two rounds of 512 resident workgroups
        slab = torch.empty(sk, st.k * st.cin, ldg, **f32)
        kw = dict(A=ptr(Xin), B=ptr(Gs), slab=ptr(slab), Krows=rows_in, A_rows=Xin.shape[0], B_rows=Gs.shape[0],
                  Mdim=st.cin, Ndim=nd, lda=st.cin, ldb=ldg, ldc=ldg, J=st.k, Tp=st.tp_in, splitk=sk,
                  slab_stride=st.k * st.cin * ldg)
        if st.pool:
            kw.update(loader=LOAD_UNPOOL, bbits=ptr(self.bits[st.idx]), ld_bbits=st.cout // 32, Tvalid=2 * st.tout)
        else:
            kw.update(loader=LOAD_DIRECT, Tvalid=st.tout)
        # one-tap direct kernel: the bias gradient rides in the launch (see the 1x1 stack in backward())
        fold = (not st.pool) and st.k == 1 and rows_in > 512 and st.cin > 32
        bpart = torch.empty(sk, nd, **f32) if fold else None
        if fold:
            kw.update(colsum=ptr(bpart))
        self._tn(tag=f"conv{st.idx}_wgrad", **kw)
        # sum the split-K slabs with coalesced reads first ([j][i][o], o contiguous), then permute the
        # small result to torch's (O, I, J, 1)
        if sk > 1:
            red = torch.empty(st.k * st.cin, ldg, **f32)
            n = st.k * st.cin * ldg
            self._permute(slab, red, (1, 1, 1, n), (0, 0, 0, 1), nz=sk, zs=n)
        else:
            red = slab
        self._permute(red, gw, (1, st.cout, st.cin, st.k), (0, 1, ldg, st.cin * ldg))
        if fold:
            self._permute(bpart, gb, (1, 1, 1, st.cout), (0, 0, 0, 1), nz=sk, zs=nd)
        else:
            self._colsum(Gs, Gs.shape[0], st.cout, ldg, st.tp_out, st.tout, gb)

    def stage_dgrad(self, st: _Stage, w: torch.Tensor):
        """G[idx-1] = (dZ[idx] (*) flipped W) * LeakyReLU'(P[idx-1]).

        For stage 2 on the Winograd kernels G[1] is not stored: the epilogue contracts it with the raw
        signal into per-row-tile partial sums of the first stage's weight / bias gradient, which are
        returned (shape (tiles, (k1 + 1) * c1), layout of ``tl_conv1_wgrad``'s partials)."""
        if self._f63(st):
            return self._stage_dgrad63(st, w)
        if self._gy_stage(st):
            return self._stage_dgrad_gy(st, w)
        S = self.S
        Xin = self._pin(st)
        Gs = self.G[st.idx]
        rows_in = S * st.tp_in
        ldg = Gs.shape[1]
        v43 = self._v43(st)
        wd = self._pack_wino43(w, False) if v43 else self._pack_conv(w, st.cin, True)   # [6 or J][cin][r4(cout)]
        kd = wd.shape[2]
        fuse = st.idx == 2 and v43 and self.fuse_c1 and self._c1_fusable()
        kw = dict(A=ptr(Gs), Bw=ptr(wd), aux=ptr(Xin), out=None if fuse else ptr(self.G[st.idx - 1]), M=rows_in,
                  A_rows=Gs.shape[0], N=st.cin, K=kd, lda=ldg, ldb=kd, ldo=st.cin, ldaux=st.cin, J=st.k,
                  row_shift=-(st.k - 1), Tp=st.tp_in, epilogue=EPI_MASK, slope=self.slope)
        if (st.idx - 1) in self.sbits:            # the input of this stage came out of a pooling epilogue
            kw.update(auxbits=ptr(self.sbits[st.idx - 1]), ld_auxbits=self.sbits[st.idx - 1].shape[1])
        if st.pool:
            kw.update(loader=LOAD_UNPOOL, abits=ptr(self.bits[st.idx]), ld_abits=st.cout // 32,
                      Tvalid_in=2 * st.tout)
        else:
            kw.update(loader=LOAD_DIRECT)
        if not v43:
            self._nt(tag=f"conv{st.idx}_dgrad", fn="tl_gemm_nt_window", **kw)
            return None
        part = None
        if fuse:
            ntm = (rows_in + 511) // 512
            part = torch.empty(ntm, (self.k1 + 1) * self.c1, dtype=torch.float32, device=self._dev)
            kw.update(epilogue=EPI_C1WGRAD, out=None, c1x=ptr(self._x), c1bits=ptr(self.bits[1]), c1partial=ptr(part),
                      c1T=self.T, c1kt=self.k1, Tvalid=self.tout1)
        if self._vd_ready.get(st.idx) != self.generation or st.idx not in self.Vd:
            raise RuntimeError("F(4,3) input gradient: the stage's weight-gradient pass (which writes Vd) must run first")
        # the weight-gradient kernel of this stage (run just before) left Vd = B^T (un-pooled dZ rows 4q-2 .. 4q+3)
        Vd = self.Vd[st.idx]
        kw.update(A=ptr(Vd), A_rows=Vd.shape[0], lda=Vd.shape[2], loader=LOAD_V)
        self._vd_ready[st.idx] = -1
        self._nt(tag=f"conv{st.idx}_dgrad", fn="tl_conv3_wino43v_nt", **kw)
        return part

    def _lstm_forward(self, prm, xu, U, L, dev, training, label_table) -> None:
        """The label LSTM on the U distinct label rows (torch's current stream: the forward runs it on a side stream beside
        the convolution stack, which it does not depend on - four 5.4 GB streams of W_hh next to MFMA-bound kernels)."""
        lib, st_ = self.lib, self._stream()
        H = self.H
        f32 = dict(dtype=torch.float32, device=dev)
        self._act = torch.empty(L, U, 4 * H, **f32)
        self._c = torch.empty(L, U, H, **f32)
        self._h = torch.empty(L, U, H, **f32)
        hh = torch.empty(U, 4 * H, **f32) if L > 1 else None
        w_ih, w_hh = prm["label_lstm.weight_ih_l0"], prm["label_lstm.weight_hh_l0"]
        b_ih, b_hh = prm["label_lstm.bias_ih_l0"], prm["label_lstm.bias_hh_l0"]
        bm = 32 if U <= 64 else 128
        # row shard of W_hh for this step: needs the caller's label table (same U rows on every rank), an even
        # split of the 4H gate rows and a factor rank the fused optimiser takes
        self._sh = None
        if (self.lstm_shard is not None and label_table is not None and training and L > 1 and (L - 1) * U <= 64
                and (4 * H) % (4 * self.lstm_shard[1]) == 0):
            rk, wd = self.lstm_shard
            self._sh = (rk * (4 * H // wd), 4 * H // wd, wd)
        nloc = self._sh[1] if self._sh else 4 * H
        # (measured at the north-star shape, 576 column tiles: 7 splits 1.048 ms, 8 0.996, 16 0.990, 32 1.017 - the 32-row kernel
        # keeps three workgroups per CU, so nine rounds of 512 are six whole rounds of 768: profiles/r06_kernel_notes.md 7)
        sk_f = self._splitk_rounds(((U + bm - 1) // bm) * ((nloc + 127) // 128), (H + 31) // 32, min_rounds=8.9) if L > 1 else 1
        slab_f = torch.empty(sk_f, U, nloc, **f32) if sk_f > 1 else None
        if self._sh:
            from . import parallel
            hh_loc = torch.empty(U, nloc, **f32)
            hh_all = torch.empty(self._sh[2], U, nloc, **f32)
        for t in range(L):
            if t > 0 and self._sh:
                # this rank's gate rows of h W_hh^T, all-gathered: every rank then holds the full (U, 4H) product
                r0, R, wd = self._sh
                self._nt(A=ptr(self._h[t - 1]), Bw=w_hh.data_ptr() + 4 * r0 * H, out=ptr(slab_f if sk_f > 1 else hh_loc),
                         M=U, A_rows=U, N=R, K=H, lda=H, ldb=H, ldo=R, loader=LOAD_DIRECT, epilogue=EPI_STORE, bm=bm,
                         splitk=sk_f, slab_stride=U * R)
                if sk_f > 1:
                    self._permute(slab_f, hh_loc, (1, 1, 1, U * R), (0, 0, 0, 1), nz=sk_f, zs=U * R)
                parallel.all_gather_blocks(hh_all, hh_loc)
                self._permute(hh_all, hh, (1, U, wd, R), (0, R, U * R, 1))
            elif t > 0:
                self._nt(A=ptr(self._h[t - 1]), Bw=ptr(w_hh), out=ptr(slab_f if sk_f > 1 else hh), M=U, A_rows=U,
                         N=4 * H, K=H, lda=H, ldb=H, ldo=4 * H, loader=LOAD_DIRECT, epilogue=EPI_STORE, bm=bm,
                         splitk=sk_f, slab_stride=U * 4 * H)
                if sk_f > 1:
                    n = U * 4 * H
                    self._permute(slab_f, hh, (1, 1, 1, n), (0, 0, 0, 1), nz=sk_f, zs=n)
            check(lib.tl_lstm_cell_fwd(ptr(hh) if t > 0 else None, ptr(xu[t]), ptr(w_ih), ptr(b_ih), ptr(b_hh),
                                       ptr(self._c[t - 1]) if t > 0 else None, ptr(self._act[t]), ptr(self._c[t]),
                                       ptr(self._h[t]), U, H, 2, 4 * H, st_), "tl_lstm_cell_fwd")

    # ------------------------------------------------------------------ forward
    def forward(self, prm: Dict[str, torch.Tensor], x: torch.Tensor, labels: torch.Tensor, training: bool,
                save: bool, seed: int = 0, row0: int = 0, label_ids: Optional[torch.Tensor] = None,
                label_table: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``row0``: index of this shard's first window in the global batch (data parallel): the dropout
        hash is indexed by the global element, so N ranks draw the masks of one process.
        ``label_ids`` (B,) int32 + ``label_table`` (U, 2, L): the caller already knows the distinct label
        sequences (the trainer builds them from (tone, syllable) class pairs) - ``labels[b] == label_table[ids[b]]``;
        the LSTM then runs on the table rows and no ``torch.unique`` (a host synchronisation) is needed."""
        B, Cn, T = x.shape
        if Cn != self.C or T != self.T:
            raise ValueError(f"expected ECoG input (B, {self.C}, {self.T}), got {tuple(x.shape)}")
        if labels.dim() != 3 or labels.shape[0] != B or labels.shape[1] != 2:
            raise ValueError(f"expected labels (B, 2, L), got {tuple(labels.shape)}")
        x = x.contiguous().float()
        labels = labels.contiguous().float()
        dev = x.device
        self._alloc(B, dev)
        lib, st_ = self.lib, self._stream()
        S = self.S
        self.generation += 1
        p_drop = self.p_drop if training else 0.0
        self._p_drop_used, self._seed_used = p_drop, seed
        self._drop_row0 = int(row0) * self.C * self.tp5
        self._x = x
        self._v_ready = {}
        # ---- stage 1 (C_in = 1) ----
        w1 = prm["ecog_conv_block.0.weight"].reshape(self.c1, self.k1).contiguous()
        if self.wino63:
            V1 = self._v_hex_buffer(self.V, 1, S * self.tp1, self.c1)
            if self.store_p1 and 1 not in self.P:
                self.P[1] = torch.zeros(S * self.tp1, self.c1, dtype=torch.float32, device=dev)
            ev = self._tick("conv1_fwd")
            check(lib.tl_conv1_fwd_v6(ptr(x), ptr(w1), ptr(prm["ecog_conv_block.0.bias"]),
                                      ptr(self.P[1]) if self.store_p1 else None, ptr(V1), ptr(self.bits[1]), ptr(self.sbits[1]),
                                      S, T, self.k1, self.c1, self.tp1, self.tout1, self.slope, st_), "tl_conv1_fwd_v6")
            if ev:
                ev[1].record()
            self._v_ready[1] = V1
        elif self._conv1_writes_v():
            if self.store_p1 and 1 not in self.P:
                self.P[1] = torch.zeros(S * self.tp1, self.c1, dtype=torch.float32, device=dev)
            V1 = self._v_buffer(1, S * self.tp1, self.c1)
            check(lib.tl_conv1_fwd_v(ptr(x), ptr(w1), ptr(prm["ecog_conv_block.0.bias"]),
                                     ptr(self.P[1]) if self.store_p1 else None, ptr(V1), ptr(self.bits[1]), ptr(self.sbits[1]),
                                     S, T, self.k1, self.c1, self.tp1, self.tout1, self.slope, st_), "tl_conv1_fwd_v")
            self._v_ready[1] = V1
        else:
            check(lib.tl_conv1_fwd(ptr(x), ptr(w1), ptr(prm["ecog_conv_block.0.bias"]), ptr(self.P[1]), ptr(self.bits[1]),
                                   ptr(self.sbits[1]), S, T, self.k1, self.c1, self.tp1, self.tout1, self.slope, st_), "tl_conv1_fwd")
        # ---- label LSTM on the distinct label sequences ----
        L = labels.shape[2]
        flat = labels.reshape(B, 2 * L)
        self._table_labels = label_ids is not None and label_table is not None
        if self._table_labels:
            uniq, inv = label_table.reshape(label_table.shape[0], 2 * L).float(), label_ids
        else:
            uniq, inv = torch.unique(flat, dim=0, return_inverse=True)
        U = uniq.shape[0]
        self._U, self._L = U, L
        self._uid = inv.to(torch.int32).contiguous()
        H = self.H
        xu = uniq.reshape(U, 2, L).permute(2, 0, 1).contiguous()          # (L, U, 2) time-major
        self._xu = xu
        f32 = dict(dtype=torch.float32, device=dev)
        # the LSTM does not depend on the convolution stack: it runs on a side stream beside it (HBM-bound next to
        # MFMA-bound), joined in front of the concat kernel.  Not under the row-sharded data-parallel LSTM (collectives).
        side = None         # (a side stream for the LSTM beside the convolutions measured equal to one stream - round 3; retired)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._lstm_forward(prm, xu, U, L, dev, training, label_table)
        # ---- stages 2..5: windowed implicit GEMM on fp32 MFMA ----
        for st in self.stages:
            self.stage_forward(st, prm[self.STAGE_NAMES[st.idx] + ".weight"], prm[self.STAGE_NAMES[st.idx] + ".bias"])
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        else:
            self._lstm_forward(prm, xu, U, L, dev, training, label_table)
        # ---- dropout + concat ----
        check(lib.tl_concat_pack(ptr(self.P[5]), ptr(self._h[L - 1]), ptr(self._uid), ptr(self.Xc), B, self.C,
                                 self.tp5, self.lat, self.Cc, self.Lc, self.ld5, H, self.ldx, p_drop, seed,
                                 self._drop_row0, st_), "tl_concat_pack")
        # ---- concat 1x1 stack ----
        src = self.Xc
        self._wc = []
        for i, (cin_t, cin_ld, cout_t, cout_ld) in enumerate(self.concat_dims):
            w = prm[f"concat_conv_block.{2 * i}.weight"]
            bia = prm[f"concat_conv_block.{2 * i}.bias"]
            wp = self._pack_conv(w, cin_ld, False)
            self._nt(A=ptr(src), Bw=ptr(wp), bias=ptr(bia), out=ptr(self.Y[i]), M=self.rows5, A_rows=self.rows5,
                     N=cout_t, K=cin_ld, lda=cin_ld, ldb=cin_ld, ldo=cout_ld, loader=LOAD_DIRECT,
                     epilogue=EPI_LRELU, slope=self.cslope, Tp=self.tp5, Tvalid=self.lat)
            src = self.Y[i]
        # ---- output Linear: (B, kflat) x (out_dim, kflat)^T, split-K ----
        wo = prm["output_layer.weight"]
        wpo = torch.empty(self.out_dim, self.kflat, **f32)
        latC = self.lat * self.C
        self._permute(wo, wpo, (self.out_dim, self.C, self.tp5, self.ldy5), (self.Cc * latC, 1, self.C, latC),
                      (self.out_dim, self.C, self.lat, self.Cc))
        self._wpo = wpo
        nkc = (self.kflat + 31) // 32
        tiles = ((B + 127) // 128) * ((self.out_dim + 127) // 128)
        sk = self._splitk(tiles, nkc, 1024)
        slab = torch.empty(sk, B, self.out_dim, **f32)
        self._nt(A=ptr(self.Y[-1]), Bw=ptr(wpo), out=ptr(slab), M=B, A_rows=B, N=self.out_dim, K=self.kflat,
                 lda=self.kflat, ldb=self.kflat, ldo=self.out_dim, loader=LOAD_DIRECT, epilogue=EPI_STORE,
                 splitk=sk, slab_stride=B * self.out_dim)
        out = torch.empty(B, self.out_dim, **f32)
        self._permute(slab, out, (1, 1, B, self.out_dim), (0, 0, self.out_dim, 1), nz=sk, zs=B * self.out_dim,
                      bias=prm["output_layer.bias"])
        if save:
            self._saved_generation = self.generation
        return out

    def _lstm_backward(self, prm, grads, dh_ext, gather_whh, whh_factors, reduce_rows, on_factors) -> None:
        """BPTT of the label LSTM on the distinct rows, its W_ih / bias gradients and the W_hh gradient (dense, or left as
        factors in ``self.whh_factors``) - on torch's current stream.  ``on_factors()`` is called at the end (the trainer
        updates W_hh there, on the same stream)."""
        lib, st_ = self.lib, self._stream()
        dev = self._dev
        f32 = dict(dtype=torch.float32, device=dev)
        H, U, L = self.H, self._U, self._L
        # ---- LSTM BPTT on the distinct rows ----
        sh = self._sh
        if sh:      # every rank runs the same BPTT on the gradient of the GLOBAL batch
            from . import parallel
            parallel.all_reduce_(dh_ext)
        w_hh = prm["label_lstm.weight_hh_l0"]
        ldt = (U + 31) // 32 * 32
        dg = torch.empty(L, U, 4 * H, **f32)
        # dgates . W_hh on tl_lstm_gw (reads dg[t] itself: no transposed copy)
        stream_gw = U <= 8 and H % 4 == 0 and _kernels.get("whh_stream") != "0"
        dgt = torch.zeros(4 * H, ldt, **f32) if (L > 1 and not stream_gw) else None
        dc = [torch.empty(U, H, **f32), torch.empty(U, H, **f32)]
        dhrec = torch.empty(U, H, **f32) if L > 1 else None
        if L > 1 and not stream_gw:
            kloc = sh[1] if sh else 4 * H                      # gate rows this rank contracts over
            if ldt <= 32:        # skinny streaming kernel: 32 x 512 tiles, 16-deep K stages
                sk_h = self._splitk_rounds((H + 511) // 512, (kloc + 15) // 16)
            else:
                sk_h = self._splitk((ldt + 127) // 128 * ((H + 127) // 128), (kloc + 31) // 32, 1024)
            slab_h = torch.empty(sk_h, ldt, H, **f32)
        # single process, factored update: the last BPTT product dh_1 = dgates_2 . W_hh comes out of the W_hh NAdam pass itself
        # (tl_nadam_lowrank_dh): the factors are complete one step before the BPTT ends - h_0 = 0, the gradient has no term for the
        # first step - and that product needs only the OLD weight.  One 5.4 GB stream of W_hh less per step
        fuse_dh = (L > 1 and not sh and gather_whh is None and whh_factors and (L - 1) * U <= 64 and U <= 8
                   and on_factors is not None and getattr(on_factors, "fuse_dh", False) and _kernels.get("whh_dh") != "0")
        fused_done = False
        slab_g = None
        for t in range(L - 1, -1, -1):
            check(lib.tl_lstm_cell_bwd(ptr(dh_ext) if t == L - 1 else None, ptr(dhrec) if t < L - 1 else None,
                                       ptr(dc[(t + 1) & 1]) if t < L - 1 else None, ptr(self._act[t]), ptr(self._c[t]),
                                       ptr(self._c[t - 1]) if t > 0 else None, ptr(dg[t]),
                                       ptr(dgt) if (t > 0 and dgt is not None) else None, ptr(dc[t & 1]), U, H, ldt, st_),
                  "tl_lstm_cell_bwd")
            if t == 1 and fuse_dh:
                kr = (L - 1) * U
                self.whh_factors = (dg[1:].reshape(kr, 4 * H), self._h[:L - 1].reshape(kr, H))
                row_tiles = 16
                nslab = -(-(-(-4 * H // 32)) // row_tiles)
                slab_d = torch.empty(nslab, U, H, **f32)
                on_factors(dh=(slab_d, U, row_tiles))
                self._permute(slab_d, dhrec, (1, 1, U, H), (0, 0, H, 1), nz=nslab, zs=U * H)
                fused_done = True
            elif t > 0 and sh:
                # partial dgates . W_hh over this rank's gate rows, summed over the ranks
                r0, R, _wd = sh
                if stream_gw:
                    rpb = 512
                    nb = -(-R // rpb)
                    if slab_g is None:
                        slab_g = torch.empty(nb, U, H, **f32)
                    check(lib.tl_lstm_gw(dg[t].data_ptr() + 4 * r0, w_hh.data_ptr() + 4 * r0 * H, ptr(slab_g), U, R, H, 4 * H, H,
                                         rpb, st_), "tl_lstm_gw")
                    self._permute(slab_g, dhrec, (1, 1, U, H), (0, 0, H, 1), nz=nb, zs=U * H)
                else:
                    self._tn(A=dgt.data_ptr() + 4 * r0 * ldt, B=w_hh.data_ptr() + 4 * r0 * H, slab=ptr(slab_h), Krows=R,
                             A_rows=R, B_rows=R, Mdim=ldt, Ndim=H, lda=ldt, ldb=H, ldc=H, loader=LOAD_DIRECT, splitk=sk_h,
                             slab_stride=ldt * H)
                    self._permute(slab_h, dhrec, (1, 1, U, H), (0, 0, H, 1), nz=sk_h, zs=ldt * H)
                parallel.all_reduce_(dhrec)
            elif t > 0 and stream_gw:
                # dgates_t . W_hh as a pure stream of the 5.4 GB weight (tl_lstm_gw, round 6): 512-row blocks, the partial
                # products of the 144 blocks summed behind it
                rpb = 512
                nb = -(-4 * H // rpb)
                if slab_g is None:
                    slab_g = torch.empty(nb, U, H, **f32)
                check(lib.tl_lstm_gw(ptr(dg[t]), ptr(w_hh), ptr(slab_g), U, 4 * H, H, 4 * H, H, rpb, st_), "tl_lstm_gw")
                self._permute(slab_g, dhrec, (1, 1, U, H), (0, 0, H, 1), nz=nb, zs=U * H)
            elif t > 0:
                self._tn(A=ptr(dgt), B=ptr(w_hh), slab=ptr(slab_h), Krows=4 * H, A_rows=4 * H, B_rows=4 * H, Mdim=ldt,
                         Ndim=H, lda=ldt, ldb=H, ldc=H, loader=LOAD_DIRECT, splitk=sk_h, slab_stride=ldt * H)
                self._permute(slab_h, dhrec, (1, 1, U, H), (0, 0, H, 1), nz=sk_h, zs=ldt * H)
        def dense_whh():                       # allocated on demand: the trainer path never needs it
            if "label_lstm.weight_hh_l0" not in grads:
                grads["label_lstm.weight_hh_l0"] = torch.empty(4 * H, H, **f32)
            return grads["label_lstm.weight_hh_l0"]
        if fused_done:
            pass                                   # factors handed over and consumed inside the loop
        elif L > 1:
            kr = (L - 1) * U
            fa, fb = dg[1:].reshape(kr, 4 * H), self._h[:L - 1].reshape(kr, H)
            if sh:
                # dgates already belong to the global batch (identical on every rank): this rank updates its own
                # rows of W_hh from the matching columns of the factor - nothing to gather
                if not whh_factors:
                    raise RuntimeError("the row-sharded label LSTM needs the factored W_hh update (whh_factors=True)")
                self.whh_factors = (fa[:, sh[0]:sh[0] + sh[1]], fb.contiguous(), sh[0], sh[1])
                fa = fb = None
            elif gather_whh is not None and self._table_labels and reduce_rows is not None:
                # the label table is the same on every rank: row (t, u) of the factors means the same (step, label
                # sequence) everywhere and carries bit-identical h, so the factor of the GLOBAL gradient is simply the sum
                # of the ranks' dgates rows - one small all-reduce ((L-1) U x 4H floats), no gather, no host sync
                fa = fa.clone()                  # dg itself stays local: the W_ih / bias gradients are reduced with the buckets
                reduce_rows(fa)
            elif gather_whh is not None:
                # arbitrary label tensors: the distinct rows differ per rank (torch.unique above already synchronised)
                # key of row (t, u): the step and the label sequence h_t was unrolled from
                steps = torch.arange(L - 1, device=dev, dtype=torch.float32).repeat_interleave(U).unsqueeze(1)
                keys = torch.cat([steps, self._xu.permute(1, 0, 2).reshape(U, 2 * L).repeat(L - 1, 1)], dim=1)
                fa, fb = gather_whh(fa, fb, keys)
                kr = fa.shape[0]
            if sh:
                pass
            elif whh_factors and kr <= 64:
                # hand the factors to the optimiser (tl_nadam_lowrank): the 5.4 GB gradient is never formed
                self.whh_factors = (fa.contiguous(), fb.contiguous())
            else:
                self._tn(A=ptr(fa), B=ptr(fb), slab=ptr(dense_whh()), Krows=kr, A_rows=kr, B_rows=kr, Mdim=4 * H,
                         Ndim=H, lda=4 * H, ldb=H, ldc=H, loader=LOAD_DIRECT)
            del fa, fb
        elif whh_factors:
            self.whh_factors = (None, None)
        else:
            dense_whh().zero_()
        gb = grads["label_lstm.bias_ih_l0"]
        check(lib.tl_lstm_ih_grad(ptr(dg), ptr(self._xu), ptr(grads["label_lstm.weight_ih_l0"]), ptr(gb),
                                  ptr(grads["label_lstm.bias_hh_l0"]), L, U, H, 2, st_), "tl_lstm_ih_grad")
        del dg, dgt
        if on_factors is not None and not fused_done:
            on_factors()

    # ------------------------------------------------------------------ backward
    def grad_order(self) -> List[str]:
        """Parameter names in the order ``backward`` finishes their gradients (the layout of a flat gradient buffer whose
        exchange buckets are contiguous slices): output layer, 1x1 stack last to first, label LSTM, ecog stages 5..1."""
        order = ["output_layer.weight", "output_layer.bias"]
        for i in range(len(self.concat_dims) - 1, -1, -1):
            order += [f"concat_conv_block.{2 * i}.weight", f"concat_conv_block.{2 * i}.bias"]
        order += ["label_lstm.weight_ih_l0", "label_lstm.bias_ih_l0", "label_lstm.bias_hh_l0"]
        for st in reversed(self.stages):
            order += [self.STAGE_NAMES[st.idx] + ".weight", self.STAGE_NAMES[st.idx] + ".bias"]
        order += ["ecog_conv_block.0.weight", "ecog_conv_block.0.bias"]
        return order

    def backward(self, prm: Dict[str, torch.Tensor], dout: torch.Tensor, grads: Dict[str, torch.Tensor],
                 gather_whh=None, whh_factors: bool = False, reduce_rows=None, on_factors=None, on_grad_ready=None) -> None:
        """dout: (B, ldd) gradient of the loss w.r.t. the output (pad columns zero).
        Fills ``grads[name]`` (torch layouts) for every parameter.  ``gather_whh(dg, h)`` may
        return the low-rank factors of every data-parallel rank (parallel.gather_lowrank): the
        W_hh gradient written is then already the sum over ranks.  With ``whh_factors`` the W_hh
        gradient is not written at all: ``self.whh_factors = (fa, fb)`` (gradient = fa^T . fb) is left for
        ``FusedNAdam.step(lowrank=...)``; ``None`` afterwards means the dense gradient was written instead
        (rank above 64)."""
        self.whh_factors = None
        if self._saved_generation != self.generation:
            raise RuntimeError("SynthesisModelCNN backward: the forward intermediates were overwritten by a later "
                               "forward pass (one forward/backward pair at a time per model)")
        self._alloc_bwd()
        lib, st_ = self.lib, self._stream()
        B, S, dev = self._B, self.S, self._dev
        f32 = dict(dtype=torch.float32, device=dev)
        H, U, L = self.H, self._U, self._L
        rows5 = self.rows5

        colsum = self._colsum

        # ---- output layer ----
        gw = grads["output_layer.weight"]
        slab = torch.empty(self.ldd, self.kflat, **f32)
        self._tn(A=ptr(dout), B=ptr(self.Y[-1]), slab=ptr(slab), Krows=B, A_rows=B, B_rows=B, Mdim=self.ldd,
                 Ndim=self.kflat, lda=self.ldd, ldb=self.kflat, ldc=self.kflat, loader=LOAD_DIRECT)
        latC = self.lat * self.C
        self._permute(slab, gw, (self.out_dim, self.Cc, self.lat, self.C),
                      (self.kflat, 1, self.ldy5, self.tp5 * self.ldy5))
        colsum(dout, B, self.out_dim, self.ldd, 1, 1, grads["output_layer.bias"])
        if on_grad_ready is not None:
            on_grad_ready("output_layer")           # both gradients of the Linear layer are final (enqueued) from here on
        # dY5 = dout . Wp_out, masked by lrelu'(Y5)
        wpt = torch.empty(self.kflat, self.ldd, **f32)
        self._permute(prm["output_layer.weight"], wpt, (self.C, self.tp5, self.ldy5, self.ldd),
                      (1, self.C, latC, self.Cc * latC), (self.C, self.lat, self.Cc, self.out_dim))
        self._nt(A=ptr(dout), Bw=ptr(wpt), aux=ptr(self.Y[-1]), out=ptr(self.GY[-1]), M=B, A_rows=B, N=self.kflat,
                 K=self.ldd, lda=self.ldd, ldb=self.ldd, ldo=self.kflat, ldaux=self.kflat, loader=LOAD_DIRECT,
                 epilogue=EPI_MASK, slope=self.cslope)
        del wpt
        # ---- concat 1x1 stack, last to first ----
        for i in range(len(self.concat_dims) - 1, -1, -1):
            cin_t, cin_ld, cout_t, cout_ld = self.concat_dims[i]
            src = self.Xc if i == 0 else self.Y[i - 1]
            Gi = self.GY[i]
            name = f"concat_conv_block.{2 * i}"
            tiles = ((cin_ld + 127) // 128) * ((cout_ld + 127) // 128)
            # (512 splits: the launch itself is flat between 256 and 1 024 - scripts/bench_tn1.py -, the slab reduction behind
            # it reads half of what 1 024 leave)
            sk = self._splitk(tiles, (rows5 + 31) // 32, 512)
            slab = torch.empty(sk, cin_ld, cout_ld, **f32)
            # the bias gradient (column sums of Gi over the valid rows) rides in the weight-gradient launch where that is the
            # one-tap direct kernel (not its short-reduction / skinny forms, whose colsum means something else): Gi is not read
            # a third time
            fold = rows5 > 512 and cin_ld > 32
            bpart = torch.empty(sk, cout_ld, **f32) if fold else None
            self._tn(A=ptr(src), B=ptr(Gi), slab=ptr(slab), Krows=rows5, A_rows=rows5, B_rows=rows5, Mdim=cin_ld,
                     Ndim=cout_ld, lda=cin_ld, ldb=cout_ld, ldc=cout_ld, loader=LOAD_DIRECT, Tp=self.tp5,
                     Tvalid=self.lat, splitk=sk, slab_stride=cin_ld * cout_ld, colsum=ptr(bpart))
            self._permute(slab, grads[name + ".weight"], (1, 1, cout_t, cin_t), (0, 0, 1, cout_ld), nz=sk,
                          zs=cin_ld * cout_ld)
            if fold:
                self._permute(bpart, grads[name + ".bias"], (1, 1, 1, cout_t), (0, 0, 0, 1), nz=sk, zs=cout_ld)
            else:
                colsum(Gi, rows5, cout_t, cout_ld, self.tp5, self.lat, grads[name + ".bias"])
            wd = self._pack_conv(prm[name + ".weight"], cin_ld, True)          # [1][cin_ld][cout_ld]
            if i > 0:
                self._nt(A=ptr(Gi), Bw=ptr(wd), aux=ptr(src), out=ptr(self.GY[i - 1]), M=rows5, A_rows=rows5,
                         N=cin_ld, K=cout_ld, lda=cout_ld, ldb=cout_ld, ldo=cin_ld, ldaux=cin_ld, loader=LOAD_DIRECT,
                         epilogue=EPI_MASK, slope=self.cslope)
            else:
                self._nt(A=ptr(Gi), Bw=ptr(wd), out=ptr(self.dXc), M=rows5, A_rows=rows5, N=cin_ld, K=cout_ld,
                         lda=cout_ld, ldb=cout_ld, ldo=cin_ld, loader=LOAD_DIRECT, epilogue=EPI_STORE)
        # ---- un-concat: G5 (dropout + lrelu') and dh summed over duplicates ----
        # duplicates of a label sequence: the kernel scans the batch's label ids in batch order (what a stable argsort
        # would list) - no argsort / scatter_add_ / cumsum in front of it
        dh_ext = torch.empty(U, H, **f32)
        check(lib.tl_concat_unpack_bwd(ptr(self.dXc), ptr(self.P[5]), None, ptr(self._uid), ptr(self.G[5]),
                                       ptr(dh_ext), B, U, self.C, self.tp5, self.lat, self.Cc, self.Lc, self.ld5, H,
                                       self.ldx, self.slope, self._p_drop_used, self._seed_used, self._drop_row0,
                                       st_), "tl_concat_unpack_bwd")
        # ---- LSTM BPTT on the distinct rows: independent of the convolution backward below - on the side stream beside it
        # when nothing in it is a collective (single process); the trainer's W_hh update rides along (on_factors) ----
        side = None         # (the same for the BPTT beside the convolution backward)
        if side is not None:
            dh_ext.record_stream(side)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._lstm_backward(prm, grads, dh_ext, gather_whh, whh_factors, reduce_rows, on_factors)
        else:
            self._lstm_backward(prm, grads, dh_ext, gather_whh, whh_factors, reduce_rows, on_factors)
        # ---- ecog stages 5..2 ----
        for st in reversed(self.stages):
            name = self.STAGE_NAMES[st.idx]
            self.stage_wgrad(st, grads[name + ".weight"], grads[name + ".bias"])
            part = self.stage_dgrad(st, prm[name + ".weight"])
        # ---- stage 1 weight / bias gradient (partials come out of the stage-2 epilogue when fused) ----
        if part is None:
            nblk = int(min(2048, S))
            part = torch.empty(nblk, (self.k1 + 1) * self.c1, **f32)
            check(lib.tl_conv1_wgrad(ptr(self._x), ptr(self.G[1]), ptr(self.bits[1]), ptr(part), nblk, S, self.T, self.k1,
                                     self.c1, self.tp1, self.tout1, st_), "tl_conv1_wgrad")
        self._reduce_c1_partials(part, grads["ecog_conv_block.0.weight"], grads["ecog_conv_block.0.bias"])
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
