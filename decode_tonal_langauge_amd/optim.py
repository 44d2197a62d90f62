"""Fused NAdam on MI355X: one HBM pass per parameter tensor (``tl_nadam``).

Same update rule and defaults as ``torch.optim.NAdam`` built by the reference trainer
(reference models/synthesis_trainer.py:131-137): coupled L2 ``weight_decay`` (the reference's
``schedule_decay`` argument lands there), ``momentum_decay=0.004``, non-decoupled.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch

from . import _lib
from ._lib import check, ptr


def nadam_scalars(step: int, mu_product: float, lr: float, beta1: float, beta2: float, momentum_decay: float):
    """Scalar schedule of torch's ``_single_tensor_nadam`` for 1-based ``step`` (host, float64)."""
    bc2 = 1.0 - beta2 ** step
    mu = beta1 * (1.0 - 0.5 * (0.96 ** (step * momentum_decay)))
    mu_next = beta1 * (1.0 - 0.5 * (0.96 ** ((step + 1) * momentum_decay)))
    mu_product = mu_product * mu
    coef_grad = lr * (1.0 - mu) / (1.0 - mu_product)
    coef_mom = lr * mu_next / (1.0 - mu_product * mu_next)
    return coef_grad, coef_mom, bc2, mu_product


class FusedNAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 2e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, momentum_decay: float = 4e-3):
        if lr < 0 or eps < 0 or weight_decay < 0 or momentum_decay < 0:
            raise ValueError("FusedNAdam: negative hyper-parameter")
        if not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError(f"Invalid beta parameters: {betas}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      momentum_decay=momentum_decay))
        self._lib = _lib.load()
        self._tables = {}
        self.multi_tensor = True       # one launch for all dense tensors at the same point of the schedule (tests may clear it)

    def _state_for(self, p, shard_rows=None):
        """``shard_rows`` = (row0, rows): this rank owns (and keeps moments for) only those rows of ``p``."""
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["mu_product"] = 1.0
            like = p if shard_rows is None else p[shard_rows[0]:shard_rows[0] + shard_rows[1]]
            if like.is_cuda and like.numel() * like.element_size() >= (1 << 30):
                # Moments of a multi-GB parameter (W_hh: 5.4 GB each): give each its OWN device allocation.  Carved out of one
                # cached segment of torch's allocator (whatever the step freed last) the three streams p / m / v of the update
                # kernel run 16 % slower - 6.58 ms against 5.68 ms for the same kernel on the same data, whatever their
                # relative padding (profiles/r06_kernel_notes.md 8) - so the cache is emptied first: the two requests below then
                # become two fresh allocations of exactly their size.  Once per parameter, at its first step.
                torch.cuda.empty_cache()
            st["exp_avg"] = torch.zeros_like(like, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(like, memory_format=torch.preserve_format)
            st["shard_rows"] = shard_rows
        elif st.get("shard_rows") != shard_rows:
            raise RuntimeError("FusedNAdam: the row shard of a parameter cannot change between steps")
        return st

    LOWRANK_MAX = 64          # largest factor rank tl_nadam_lowrank takes

    @torch.no_grad()
    def step_lowrank(self, lowrank: Dict[torch.nn.Parameter, tuple], grad_scale: float = 1.0, dh=None) -> None:
        """The low-rank part of ``step`` alone, on torch's current stream: lets the caller update such a parameter as soon
        as its factors exist (the trainer runs it on a side stream beside the convolution backward) and hand the same
        parameters to ``step(..., skip=...)`` afterwards.

        ``dh = (slab (n, U, cols), U, row_tiles)`` (one parameter only): the same pass writes the partial sums of
        ``fa[0:U] . p_old`` - the product of the factor's first U rows with the parameter as it was BEFORE this update
        (``tl_nadam_lowrank_dh``: the label LSTM's last BPTT step rides in the W_hh update); ``slab.sum(0)`` is the product."""
        stream = torch.cuda.current_stream().cuda_stream
        if dh is not None and len(lowrank) != 1:
            raise RuntimeError("FusedNAdam.step_lowrank: the fused product takes exactly one parameter")
        for group in self.param_groups:
            for p in group["params"]:
                if p in lowrank:
                    self._step_lowrank(p, group, lowrank[p], grad_scale, stream, dh)

    def _step_lowrank(self, p, group, spec, grad_scale, stream, dh=None) -> None:
        b1, b2 = group["betas"]
        fa, fb = spec[0], spec[1]
        shard = (int(spec[2]), int(spec[3])) if len(spec) > 2 else None
        kr = 0 if fa is None else fa.shape[0]
        if p.dim() != 2 or not p.is_contiguous() or kr > self.LOWRANK_MAX:
            raise RuntimeError("FusedNAdam: low-rank update needs a contiguous 2-D parameter and rank <= 64")
        row0, rows = shard if shard is not None else (0, p.shape[0])
        if row0 < 0 or rows < 1 or row0 + rows > p.shape[0]:
            raise RuntimeError("FusedNAdam: row shard outside the parameter")
        if kr and (fa.shape[1] != rows or fb.shape[1] != p.shape[1] or fb.shape[0] != kr
                   or fa.stride(1) != 1 or fb.stride(1) != 1):
            raise RuntimeError("FusedNAdam: low-rank factors do not match the parameter")
        _lib.require_gpu(p, "FusedNAdam.step")
        st = self._state_for(p, shard)
        st["step"] += 1
        cg, cm, bc2, st["mu_product"] = nadam_scalars(st["step"], st["mu_product"], group["lr"], b1, b2,
                                                      group["momentum_decay"])
        if dh is not None:
            slab, U, row_tiles = dh
            nslab = -(-(-(-rows // 32)) // row_tiles)
            if (shard is not None or not kr or U > kr or tuple(slab.shape) != (nslab, U, p.shape[1]) or not slab.is_contiguous()
                    or slab.dtype != torch.float32):
                raise RuntimeError("FusedNAdam: the fused product needs the whole parameter, U <= rank and a contiguous fp32 slab "
                                   f"({nslab}, {U}, {p.shape[1]})")
            check(self._lib.tl_nadam_lowrank_dh(p.data_ptr(), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(fa), ptr(fb), kr, rows,
                                                p.shape[1], fa.stride(0), fb.stride(0), cg, cm, b1, b2, bc2, group["eps"],
                                                group["weight_decay"], grad_scale, ptr(slab), U, row_tiles, stream),
                  "tl_nadam_lowrank_dh")
            return
        check(self._lib.tl_nadam_lowrank(p.data_ptr() + 4 * row0 * p.shape[1], ptr(st["exp_avg"]),
                                         ptr(st["exp_avg_sq"]), ptr(fa), ptr(fb),
                                         kr, rows, p.shape[1], fa.stride(0) if kr else rows,
                                         fb.stride(0) if kr else p.shape[1], cg, cm, b1, b2, bc2,
                                         group["eps"], group["weight_decay"], grad_scale, stream),
              "tl_nadam_lowrank")

    @torch.no_grad()
    def step(self, closure=None, grads: Optional[Dict[torch.nn.Parameter, torch.Tensor]] = None,
             grad_scale: float = 1.0, lowrank: Optional[Dict[torch.nn.Parameter, tuple]] = None, skip=None):
        """``grads`` optionally maps parameter -> gradient tensor (fused trainer path, no ``.grad``).
        ``lowrank`` maps a 2-D parameter (rows, cols) to factors ``(fa (k, rows), fb (k, cols))`` of its
        gradient ``fa^T . fb``, k <= LOWRANK_MAX (or ``(None, None)`` for a zero gradient): the update is
        applied without materialising the gradient.  ``(fa (k, n), fb, row0, n)`` updates only rows
        [row0, row0 + n) of the parameter (a data-parallel rank that owns a row shard of it).
        ``skip``: parameters already updated this step through ``step_lowrank``."""
        loss = closure() if closure is not None else None
        stream = torch.cuda.current_stream().cuda_stream
        dense = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if lowrank is not None and p in lowrank:
                    self._step_lowrank(p, group, lowrank[p], grad_scale, stream)
                    continue
                if skip is not None and p in skip:
                    continue
                g = grads.get(p) if grads is not None else p.grad
                if g is None:
                    continue
                _lib.require_gpu(p, "FusedNAdam.step")
                if not (p.is_contiguous() and g.is_contiguous()):
                    raise RuntimeError("FusedNAdam needs contiguous parameters and gradients")
                st = self._state_for(p)
                st["step"] += 1
                dense.setdefault((st["step"], st["mu_product"]), []).append((p, g, st))
            # tensors at the same point of the schedule (normally all of them) share one launch
            for (step_no, mu_prod), items in dense.items():
                cg, cm, bc2, mu_prod = nadam_scalars(step_no, mu_prod, group["lr"], b1, b2, group["momentum_decay"])
                for _, _, st in items:
                    st["mu_product"] = mu_prod
                if len(items) == 1 or not self.multi_tensor:
                    for p, g, st in items:
                        check(self._lib.tl_nadam(ptr(p), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), p.numel(), cg,
                                                 cm, b1, b2, bc2, group["eps"], group["weight_decay"], grad_scale, stream),
                              "tl_nadam")
                    continue
                table, blocks = self._table(items)
                check(self._lib.tl_nadam_multi(ptr(table), len(items), blocks, cg, cm, b1, b2, bc2, group["eps"],
                                               group["weight_decay"], grad_scale, stream), "tl_nadam_multi")
            dense.clear()
        return loss

    # ---- HIP-graph support: the step's scalars live in device memory, the host only keeps the schedule ----
    @torch.no_grad()
    def step_graph(self, grads: Dict[torch.nn.Parameter, torch.Tensor], scalars_dev: torch.Tensor,
                   grad_scale: float = 1.0) -> None:
        """Enqueue the update of every parameter in ``grads`` with coef_grad / coef_mom / bias_corr2 read from
        ``scalars_dev[0..2]`` (``tl_nadam_multi_dev``): the launch can be captured in a HIP graph.  Does NOT advance the
        schedule - call ``advance_scalars()`` once per (re)play and copy its result into ``scalars_dev`` first."""
        stream = torch.cuda.current_stream().cuda_stream
        for group in self.param_groups:
            b1, b2 = group["betas"]
            items = []
            for p in group["params"]:
                g = grads.get(p)
                if g is None:
                    continue
                _lib.require_gpu(p, "FusedNAdam.step_graph")
                if not (p.is_contiguous() and g.is_contiguous()):
                    raise RuntimeError("FusedNAdam needs contiguous parameters and gradients")
                items.append((p, g, self._state_for(p)))
            if not items:
                continue
            table, blocks = self._table(items)
            check(self._lib.tl_nadam_multi_dev(ptr(table), len(items), blocks, ptr(scalars_dev), b1, b2, group["eps"],
                                               group["weight_decay"], grad_scale, stream), "tl_nadam_multi_dev")

    def advance_scalars(self, params) -> tuple:
        """Advance the schedule of ``params`` (all at the same point of it) by one step on the host and return
        (coef_grad, coef_mom, bias_corr2) of that step - what ``step`` would have passed by value."""
        out = None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p not in params:
                    continue
                st = self._state_for(p)
                st["step"] += 1
                cg, cm, bc2, st["mu_product"] = nadam_scalars(st["step"], st["mu_product"], group["lr"], b1, b2,
                                                              group["momentum_decay"])
                if out is not None and out != (cg, cm, bc2):
                    raise RuntimeError("FusedNAdam.advance_scalars: parameters at different points of the schedule")
                out = (cg, cm, bc2)
        return out

    def _table(self, items):
        """Device table of ``tl_nadam_entry`` rows for ``items`` = [(param, grad, state)], cached while the
        pointers stay the same (the fused trainer keeps its gradient buffers)."""
        key = tuple((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                    for p, g, st in items)
        hit = self._tables.get(key)
        if hit is None:
            if len(self._tables) > 8:
                self._tables.clear()
            chunk = self._lib.tl_nadam_multi_chunk()
            rows, block0 = [], 0
            for pp, gp, mp, vp, n in key:
                if (pp | gp | mp | vp) & 15:
                    raise RuntimeError("FusedNAdam: parameter / gradient storage must be 16-byte aligned")
                rows.append((pp, gp, mp, vp, n, block0))
                block0 += (n + chunk - 1) // chunk
            dev = items[0][0].device
            hit = self._tables[key] = (torch.tensor(rows, dtype=torch.int64).to(dev), block0)
        return hit
