"""``SynthesisTrainer`` on MI355X (mirror of reference models/synthesis_trainer.py:46-302).

Same constructor keywords, ``train(loader, epochs, verbose) -> [(loss, mcd)]`` and
``evaluate(loader) -> (mcd, recon, origin)``.  What changes is where the work happens:

* the label dynamics are gathered on the device (``tl_tone_dynamics``) instead of the
  ``.cpu().numpy()`` -> Python loop -> ``torch.Tensor`` round trip (:212-218);
* for the models of this package the step is fused: forward, L1 (+ MCD statistics), backward and
  NAdam run as HIP kernels on one stream with no autograd graph and no host sync per step
  (the reference syncs three times per step, :214-215,228,229); loss / MCD are accumulated on
  the device and read once per epoch;
* under ``torch.distributed`` (one process per GPU) every rank takes its row shard of each
  global batch and gradients are reduced over RCCL (parallel.py).

Reference quirks kept on purpose: training targets are truncated to integers (:222), evaluation
targets are not (:290); the classifiers are never updated (their parameters are not in the
optimizer, :131-137) - ``train_classifiers`` only toggles ``.train()``.
"""
from __future__ import annotations

import os

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from .. import _kernels, _lib, parallel
from .._lib import check, ptr
from ..optim import FusedNAdam
from .classifier import ClassifierModel
from .synthesis_models import SynthesisModel


def compute_mcd(true_mcc: torch.Tensor, pred_mcc: torch.Tensor) -> float:
    """Mean over the batch of 10/ln10 * sqrt(2 * sum_k (t - p)^2) (reference :14-43), computed by
    the ``tl_l1_mcd`` kernel."""
    t = true_mcc.float().contiguous()
    p = pred_mcc.detach().float().contiguous()
    _lib.require_gpu(p, "compute_mcd")
    stats = torch.zeros(4, dtype=torch.float32, device=p.device)
    B, D = p.shape
    check(_lib.load().tl_l1_mcd(ptr(p), ptr(t.to(p.device)), None, ptr(stats), B, D, D, 0, 1.0,
                                torch.cuda.current_stream().cuda_stream), "tl_l1_mcd")
    return float(stats[3].item())


class SynthesisTrainer:
    def __init__(self, synthesize_model: SynthesisModel, tone_model: ClassifierModel, syllable_model: ClassifierModel,
                 tone_dynamic_mapping: Dict[str, List[int]], device: torch.device = torch.device("cpu"),
                 learning_rate: float = 0.0005, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-08,
                 schedule_decay: float = 0.004, verbose: bool = True, train_classifiers: bool = False) -> None:
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SynthesisTrainer (MI355X build) needs a 'cuda' device: the synthesis path runs "
                               "hand-written HIP kernels and has no CPU fallback")
        self.lib = _lib.load()
        self.train_classifiers = train_classifiers
        self.tone_dynamic_mapping = tone_dynamic_mapping
        self.model = synthesize_model.to(self.device)
        if verbose:
            print(f"Number of trainable parameters in the synthesis model: {self.model.get_nparams():,}")
        self.optimizer = FusedNAdam(self.model.parameters(), lr=learning_rate, betas=(beta_1, beta_2), eps=epsilon,
                                    weight_decay=schedule_decay)
        self.criterion = nn.L1Loss()          # generic (autograd) path only
        self.tone_model = tone_model.to(self.device)
        self.syllable_model = syllable_model.to(self.device)
        if not train_classifiers:
            self.tone_model.eval()
            self.syllable_model.eval()
        elif verbose:
            for nm, m in (("tone", self.tone_model), ("syllable", self.syllable_model)):
                n = sum(p.numel() for p in m.parameters() if p.requires_grad)
                print(f"Number of trainable parameters in the {nm} model: {n:,}")
        # tone dynamics table: row t = mapping[str(t)]
        keys = sorted(int(k) for k in tone_dynamic_mapping)
        lens = {len(v) for v in tone_dynamic_mapping.values()}
        if len(lens) != 1:
            raise ValueError("every entry of tone_dynamic_mapping must have the same length")
        self._L = lens.pop()
        self._n_rows = (max(keys) + 1) if keys else 0
        table = torch.full((max(self._n_rows, 1), self._L), float("nan"))
        for k in keys:
            if k >= 0:
                table[k] = torch.tensor(tone_dynamic_mapping[str(k)], dtype=torch.float32)
        self._table_host = table
        n_cls = getattr(self.tone_model, "n_classes", None)
        self._need_check = n_cls is None or any(str(t) not in tone_dynamic_mapping for t in range(n_cls))
        self._table = table.to(self.device)
        # every (tone, syllable) class pair as a label sequence: the synthesis model's LSTM runs on these rows and
        # the batch gathers from them by id = tone * n_syllables + syllable (no torch.unique, no host sync per step)
        n_syl = getattr(self.syllable_model, "n_classes", None)
        self._pair_table = None
        # Only the tone classes the classifier can predict (rows 0 .. n_classes - 1, all present when _need_check is
        # False) enter the table: a mapping with extra keys above them (say "5" beside a 4-class model) would otherwise
        # contribute NaN rows for the keys in between - no batch element gathers them, but the LSTM unrolls over every
        # table row and 0 * NaN in its backward pass would poison all of W_hh, W_ih and the biases.
        if not self._need_check and n_syl is not None and n_syl >= 1 and n_cls * n_syl <= 16:
            syl_col = torch.arange(n_syl, dtype=torch.float32).view(1, n_syl, 1).expand(n_cls, n_syl, self._L)
            dyn = table[:n_cls].view(n_cls, 1, self._L).expand(n_cls, n_syl, self._L)
            assert not torch.isnan(dyn).any()
            self._pair_table = torch.stack([syl_col, dyn], dim=2).reshape(n_cls * n_syl, 2, self._L).contiguous() \
                .to(self.device)
            self._n_syl = int(n_syl)
        self._pair_ids = None
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._stats = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._grads = None
        self.rank, self.world = parallel.world()
        self.dp = parallel.active()
        self._whh_dirty = False
        self._graph_enabled = _kernels.get("graph") != "0"
        self._graphs, self._g_scal = {}, None
        eng = getattr(self.model, "_engine", None)
        if (self.dp and eng is not None and hasattr(eng, "lstm_shard") and self._pair_table is not None
                and _kernels.get("lstm_shard") != "0"):
            eng.lstm_shard = (self.rank, self.world)        # row-sharded label LSTM (parallel.py docstring)
            # Between steps each rank holds current values only for its own rows of weight_hh_l0.  `train` / `evaluate`
            # re-assemble it, but `train_step` is public: anything that reads the whole model (state_dict -> checkpoints)
            # first runs the collective, on every rank (a collective: all ranks must call state_dict together).  The
            # NAdam moments of that parameter stay per-rank shards: optimizer state under data parallelism is per rank.
            self.model.register_state_dict_pre_hook(self._state_dict_guard)
            self._guard_hooked = True

    def _state_dict_guard(self, module, prefix, keep_vars) -> None:
        """``state_dict()`` of the model while its label LSTM is row-sharded and out of date.  Re-assembling the weight is a
        collective: hidden inside ``state_dict`` it would deadlock the usual ``if rank == 0: torch.save(model.state_dict())``
        (and fire again for every nested / partial call).  So this raises; ``train`` / ``evaluate`` synchronise themselves,
        after a bare ``train_step`` call ``sync_parameters()`` (or ``model_state_dict()``) on EVERY rank first."""
        if self._whh_dirty:
            raise RuntimeError(
                "SynthesisTrainer: label_lstm.weight_hh_l0 is row-sharded over the data-parallel ranks and this rank only "
                "holds current values for its own rows.  Call trainer.sync_parameters() (or trainer.model_state_dict()) on "
                "every rank before model.state_dict() - it is a collective and is therefore not run implicitly here.")

    # ------------------------------------------------------------------ helpers
    def _labels(self, inputs_tone, inputs_syllable) -> torch.Tensor:
        """argmax of both classifiers + device gather of the dynamics (reference :207-218)."""
        st, ss = self.tone_model(inputs_tone), self.syllable_model(inputs_syllable)
        B = st.shape[0]
        labels = torch.empty(B, 2, self._L, dtype=torch.float32, device=self.device)
        self._pair_ids = None
        if st.dim() == 2 and ss.dim() == 2 and st.dtype == torch.float32 and ss.dtype == torch.float32 and st.is_cuda and ss.is_cuda:
            # arg-max of both score matrices, the dynamics gather and the pair id in one launch
            st, ss = st.detach().contiguous(), ss.detach().contiguous()
            tone = torch.empty(B, dtype=torch.int64, device=self.device)
            syl = torch.empty(B, dtype=torch.int64, device=self.device)
            pair = torch.empty(B, dtype=torch.int32, device=self.device) if self._pair_table is not None else None
            check(self.lib.tl_labels_from_scores(ptr(st), ptr(ss), ptr(self._table), ptr(labels), ptr(tone), ptr(syl),
                                                 ptr(pair) if pair is not None else None, ptr(self._err), B, st.shape[1],
                                                 ss.shape[1], self._n_rows, self._n_syl if pair is not None else 1, self._L,
                                                 torch.cuda.current_stream().cuda_stream), "tl_labels_from_scores")
            if pair is not None:                  # ids are valid for exactly this label tensor
                self._pair_ids = (pair, labels)
        else:
            tone = torch.argmax(st, dim=1).contiguous()
            syl = torch.argmax(ss, dim=1).contiguous()
            if self._pair_table is not None:          # ids are valid for exactly this label tensor
                self._pair_ids = ((tone * self._n_syl + syl).to(torch.int32), labels)
            check(self.lib.tl_tone_dynamics(ptr(tone), ptr(syl), ptr(self._table), ptr(labels), ptr(self._err), B,
                                            self._n_rows, self._L, torch.cuda.current_stream().cuda_stream),
                  "tl_tone_dynamics")
        if self._need_check:
            # a predicted tone may be absent from the mapping: keep the reference's immediate error.  Under
            # data parallelism the decision is taken by all ranks together (a rank that raised alone would
            # leave its peers blocked in the next collective).
            bad = [int(t) for t in tone.tolist() if str(int(t)) not in self.tone_dynamic_mapping]
            flag = torch.tensor([bad[0] + 1 if bad else 0], dtype=torch.int64, device=self.device)
            if self.dp:
                parallel.all_reduce_(flag, op=torch.distributed.ReduceOp.MAX)
            worst = int(flag.item())
            if worst:
                raise ValueError(f"Tone {worst - 1} not found in tone_dynamic_mapping."
                                 f"Available tones in mapping: {list(self.tone_dynamic_mapping.keys())}")
        return labels

    def _shard(self, *tensors):
        """Row shard of a global batch for this rank.  Sets ``self._row0`` (first global row of the shard)
        and ``self._weight`` = (rows this rank contributes) / (rows of the global batch): the loss is a
        mean over the GLOBAL batch, so a rank's gradient enters the all-reduce sum with that weight
        (uneven shards of a ragged last batch included).  A batch with fewer rows than ranks leaves
        some ranks without rows: they recompute row ``rank % n`` with weight 0, so every rank still
        takes part in every collective - decided identically everywhere from the global row count."""
        n = tensors[0].shape[0]
        if self.world == 1:
            self._row0, self._weight = 0, 1.0
            return tensors
        if n >= self.world:
            sl = parallel.shard_rows(n, self.rank, self.world)
            self._row0, self._weight = sl.start, (sl.stop - sl.start) / n
        elif self.rank < n:
            sl = slice(self.rank, self.rank + 1)
            self._row0, self._weight = self.rank, 1.0 / n
        else:
            r = self.rank % n
            sl = slice(r, r + 1)
            self._row0, self._weight = r, 0.0
        return tuple(t[sl] for t in tensors)

    def _loss_stats(self, out, targets, dout, ldd, trunc: int, stats=None) -> None:
        """``tl_l1_mcd``: loss gradient (scaled by this rank's weight in the global mean) and the
        L1 / MCD statistics.  Under data parallelism the per-rank statistics are accumulated with the
        same weight so that their sum over ranks is the statistic of the global batch."""
        stats = self._stats if stats is None else stats
        B, D = out.shape
        w = getattr(self, "_weight", 1.0)
        if not self.dp:
            check(self.lib.tl_l1_mcd(ptr(out), ptr(targets), ptr(dout), ptr(stats), B, D, ldd, trunc, 1.0,
                                     torch.cuda.current_stream().cuda_stream), "tl_l1_mcd")
            return
        tmp = torch.zeros(4, dtype=torch.float32, device=out.device)
        check(self.lib.tl_l1_mcd(ptr(out), ptr(targets), ptr(dout), ptr(tmp), B, D, ldd, trunc, float(w),
                                 torch.cuda.current_stream().cuda_stream), "tl_l1_mcd")
        stats[:2] += w * tmp[:2]
        stats[2:] = w * tmp[2:]

    def _timed(self, fn):
        """Wrap an exchange-step call with HIP events when ``self.exchange_events`` is a list (bench.py)."""
        events = getattr(self, "exchange_events", None)
        if events is None:
            return fn

        def run(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            events.append((e0, e1))
            return r
        return run

    def _fused_step(self, inputs_non, inputs_label, targets, graph: bool = False) -> None:
        model = self.model
        eng = model._engine
        names = model._pnames
        params = dict(model.named_parameters())
        prm = {k: params[k].detach() for k in names}
        skip = getattr(eng, "lowrank_param", None)
        if self._grads is None:
            # the low-rank parameter's gradient (5.4 GB at the north-star shape) is only materialised
            # if its rank exceeds what the fused optimiser kernel takes (the engine then allocates it)
            if self.dp and hasattr(eng, "grad_order"):
                # data parallel: ONE flat buffer in the order the backward pass finishes the gradients - a bucket of the
                # exchange step is a contiguous slice (no torch.cat in front of the collective, no copy back behind it)
                sharded = getattr(eng, "lstm_shard", None) is not None and self._pair_table is not None
                local_only = ("label_lstm.weight_ih_l0", "label_lstm.bias_ih_l0", "label_lstm.bias_hh_l0") if sharded else ()
                self._flat = parallel.FlatGrads({k: v.shape for k, v in prm.items() if k != skip}, eng.grad_order(),
                                                prm[names[0]].device, local=local_only)
                self._grads = self._flat.views
            else:
                self._grads = {k: torch.empty_like(v) for k, v in prm.items() if k != skip}
        prm.update(model._engine_buffers())
        kw = {}
        ids = getattr(self, "_pair_ids", None)
        if ids is not None and ids[1] is inputs_label and getattr(eng, "lowrank_param", None) is not None:
            kw = dict(label_ids=ids[0], label_table=self._pair_table)             # CNN engine: skip torch.unique
        # (HIP-graph capture: the engine reads the dropout seed from device memory, the host seed counter is advanced by
        # the caller once per replay)
        out = eng.forward(prm, inputs_non, inputs_label, training=model.training, save=True,
                          seed=0 if graph else model._next_seed(), row0=getattr(self, "_row0", 0), **kw)
        B, D = out.shape
        self._last_out = out             # the step's outputs (pre-update), for callers that track them (tests)
        # (tl_l1_mcd writes all D columns of a row: the zero fill is only for padding columns)
        dout = (torch.empty if eng.ldd == D else torch.zeros)(B, eng.ldd, dtype=torch.float32, device=out.device)
        self._loss_stats(out, targets, dout, eng.ldd, 1)
        gather = self._timed(parallel.gather_lowrank) if self.dp else None
        reduce_rows = self._timed(parallel.all_reduce_) if self.dp else None
        scale = 1.0          # the 1/N of the global mean is already in dout (weight of this rank's rows)
        early = []

        def on_factors(dh=None):
            # single process: the W_hh update (33 GB of HBM traffic) is issued as soon as its gradient factors exist, on the
            # stream the LSTM backward runs on - beside the convolution backward instead of behind it.  ``dh``: the engine
            # asks for the last BPTT product out of the same pass (they exist one step early: h_0 = 0)
            factors = getattr(eng, "whh_factors", None)
            if factors is not None and not self.dp:
                self.optimizer.step_lowrank({params[skip]: factors}, grad_scale=scale, dh=dh)
                early.append(params[skip])
        on_factors.fuse_dh = not self.dp
        flat = getattr(self, "_flat", None) if self.dp else None
        pending = []

        def on_grad_ready(group):
            # Data parallel: the gradients of the output layer (63 MB of the 72 MB that are exchanged at the north-star
            # shape) are final before the convolution backward starts - their all-reduce is started here and runs beside
            # that backward pass ("launched as soon as each grad is final"); the optimiser waits for it below
            if flat is not None and group == "output_layer":
                pending.append(self._timed(parallel.all_reduce_async)(flat.span(["output_layer.weight", "output_layer.bias"])))
        bkw = dict(on_grad_ready=on_grad_ready) if flat is not None else {}
        eng.backward(prm, dout, self._grads, gather_whh=gather, whh_factors=skip is not None, reduce_rows=reduce_rows,
                     on_factors=on_factors if skip is not None else None, **bkw)
        if self.dp:
            sharded = getattr(eng, "_sh", None) is not None
            # with the row-sharded LSTM its dgates - hence the W_ih / bias gradients - already belong to the global
            # batch on every rank: they stay out of the all-reduce
            local_only = ("label_lstm.weight_ih_l0", "label_lstm.bias_ih_l0", "label_lstm.bias_hh_l0") if sharded else ()
            if flat is not None and tuple(local_only) == tuple(flat.local):
                rest = [k for k in flat.offsets if k not in ("output_layer.weight", "output_layer.bias") and k not in local_only]
                if not pending:                              # (an engine that never reported the early group)
                    rest = [k for k in flat.offsets if k not in local_only]
                if rest:
                    pending.append(self._timed(parallel.all_reduce_async)(flat.span(rest)))
                wait_events = getattr(self, "exchange_wait_events", None)
                for h in pending:
                    h.wait(wait_events)
            else:
                # The row-sharded LSTM fell back for this step (its shard decision is per forward: (L - 1) U > 64, a gate-row
                # count the world size does not divide, no label table) or started to shard after the flat layout was fixed:
                # the flat span no longer matches what must be reduced.  The early bucket (output layer) is already in
                # flight: wait for it and reduce every OTHER gradient exactly once.
                wait_events = getattr(self, "exchange_wait_events", None)
                reduced = set()
                for h in pending:
                    h.wait(wait_events)
                    reduced.update(("output_layer.weight", "output_layer.bias"))
                self._timed(parallel.allreduce_bucketed)([g for k, g in self._grads.items()
                                                          if k != skip and k not in local_only and k not in reduced])
            self._whh_dirty = self._whh_dirty or sharded
        factors = getattr(eng, "whh_factors", None)
        if graph:            # scalars of the step from device memory (advanced by the caller per replay)
            self.optimizer.step_graph({params[k]: self._grads[k] for k in names}, self._g_scal, grad_scale=scale)
        elif early:            # W_hh is already updated
            self.optimizer.step(grads={params[k]: self._grads[k] for k in names if k != skip}, grad_scale=scale, skip=set(early))
        elif factors is not None:          # the optimiser forms that gradient from its factors on the fly
            self.optimizer.step(grads={params[k]: self._grads[k] for k in names if k != skip}, grad_scale=scale,
                                lowrank={params[skip]: factors})
        else:
            self.optimizer.step(grads={params[k]: self._grads[k] for k in names}, grad_scale=scale)

    def _generic_step(self, inputs_non, inputs_label, targets) -> None:
        """Any other ``SynthesisModel`` subclass: torch autograd for the model, fused NAdam."""
        self.optimizer.zero_grad()
        outputs = self.model(inputs_non, inputs_label)
        tgt = targets.long()
        loss = self.criterion(outputs, tgt)
        if self.dp:
            (loss * getattr(self, "_weight", 1.0)).backward()
            for p in self.model.parameters():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            self._timed(parallel.allreduce_bucketed)([p.grad for p in self.model.parameters()])
        else:
            loss.backward()
        self.optimizer.step(grad_scale=1.0)
        out = outputs.detach().float().contiguous()
        self._loss_stats(out, targets.float().contiguous(), None, out.shape[1], 1)

    def sync_parameters(self) -> None:
        """Re-assemble parameters that data-parallel ranks update shard-wise (the row-sharded ``weight_hh_l0``):
        afterwards every rank holds the full, current model (``state_dict``, ``evaluate``)."""
        if not self._whh_dirty:
            return
        eng = self.model._engine
        p = dict(self.model.named_parameters())[eng.lowrank_param]
        rows = p.shape[0] // self.world
        parallel.all_gather_param_rows_(p.data, self.rank * rows, rows)
        self._whh_dirty = False

    def set_lstm_shard(self, on: bool) -> bool:
        """Switch the gate-row sharding of the label LSTM on or off BETWEEN steps (data parallel only; a collective: every
        rank must call it with the same value).  The weight is re-assembled first; the NAdam state of ``weight_hh_l0``
        follows the mode - its moments are cut to this rank's rows (whole -> shard: every rank holds the same full
        moments) or all-gathered (shard -> whole), its step count and mu product stay - so the parameter keeps the
        single-process trajectory across a switch (bench.py ``--lstm-shard auto`` switches during warm-up).  The flat gradient
        layout is rebuilt.  Returns whether the LSTM will run sharded (False where the model / label table does not allow it)."""
        eng = getattr(self.model, "_engine", None)
        if not self.dp or eng is None or not hasattr(eng, "lstm_shard"):
            return False
        self.sync_parameters()
        want = (self.rank, self.world) if (on and self._pair_table is not None and self.world > 1) else None
        if want != eng.lstm_shard:
            eng.lstm_shard = want
            p = dict(self.model.named_parameters())[eng.lowrank_param]
            st = self.optimizer.state.get(p)
            if st:
                rows = p.shape[0] // self.world
                r0 = self.rank * rows
                for key in ("exp_avg", "exp_avg_sq"):
                    if p.is_cuda:
                        torch.cuda.empty_cache()              # (a fresh allocation per moment tensor: optim.FusedNAdam._state_for)
                    if want is not None:                      # whole -> this rank's rows
                        st[key] = st[key][r0:r0 + rows].clone()
                    else:                                     # row shards -> the whole matrix on every rank
                        full = torch.zeros_like(p, memory_format=torch.preserve_format)
                        full[r0:r0 + rows].copy_(st[key])
                        st[key] = parallel.all_gather_param_rows_(full, r0, rows)
                st["shard_rows"] = (r0, rows) if want is not None else None
            self._grads, self._flat = None, None
            if want is not None and not getattr(self, "_guard_hooked", False):
                self.model.register_state_dict_pre_hook(self._state_dict_guard)
                self._guard_hooked = True
        return eng.lstm_shard is not None

    def model_state_dict(self):
        """``model.state_dict()`` with the shard-wise updated parameters re-assembled first.  Collective under data
        parallelism (every rank must call it).  The NAdam moments of a row-sharded parameter stay per rank and
        shard-shaped: an optimizer checkpoint is only resumable with the same world size."""
        self.sync_parameters()
        return self.model.state_dict()

    def train_step(self, inputs_non, inputs_syllable, inputs_tone, targets) -> None:
        """One body of the batch loop (reference :201-229).  Loss / MCD go to ``self._stats``.

        Under data parallelism with the row-sharded label LSTM a rank holds current values only for its own gate rows of
        ``label_lstm.weight_hh_l0`` between steps: ``train`` and ``evaluate`` re-assemble them (``sync_parameters``); a caller
        that drives ``train_step`` itself must call ``sync_parameters()`` (or ``model_state_dict()``) on every rank before it
        reads, evaluates or saves the model."""
        dev = self.device
        inputs_non = inputs_non.to(dev, non_blocking=True)
        inputs_syllable = inputs_syllable.to(dev, non_blocking=True)
        inputs_tone = inputs_tone.to(dev, non_blocking=True)
        targets = targets.to(dev, non_blocking=True).float().contiguous()
        inputs_non, inputs_syllable, inputs_tone, targets = self._shard(inputs_non, inputs_syllable, inputs_tone,
                                                                        targets)
        if self._graph_ok() and self._graph_step(inputs_non, inputs_syllable, inputs_tone, targets):
            return
        with torch.no_grad():
            inputs_label = self._labels(inputs_tone, inputs_syllable)
        if getattr(self.model, "_engine", None) is not None and hasattr(self.model._engine, "backward"):
            self._fused_step(inputs_non.float().contiguous(), inputs_label, targets.contiguous())
        else:
            self._generic_step(inputs_non, inputs_label, targets)

    # ------------------------------------------------------------------ HIP-graph replay of a launch-bound step
    def _graph_ok(self) -> bool:
        """The step of a small model (SynthesisLite: ~58 launches of 4-20 us) is bound by launch overhead, not by the GPU:
        after three eager steps at a batch shape it is captured once into a HIP graph and replayed.  What changes from step
        to step lives in device memory - the NAdam coefficients (``tl_nadam_multi_dev``) and the dropout seed
        (``tl_lite_cat_dev`` / ``tl_lite_uncat_dev``) - and is refreshed by two small copies before each replay.  Not under
        data parallelism (collectives), not with classifiers in train mode (their dropout seed is a launch argument), not
        for the big model (GPU-bound; its W_hh update takes launch-argument scalars).  TONAL_GRAPH=0 turns it off."""
        eng = getattr(self.model, "_engine", None)
        return (self._graph_enabled and eng is not None and hasattr(eng, "seed_dev") and not self.dp
                and not self.train_classifiers and not self._need_check and self.model.training)

    def _graph_fingerprint(self) -> tuple:
        """Everything a captured step has frozen into its kernel arguments besides the batch: storage of the parameters,
        of their gradients and NAdam moments, the optimiser's hyper-parameters, the dropout rate and the classifiers'
        weights (their packed copies are rebuilt when a weight's version changes).  A replay is only valid while this tuple
        is what it was at capture: ``optimizer.load_state_dict`` (new moment tensors), an edited ``param_groups`` entry, a
        re-loaded classifier or a re-assigned ``.data`` would otherwise leave the graph reading freed or stale buffers."""
        opt = self.optimizer
        fp = []
        name_of = {id(p): k for k, p in self.model.named_parameters()}       # gradient storage is keyed by parameter NAME
        for group in opt.param_groups:
            fp.append((tuple(group["betas"]), group["eps"], group["weight_decay"], group["momentum_decay"]))
            for p in group["params"]:
                st = opt.state.get(p) or {}
                g = (self._grads or {}).get(name_of.get(id(p)))
                fp.append((p.data_ptr(), None if g is None else g.data_ptr(),
                           None if "exp_avg" not in st else st["exp_avg"].data_ptr(),
                           None if "exp_avg_sq" not in st else st["exp_avg_sq"].data_ptr()))
        eng = getattr(self.model, "_engine", None)
        fp.append((getattr(eng, "p_drop", None), getattr(self.model, "training", None)))
        for clf in (self.tone_model, self.syllable_model):
            fp.append(tuple((q.data_ptr(), q._version) for q in clf.parameters()) + (clf.training,))
        return tuple(fp)

    def _graph_step(self, inputs_non, inputs_syllable, inputs_tone, targets) -> bool:
        key = (tuple(inputs_non.shape), tuple(inputs_syllable.shape), tuple(inputs_tone.shape), tuple(targets.shape),
               inputs_non.dtype, inputs_syllable.dtype, inputs_tone.dtype)
        st = self._graphs.get(key)
        if st is None:
            st = self._graphs[key] = {"warm": 0, "graph": None}
        if st["graph"] is not None and st["fingerprint"] != self._graph_fingerprint():
            # something the capture had frozen was replaced: drop the graph (and the NAdam entry table it pinned), warm
            # up eagerly again and recapture - never replay onto freed or stale buffers
            st.update(graph=None, warm=0, params=None, keep=None, fingerprint=None)
        if st["graph"] is None:
            if st["warm"] < 3 or len([v for v in self._graphs.values() if v["graph"] is not None]) >= 2:
                st["warm"] += 1
                return False                      # eager warm-up steps (and at most two captured shapes)
            model, eng = self.model, self.model._engine
            params = [p for _, p in model.named_parameters()]
            st["in"] = tuple(torch.empty_like(t) for t in (inputs_non, inputs_syllable, inputs_tone, targets))
            if self._g_scal is None:
                self._g_scal = torch.zeros(4, dtype=torch.float32, device=self.device)
                self._g_seed = torch.zeros(1, dtype=torch.int64, device=self.device)
            eng.seed_dev = self._g_seed
            for d, t in zip(st["in"], (inputs_non, inputs_syllable, inputs_tone, targets)):
                d.copy_(t)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    with torch.no_grad():
                        lab = self._labels(st["in"][2], st["in"][1])
                    self._fused_step(st["in"][0].float().contiguous(), lab, st["in"][3].contiguous(), graph=True)
            except Exception as e:      # noqa: BLE001 - capture refused (e.g. an op that synchronises): stay eager for good
                import warnings
                warnings.warn(f"SynthesisTrainer: HIP-graph capture of the train step failed ({e!r}); continuing eagerly")
                eng.seed_dev = None
                self._graph_enabled = False
                return False            # nothing was executed during the failed capture: run this step eagerly
            eng.seed_dev = None                      # eager steps (other shapes) keep passing the seed by value
            st["graph"], st["params"] = g, set(params)
            # the graph holds raw pointers into the optimiser's entry tables: keep those tensors alive with it (the
            # optimiser's own cache is cleared when it grows), and remember what the capture froze
            st["keep"] = list(self.optimizer._tables.values())
            st["fingerprint"] = self._graph_fingerprint()
            staged = ()
        else:
            staged = tuple(zip(st["in"], (inputs_non, inputs_syllable, inputs_tone, targets)))
        cg, cm, bc2 = self.optimizer.advance_scalars(st["params"])
        # the values travel as launch arguments (a pinned host buffer would be overwritten by the next step before an
        # asynchronous copy of this one has read it: the host runs ahead of the stream); the batch goes into the graph's
        # static input buffers by the same launch
        if all(t.is_cuda and t.is_contiguous() and t.dtype == d.dtype and t.shape == d.shape for d, t in staged):
            import ctypes as C
            n = len(staged)
            src = (C.c_void_p * 4)(*[t.data_ptr() for _, t in staged])
            dst = (C.c_void_p * 4)(*[d.data_ptr() for d, _ in staged])
            nby = (C.c_int64 * 4)(*[t.numel() * t.element_size() for _, t in staged])
            check(self.lib.tl_stage_step(ptr(self._g_scal), ptr(self._g_seed), cg, cm, bc2, self.model._next_seed(), src, dst, nby, n,
                                         torch.cuda.current_stream().cuda_stream), "tl_stage_step")
        else:
            for d, t in staged:
                d.copy_(t, non_blocking=True)
            check(self.lib.tl_set_step_scalars(ptr(self._g_scal), ptr(self._g_seed), cg, cm, bc2, self.model._next_seed(),
                                               torch.cuda.current_stream().cuda_stream), "tl_set_step_scalars")
        st["graph"].replay()
        return True

    # ------------------------------------------------------------------ public API
    def train(self, train_loader: DataLoader, epochs: int, verbose: bool = True) -> List[Tuple[float, float]]:
        self.model.train()
        if self.train_classifiers:
            self.tone_model.train()
            self.syllable_model.train()
        history = []
        for epoch in range(epochs):
            self._stats.zero_()
            nb = 0
            for inputs_non, inputs_syllable, inputs_tone, targets in train_loader:
                self.train_step(inputs_non, inputs_syllable, inputs_tone, targets)
                nb += 1
            self.sync_parameters()
            stats = self._stats.clone()
            if self.dp:
                parallel.all_reduce_(stats)          # per-rank statistics carry their weight in the global mean
            s = stats.tolist()                                  # the one host sync of the epoch
            epoch_loss, mcd = s[0] / max(nb, 1), s[1] / max(nb, 1)
            history.append((epoch_loss, mcd))
            if verbose:
                print(f"Epoch {epoch+1}/{epochs}, Loss: {epoch_loss:.4f}, Mean MCD: {mcd:.4f}")
        return history

    def evaluate(self, test_loader: DataLoader):
        self.sync_parameters()
        self.model.eval()
        self.tone_model.eval()
        self.syllable_model.eval()
        recon, origin = [], []
        stats = torch.zeros(4, dtype=torch.float32, device=self.device)
        nb = 0
        with torch.no_grad():
            for inputs_non, inputs_syllable, inputs_tone, targets in test_loader:
                inputs_non = inputs_non.to(self.device)
                inputs_syllable = inputs_syllable.to(self.device)
                inputs_tone = inputs_tone.to(self.device)
                targets = targets.to(self.device).float().contiguous()
                inputs_label = self._labels(inputs_tone, inputs_syllable)
                outputs = self.model(inputs_non, inputs_label).float().contiguous()
                B, D = outputs.shape
                check(self.lib.tl_l1_mcd(ptr(outputs), ptr(targets), None, ptr(stats), B, D, D, 0, 1.0,
                                         torch.cuda.current_stream().cuda_stream), "tl_l1_mcd")
                recon.append(outputs.cpu())
                origin.append(targets.cpu())
                nb += 1
        mcd = float(stats[1].item()) / max(nb, 1)
        return mcd, torch.cat(recon, dim=0).numpy(), torch.cat(origin, dim=0).numpy()
