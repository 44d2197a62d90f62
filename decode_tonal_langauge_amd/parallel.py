"""Data-parallel glue: one process per GPU, ``torch.distributed`` over RCCL/xGMI (backend "nccl").

Windows are independent, so a global batch is sharded by rows (rank r takes rows
[r*B/N, (r+1)*B/N)) and the only exchange step is the gradient reduction:
 * every parameter except ``label_lstm.weight_hh_l0``: bucketed all-reduce (sum; the 1/N of the
   mean loss is folded into the NAdam kernel's ``grad_scale``);
 * ``weight_hh_l0`` (98.7 % of the parameters, 5.5 GB at the north-star shape) is never
   all-reduced: its gradient is ``dgates^T . h`` with at most (L-1)*U rows per rank, so the
   ranks all-gather those low-rank factors (a few MB) and each computes the full-batch gradient
   locally with the TN GEMM kernel;
 * when the trainer knows the distinct label sequences up front (the usual case), the label LSTM is
   instead SHARDED by gate rows ("preferred refinement"): rank r streams, and keeps
   NAdam moments for, only rows [r*4H/N, (r+1)*4H/N) of ``weight_hh_l0``; per LSTM step the ranks
   all-gather their slice of h W_hh^T (U x 4H/N floats each) in the forward pass and all-reduce the
   partial dgates W_hh (U x H) in the backward pass; the cell updates run redundantly on every rank, so
   the replicas stay in lockstep.  This divides the per-rank W_hh traffic (8 streams of 5.4 GB + 33 GB of
   NAdam per step at the north-star shape) by N.  ``SynthesisTrainer.sync_parameters()`` re-assembles
   the full weight on every rank (end of ``train``, before ``evaluate`` / ``state_dict``).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK/WORLD_SIZE/LOCAL_RANK (torchrun) if needed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    forced = os.environ.get("TONAL_DP_FORCE") == "1"       # single-rank rehearsal of the exchange step
    if (world > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("TONAL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend in ("nccl", "tl"):
            torch.cuda.set_device(local)
        # "tl": the RCCL handle of the C ABI (include/tonal_hip.h, tl_comm_*) carries every fp32 device buffer; the process
        # group (gloo) is the control plane - rendezvous, the 128-byte id, host-side objects
        dist.init_process_group(backend="gloo" if backend == "tl" else backend, rank=rank, world_size=world)
        if backend == "tl":
            tl_comm_init(rank, world)
    return rank, world, local


# ---- the C-ABI RCCL handle as the data path (TONAL_DIST_BACKEND=tl) --------------------------------------------------------
_TL = None          # (library, communicator, side stream for the asynchronous all-reduce)


def tl_comm_init(rank: int, nranks: int) -> None:
    """Create this process's communicator through ``tl_comm_init``: rank 0 draws the id, the process group carries its 128
    bytes to the others.  From here on the helpers below hand contiguous fp32 device tensors to ``tl_allreduce`` /
    ``tl_all_gather`` on torch's current stream; everything else keeps the process group."""
    global _TL
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    buf = (C.c_char * 128)()
    if rank == 0:
        _lib.check(lib.tl_comm_unique_id(buf), "tl_comm_unique_id")
    box = [bytes(buf.raw) if rank == 0 else None]
    if nranks > 1:
        dist.broadcast_object_list(box, src=0)
    handle = C.c_void_p()
    _lib.check(lib.tl_comm_init(C.byref(handle), rank, nranks, box[0]), "tl_comm_init")
    _TL = (lib, handle, torch.cuda.Stream())


def tl_comm_destroy() -> None:
    global _TL
    if _TL is not None:
        from . import _lib
        torch.cuda.synchronize()
        _lib.check(_TL[0].tl_comm_destroy(_TL[1]), "tl_comm_destroy")
        _TL = None


def tl_active() -> bool:
    """The C-ABI RCCL handle carries the fp32 device buffers of this process (``TONAL_DIST_BACKEND=tl``)."""
    return _TL is not None


def _tl_ok(*ts: torch.Tensor) -> bool:
    return _TL is not None and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts)


def _tl_allreduce(t: torch.Tensor, opcode: int) -> None:
    from . import _lib
    _lib.check(_TL[0].tl_allreduce(_TL[1], t.data_ptr(), t.data_ptr(), t.numel(), opcode, torch.cuda.current_stream().cuda_stream),
               "tl_allreduce")


def _tl_opcode(op):
    return {dist.ReduceOp.SUM: 0, dist.ReduceOp.MAX: 1, dist.ReduceOp.MIN: 2}.get(op)


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def active() -> bool:
    """True when the gradient exchange must run: a process group with more than one rank, or a
    single-rank group under ``TONAL_DP_FORCE=1`` (drives the real RCCL calls on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("TONAL_DP_FORCE") == "1"


def _staged() -> bool:
    """True when the process group cannot move device tensors itself (gloo): collectives are then
    staged through host memory.  Production runs use RCCL ("nccl"), which never takes this path;
    it exists so the data-parallel trainer logic can be exercised by two processes on one GPU."""
    return dist.get_backend() == "gloo"


def all_reduce_(t: torch.Tensor, op=None) -> torch.Tensor:
    """In-place all-reduce (sum by default) of a tensor on any device."""
    op = dist.ReduceOp.SUM if op is None else op
    if _tl_ok(t) and _tl_opcode(op) is not None and t.numel() > 0:
        _tl_allreduce(t, _tl_opcode(op))
    elif t.is_cuda and _staged():
        h = t.detach().cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def _all_gather_rows(out: torch.Tensor, t: torch.Tensor) -> None:
    if _tl_ok(out, t) and t.numel() > 0:
        from . import _lib
        _lib.check(_TL[0].tl_all_gather(_TL[1], t.data_ptr(), out.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream),
                   "tl_all_gather")
    elif t.is_cuda and _staged():
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(ho, t.detach().cpu().contiguous())
        out.copy_(ho)
    else:
        dist.all_gather_into_tensor(out, t)


def all_gather_blocks(out: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """``out`` (world, *t.shape) <- every rank's ``t`` (equal shapes)."""
    t2 = t.contiguous().view(-1, t.shape[-1])
    _all_gather_rows(out.view(-1, t.shape[-1]), t2)         # (world * rows, cols): the layout both backends accept
    return out


def all_gather_param_rows_(p: torch.Tensor, row0: int, rows: int) -> torch.Tensor:
    """In place: every rank contributes rows [row0, row0 + rows) of ``p`` (equal, rank-ordered row shards) and
    receives all others - re-assembles a parameter whose rows were updated shard-wise."""
    rank_, world_ = world()
    if row0 != rank_ * rows or rows * world_ != p.shape[0]:
        # (both in-place forms below put rank r's block at rows [r * rows, (r + 1) * rows) of the result)
        raise ValueError(f"all_gather_param_rows_: rank {rank_} of {world_} must own rows [{rank_ * rows}, {(rank_ + 1) * rows}) of "
                         f"{p.shape[0]}, got row0 = {row0}, rows = {rows}")
    mine = p[row0:row0 + rows]
    if _tl_ok(p, mine) and mine.numel() > 0:
        from . import _lib                       # in place: this rank's rows sit where its block of the result belongs
        _lib.check(_TL[0].tl_all_gather(_TL[1], mine.data_ptr(), p.data_ptr(), mine.numel(), torch.cuda.current_stream().cuda_stream),
                   "tl_all_gather")
    elif p.is_cuda and _staged():
        ho = torch.empty(p.shape, dtype=p.dtype)
        dist.all_gather_into_tensor(ho, mine.detach().cpu().contiguous())
        p.copy_(ho)
    else:
        try:
            dist.all_gather_into_tensor(p, mine)    # in place: the input is this rank's chunk of the output (NCCL's in-place form)
        except (RuntimeError, ValueError):
            # an argument check that refuses the aliased input fires on every rank alike, before anything is enqueued: take the
            # out-of-place form (one temporary of the parameter's size) everywhere
            tmp = torch.empty_like(p)
            dist.all_gather_into_tensor(tmp, mine.contiguous())
            p.copy_(tmp)
    return p


def shard_rows(n: int, rank: int, nranks: int) -> slice:
    """Rows of a global batch owned by ``rank`` (equal shards; the remainder goes to the last ranks)."""
    base, rem = divmod(n, nranks)
    start = rank * base + max(0, rank - (nranks - rem)) if rem else rank * base
    size = base + (1 if rem and rank >= nranks - rem else 0)
    return slice(start, start + size)


def allreduce_bucketed(tensors: List[torch.Tensor], bucket_bytes: int = 64 << 20) -> None:
    """Sum-all-reduce ``tensors`` in place, coalesced into flat buckets (few, large messages:
    xGMI rings are per-link bound)."""
    if not active():
        return
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        if len(bucket) == 1:
            all_reduce_(bucket[0])
        else:
            flat = torch.cat([t.reshape(-1) for t in bucket])
            all_reduce_(flat)
            ofs = 0
            for t in bucket:
                n = t.numel()
                t.copy_(flat[ofs:ofs + n].view_as(t))
                ofs += n
        bucket, size = [], 0

    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if nbytes >= bucket_bytes:
            flush()
            all_reduce_(t)
            continue
        if size + nbytes > bucket_bytes:
            flush()
        bucket.append(t)
        size += nbytes
    flush()


class FlatGrads:
    """Gradient storage of a model as ONE flat buffer with per-parameter views, laid out in the order the backward pass
    finishes them, so that a bucket of the exchange step is a contiguous slice: no ``torch.cat`` staging copy in front of
    a collective and no copy back behind it (round 3 had both).  ``order``: parameter names, first-finished first;
    ``local`` names (gradients that already belong to the global batch on every rank) go behind the reduced range."""

    def __init__(self, shapes: Dict[str, torch.Size], order: List[str], device, local=()):
        names = [k for k in order if k in shapes and k not in local] + [k for k in shapes if k not in order and k not in local]
        names += [k for k in shapes if k in local]
        self.offsets, ofs = {}, 0
        for k in names:
            n = 1
            for d in shapes[k]:
                n *= int(d)
            self.offsets[k] = (ofs, n)
            ofs += (n + 3) // 4 * 4                      # 16-byte aligned views (the fused NAdam kernel's requirement)
        self.flat = torch.zeros(ofs, dtype=torch.float32, device=device)
        self.views = {k: self.flat[o:o + n].view(shapes[k]) for k, (o, n) in self.offsets.items()}
        self.local = tuple(local)

    def span(self, names) -> torch.Tensor:
        """The contiguous slice that covers ``names`` (which must be adjacent in the layout)."""
        lo = min(self.offsets[k][0] for k in names)
        hi = max((self.offsets[k][0] + self.offsets[k][1] + 3) // 4 * 4 for k in names)
        return self.flat[lo:hi]


class _EventWork:
    """``work.wait()`` of a collective issued on a side stream: torch's current stream waits for the event behind it."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self) -> None:
        torch.cuda.current_stream().wait_event(self.ev)


class _Pending:
    """An all-reduce in flight.  ``wait()`` makes torch's current stream wait for it; with ``wait_events`` (a list) the
    time that stream actually spends blocked is measured by a pair of HIP events around the wait."""

    def __init__(self, work, tensor, host=None):
        self.work, self.tensor, self.host = work, tensor, host

    def wait(self, wait_events=None) -> None:
        if self.work is None:
            return
        if wait_events is not None and self.tensor.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.work.wait()
            e1.record()
            wait_events.append((e0, e1))
        else:
            self.work.wait()
        self.work = None


def all_reduce_async(t: torch.Tensor) -> _Pending:
    """Start a sum-all-reduce of ``t`` (in place) and return at once.  RCCL ("nccl"): the collective runs on the process
    group's own stream, ordered behind everything already enqueued on torch's current stream - i.e. behind the kernel
    that produced ``t`` - and BESIDE whatever the caller enqueues next (the rest of the backward pass).  gloo (tests: two
    ranks sharing one GPU, collectives staged through host memory): done synchronously."""
    if not active():
        return _Pending(None, t)
    if _tl_ok(t) and t.numel() > 0:
        # the C-ABI handle: on a stream of its own, behind what torch's current stream has enqueued so far.  (One communicator
        # serves this stream and the blocking collectives on the current stream: RCCL orders the operations of a communicator
        # by issue order, which is the same program order on every rank - single host thread, no data-dependent branches
        # between collectives - so the two streams cannot cross them.)
        side, cur = _TL[2], torch.cuda.current_stream()
        side.wait_stream(cur)
        t.record_stream(side)
        with torch.cuda.stream(side):
            _tl_allreduce(t, 0)
            ev = torch.cuda.Event()
            ev.record(side)
        return _Pending(_EventWork(ev), t)
    if t.is_cuda and _staged():
        all_reduce_(t)
        return _Pending(None, t)
    return _Pending(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t)


import contextlib


@contextlib.contextmanager
def skip_param_init():
    """Construct modules WITHOUT drawing their initial weights (``torch.nn.init`` draws become no-ops): for the ranks
    that receive rank 0's weights by ``broadcast_parameters_`` - at the north-star shape every rank would otherwise spend
    ~40 s of host RNG on 1.38 G values it is about to overwrite."""
    import torch.nn.init as init
    names = [n for n in ("uniform_", "normal_", "kaiming_uniform_", "kaiming_normal_", "xavier_uniform_", "xavier_normal_",
                         "trunc_normal_", "orthogonal_") if hasattr(init, n)]
    saved = {n: getattr(init, n) for n in names}
    try:
        for n in names:
            setattr(init, n, lambda tensor, *a, **k: tensor)
        yield
    finally:
        for n, fn in saved.items():
            setattr(init, n, fn)


def broadcast_parameters_(module: torch.nn.Module, src: int = 0) -> None:
    """Every parameter and buffer of ``module`` <- rank ``src``'s (in place; a collective)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if _tl_ok(t.data) and t.numel() > 0:
                if dist.get_rank() != src:       # (a sum whose other terms are zero: the handle has no broadcast entry)
                    t.data.zero_()
                _tl_allreduce(t.data, 0)
            elif t.is_cuda and _staged():
                h = t.detach().cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)
            else:
                dist.broadcast(t.data, src=src)


def gather_lowrank(dg: torch.Tensor, h: torch.Tensor, keys: Optional[torch.Tensor] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather the factors of the W_hh gradient: dg (k, 4H), h (k, H) with a per-rank k.
    Ranks pad to the common maximum with zero rows (which add nothing to dg^T . h).

    ``keys`` (k, m) identifies what each row's ``h`` was computed from (time step + label sequence):
    rows of different ranks with equal keys carry the same ``h`` up to the summation order of the
    kernel tiling a rank picked for its own number of distinct rows, so their ``dg`` rows are summed
    after the gather and ONE representative ``h`` stands for the group - the first occurrence in
    gathered (rank-major) order, chosen identically on every rank so the replicas stay in lockstep.
    The reduction length of the gradient GEMM then stays at the number of distinct (step, label)
    pairs of the GLOBAL batch instead of growing with the number of ranks."""
    if not active():
        return dg, h
    n = dist.get_world_size()
    k = torch.tensor([dg.shape[0]], dtype=torch.int64, device="cpu" if _staged() else dg.device)
    dist.all_reduce(k, op=dist.ReduceOp.MAX)
    kmax = int(k.item())

    def pad(t, fill=0.0):
        if t.shape[0] == kmax:
            return t.contiguous()
        out = torch.full((kmax, t.shape[1]), fill, dtype=t.dtype, device=t.device)
        out[:t.shape[0]] = t
        return out

    dg_all = torch.empty(n * kmax, dg.shape[1], dtype=dg.dtype, device=dg.device)
    h_all = torch.empty(n * kmax, h.shape[1], dtype=h.dtype, device=h.device)
    _all_gather_rows(dg_all, pad(dg))
    _all_gather_rows(h_all, pad(h))
    if keys is None:
        return dg_all, h_all
    keys_all = torch.empty(n * kmax, keys.shape[1], dtype=keys.dtype, device=keys.device)
    _all_gather_rows(keys_all, pad(keys.contiguous(), fill=float("-inf")))       # pad rows share one key, dg = 0
    _, inverse = torch.unique(keys_all, dim=0, return_inverse=True)
    groups = int(inverse.max().item()) + 1
    dg_sum = torch.zeros(groups, dg.shape[1], dtype=dg.dtype, device=dg.device).index_add_(0, inverse, dg_all)
    # representative h of a group: its first member in gathered order (deterministic on every rank)
    pos = torch.arange(inverse.numel(), device=inverse.device, dtype=torch.int64)
    first = torch.full((groups,), inverse.numel(), dtype=torch.int64, device=inverse.device)
    first.scatter_reduce_(0, inverse, pos, reduce="amin")
    return dg_sum, h_all.index_select(0, first)
