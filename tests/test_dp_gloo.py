"""CPU, world_size 2 (gloo): the data-parallel exchange helpers give the same result as one
process holding the whole batch."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from decode_tonal_langauge_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world() == (rank, world)
    torch.manual_seed(100)
    full = [torch.randn(7, 5), torch.randn(3), torch.randn(4000, 50), torch.randn(11)]
    mine = [t * (rank + 1) for t in full]
    parallel.allreduce_bucketed(mine, bucket_bytes=1 << 12)
    ok = all(torch.allclose(m, f * 3.0, rtol=1e-6, atol=1e-6) for m, f in zip(mine, full))
    # low-rank W_hh factors: ranks hold different numbers of rows
    torch.manual_seed(7 + rank)
    k = 3 + 2 * rank
    dg, h = torch.randn(k, 8), torch.randn(k, 6)
    dga, ha = parallel.gather_lowrank(dg, h)
    local = dg.t() @ h
    dist.all_reduce(local)
    ok = ok and torch.allclose(dga.t() @ ha, local, rtol=1e-5, atol=1e-5)
    # keyed gather: rows with equal keys (same h on every rank) are merged, the product is unchanged
    torch.manual_seed(3)
    h_pool, key_pool = torch.randn(5, 6), torch.arange(5, dtype=torch.float32).unsqueeze(1) * torch.ones(1, 3)
    pick = torch.tensor([0, 2, 4] if rank == 0 else [2, 3, 4, 1])
    dgk = torch.randn(len(pick), 8)
    dgm, hm = parallel.gather_lowrank(dgk, h_pool[pick], key_pool[pick])
    localk = dgk.t() @ h_pool[pick]
    dist.all_reduce(localk)
    ok = ok and dgm.shape[0] == 6 and torch.allclose(dgm.t() @ hm, localk, rtol=1e-5, atol=1e-5)   # 5 keys + the pad key
    # equal keys whose h differs in the last bits between ranks (different tilings): every rank must pick the SAME
    # representative - the first occurrence in gathered order, i.e. rank 0's row - or the replicas drift apart
    hp = h_pool[pick] + (1e-6 * rank)
    _dg2, hm2 = parallel.gather_lowrank(dgk, hp, key_pool[pick])
    mine_h = hm2.clone()
    both = [torch.empty_like(mine_h) for _ in range(world)]
    dist.all_gather(both, mine_h)
    ok = ok and torch.equal(both[0], both[1])
    shared = [k for k in (2, 4)]                      # keys held by both ranks: rank 0's (unperturbed) h must win
    for k in shared:
        row = (hm2 - h_pool[k]).abs().sum(dim=1).argmin()
        ok = ok and torch.equal(hm2[row], h_pool[k])
    # row-sharded parameter: blocks gathered rank-major, rows re-assembled in place
    blk = torch.full((3, 4), float(rank + 1))
    allb = parallel.all_gather_blocks(torch.empty(world, 3, 4), blk)
    ok = ok and torch.equal(allb[0], torch.full((3, 4), 1.0)) and torch.equal(allb[1], torch.full((3, 4), 2.0))
    torch.manual_seed(42)
    full = torch.randn(8, 5)
    mine_p = torch.zeros(8, 5)
    mine_p[rank * 4:(rank + 1) * 4] = full[rank * 4:(rank + 1) * 4]       # this rank holds only its rows up to date
    parallel.all_gather_param_rows_(mine_p, rank * 4, 4)
    ok = ok and torch.equal(mine_p, full)
    sl = parallel.shard_rows(10, rank, world)
    ok = ok and (sl.stop - sl.start == 5)
    # round 4: gradients in ONE flat buffer in completion order; the exchange buckets are contiguous slices of it, started
    # asynchronously as soon as their last gradient is final; "local" gradients stay out of the reduced range
    shapes = {"a.weight": torch.Size([5, 3]), "a.bias": torch.Size([5]), "b.weight": torch.Size([7, 2]), "c.bias": torch.Size([3]),
              "lstm.bias": torch.Size([6])}
    fg = parallel.FlatGrads(shapes, ["b.weight", "a.weight", "a.bias"], "cpu", local=("lstm.bias",))
    ok = ok and list(fg.offsets) == ["b.weight", "a.weight", "a.bias", "c.bias", "lstm.bias"]
    ok = ok and all(o % 4 == 0 for o, _ in fg.offsets.values())
    torch.manual_seed(11)
    ref = {k: torch.randn(v) for k, v in shapes.items()}
    for k, v in fg.views.items():
        v.copy_(ref[k] * (rank + 1))
    first = parallel.all_reduce_async(fg.span(["b.weight"]))            # "final" early: starts while the rest is computed
    second = parallel.all_reduce_async(fg.span(["a.weight", "a.bias", "c.bias"]))
    waits = []
    first.wait(waits)
    second.wait(waits)
    for k in shapes:
        want = ref[k] * (3.0 if k != "lstm.bias" else (rank + 1))        # the local gradient was not exchanged
        ok = ok and torch.allclose(fg.views[k], want, rtol=1e-6, atol=1e-6)
    # initial weights: rank 0 draws, the others skip their draws and receive the broadcast - the same bits as drawing everywhere
    torch.manual_seed(1234)
    everywhere = torch.nn.Sequential(torch.nn.Conv2d(1, 4, (3, 1)), torch.nn.LSTM(2, 8, batch_first=True), torch.nn.Linear(8, 3))
    torch.manual_seed(1234)
    import contextlib
    with (parallel.skip_param_init() if rank != 0 else contextlib.nullcontext()):
        net = torch.nn.Sequential(torch.nn.Conv2d(1, 4, (3, 1)), torch.nn.LSTM(2, 8, batch_first=True), torch.nn.Linear(8, 3))
    if rank != 0:
        ok = ok and not all(torch.equal(a, b) for a, b in zip(net.parameters(), everywhere.parameters()))   # really skipped
    parallel.broadcast_parameters_(net, src=0)
    ok = ok and all(torch.equal(a, b) for a, b in zip(net.parameters(), everywhere.parameters()))
    ok = ok and torch.nn.init.uniform_ is not None and torch.nn.init.uniform_(torch.zeros(4)).abs().sum() > 0   # patch undone
    # round 5: the self-check bench.py prints under data parallelism (dp_check): spread of two checksums of every parameter
    # over the ranks - exactly 0 for identical replicas, non-zero as soon as ONE element of one rank differs, NaN for NaN
    import bench
    # (collectives are never placed behind a short-circuiting `ok and`: a rank that skipped one would strand its peer)
    sp_same = bench.replica_spread([net], parallel, dist)
    with torch.no_grad():
        list(net.parameters())[1].view(-1)[3] += 1e-6 * rank            # rank 1 drifts by one element
    sp_drift = bench.replica_spread([net], parallel, dist)
    with torch.no_grad():
        list(net.parameters())[0].view(-1)[0] = float("nan") if rank == 1 else 0.0
    sp = bench.replica_spread([net], parallel, dist)                    # NaN on rank 1 only: NaN on BOTH ranks
    ok = ok and sp_same == 0.0 and sp_drift > 0.0 and sp != sp
    # classifiers drawn behind a model whose draws rank != 0 skipped: only a seed of their own makes them rank-identical
    torch.manual_seed(1234)
    with (parallel.skip_param_init() if rank != 0 else contextlib.nullcontext()):
        torch.nn.Linear(64, 64)
    drifted = torch.nn.Linear(16, 4)
    torch.manual_seed(1234 + 7)
    seeded = torch.nn.Linear(16, 4)
    sp_d, sp_s = bench.replica_spread([drifted], parallel, dist), bench.replica_spread([seeded], parallel, dist)
    ok = ok and sp_d > 0.0 and sp_s == 0.0
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_dp_helpers_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
