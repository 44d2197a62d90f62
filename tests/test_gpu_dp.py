"""GPU: data-parallel equivalence of the fused trainer step.  Two processes share the one GPU of
the test box (gloo process group, collectives staged through host memory - RCCL refuses two ranks
on one device); each takes its row shard of the same global batches.  After two steps their
parameters must equal those of a single process that trained on the whole batches (dropout 0):
this exercises the row sharding, the bucketed all-reduce, the low-rank W_hh factor gather and the
1/N gradient scale exactly as the 8-GPU RCCL run uses them."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


LONG_TONE_MAP = {"0": [3] * 10, "1": list(range(1, 11)), "2": [3, 2, 1, 2, 4, 3, 2, 1, 2, 4], "3": list(range(10, 0, -1))}


def _build(dev, sizes=(8, 8), dropout=0.0, tone_map=None):
    from decode_tonal_langauge_amd.models import LogisticRegressionClassifier, SynthesisModelCNN, SynthesisTrainer
    from tests import golden_inputs as gi
    torch.manual_seed(0)
    model = SynthesisModelCNN(80, 8, 100, dropout=dropout)
    tone = LogisticRegressionClassifier(4 * 100, 4)
    syl = LogisticRegressionClassifier(4 * 100, 2)
    tr = SynthesisTrainer(model, tone, syl, tone_map or gi.TONE_MAP, device=dev, verbose=False)
    g = torch.Generator().manual_seed(11)
    batches = [(torch.randn(n, 8, 100, generator=g), torch.randn(n, 4, 100, generator=g),
                torch.randn(n, 4, 100, generator=g), 10 * torch.randn(n, 80, generator=g)) for n in sizes]
    return model, tr, batches


def _worker(rank, world, port, q, sizes=(8, 8), dropout=0.0, shard="1", tone_map=None, switch=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", TONAL_LSTM_SHARD=shard)
    from decode_tonal_langauge_amd import parallel
    parallel.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev, sizes, dropout, tone_map)
    assert tr.world == world
    model.train()
    for i, b in enumerate(batches):
        if switch is not None and i in switch:       # trainer.set_lstm_shard between steps (bench.py --lstm-shard auto)
            assert tr.set_lstm_shard(switch[i]) == switch[i]
        tr.train_step(*b)
    sharded = model._engine._sh is not None          # the label LSTM ran row-sharded over the two ranks
    tr.sync_parameters()                             # re-assemble the shard-wise updated W_hh on every rank
    torch.cuda.synchronize()
    # numpy (pickled by value): torch tensors would be shared through fds that die with this process
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}, tr._stats.cpu().numpy(), sharded))
    torch.distributed.destroy_process_group()


def test_two_rank_training_equals_single_process():
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev)
    model.train()
    for b in batches:
        tr.train_step(*b)
    ref = {k: v.detach().cpu() for k, v in model.named_parameters()}
    ref_stats = tr._stats.cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    from decode_tonal_langauge_amd.models import SynthesisModelCNN
    init = {k: v.detach().clone() for k, v in SynthesisModelCNN(80, 8, 100, dropout=0.0).named_parameters()}
    for rank, params, stats, sharded in res:
        assert sharded, "the trainer knows the label table: the LSTM must have run row-sharded"
        for k in ref:
            upd = (ref[k] - init[k]).double()
            err = float((torch.from_numpy(params[k]).double() - ref[k].double()).norm() / max(float(upd.norm()), 1e-30))
            assert err < 2e-2, (rank, k, err)          # relative to the size of the two-step update
    # both ranks hold identical parameters
    for k in ref:
        assert (res[0][1][k] == res[1][1][k]).all(), k
    # each rank accumulates its loss with its weight in the global mean: the sum over ranks is the global loss
    loss_dp = float(res[0][2][0]) + float(res[1][2][0])
    assert abs(loss_dp - float(ref_stats[0])) < 1e-3 * abs(float(ref_stats[0]))


def test_two_rank_training_ragged_batches_with_dropout():
    """Batches of 7 rows (uneven shards 3 + 4), 1 row (fewer rows than ranks: rank 1 recomputes row 0 with
    weight 0) and 8 rows, train-mode dropout 0.5: the dropout hash is indexed by the global element, so the
    two ranks draw exactly the masks of the single process and must end with its parameters."""
    sizes, drop = (7, 1, 8), 0.5
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev, sizes, drop)
    model.train()
    for b in batches:
        tr.train_step(*b)
    ref = {k: v.detach().cpu() for k, v in model.named_parameters()}
    ref_stats = tr._stats.cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sizes, drop)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    from decode_tonal_langauge_amd.models import SynthesisModelCNN
    init = {k: v.detach().clone() for k, v in SynthesisModelCNN(80, 8, 100, dropout=drop).named_parameters()}
    for rank, params, stats, sharded in res:
        assert sharded
        for k in ref:
            upd = (ref[k] - init[k]).double()
            err = float((torch.from_numpy(params[k]).double() - ref[k].double()).norm() / max(float(upd.norm()), 1e-30))
            assert err < 3e-2, (rank, k, err)
    for k in ref:
        assert (res[0][1][k] == res[1][1][k]).all(), k
    loss_dp = float(res[0][2][0]) + float(res[1][2][0])
    assert abs(loss_dp - float(ref_stats[0])) < 1e-3 * abs(float(ref_stats[0]))


def _rccl_worker(port, q):
    """One rank on the RCCL ("nccl") backend with the exchange step forced on: the same
    all_reduce / all_gather_into_tensor / barrier calls the multi-GPU run issues, on device tensors."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      TONAL_DP_FORCE="1")
    from decode_tonal_langauge_amd import parallel
    import torch.distributed as dist
    os.environ["WORLD_SIZE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert parallel.active() and not parallel._staged()
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev)
    assert tr.dp and tr.world == 1
    model.train()
    for b in batches:
        tr.train_step(*b)
    dist.barrier()
    torch.cuda.synchronize()
    q.put(({k: v.detach().cpu().numpy() for k, v in model.named_parameters()}, tr._stats.cpu().numpy()))
    dist.destroy_process_group()


def _tl_worker(port, q):
    """One rank whose exchange step runs through the RCCL handle of the C ABI (TONAL_DIST_BACKEND=tl: tl_comm_init /
    tl_allreduce / tl_all_gather on device buffers, the process group only as control plane), forced on with one rank; then
    the six entry points directly."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      TONAL_DP_FORCE="1", TONAL_DIST_BACKEND="tl")
    import ctypes as C
    from decode_tonal_langauge_amd import _lib, parallel
    import torch.distributed as dist
    rank, world, local = parallel.init_from_env()
    assert (rank, world) == (0, 1) and parallel.active() and parallel._TL is not None and dist.get_backend() == "gloo"
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev)
    assert tr.dp and tr.world == 1
    model.train()
    for b in batches:
        tr.train_step(*b)
    torch.cuda.synchronize()
    out = ({k: v.detach().cpu().numpy() for k, v in model.named_parameters()}, tr._stats.cpu().numpy())
    # ---- the entry points themselves (one rank: every collective is the identity)
    lib, comm, _ = parallel._TL
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(1000, device=dev)
    for op in (0, 1, 2):
        y = torch.empty_like(x)
        _lib.check(lib.tl_allreduce(comm, x.data_ptr(), y.data_ptr(), x.numel(), op, st), "tl_allreduce")
        torch.cuda.synchronize()
        assert torch.equal(x, y)
    y = torch.empty_like(x)
    _lib.check(lib.tl_all_gather(comm, x.data_ptr(), y.data_ptr(), x.numel(), st), "tl_all_gather")
    z = torch.empty_like(x)
    _lib.check(lib.tl_reduce_scatter(comm, x.data_ptr(), z.data_ptr(), x.numel(), st), "tl_reduce_scatter")
    torch.cuda.synchronize()
    assert torch.equal(x, y) and torch.equal(x, z)
    t = torch.arange(12, device=dev, dtype=torch.float32).view(3, 4)
    assert torch.equal(parallel.all_gather_param_rows_(t.clone(), 0, 3), t)          # in place through the handle
    # bench.py's replica check carried by the handle (fp32 pieces of the fp64 checksums through tl_allreduce MIN / MAX)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert parallel.tl_active() and bench.replica_spread_tl([model, tr.tone_model], parallel) == 0.0
    bad = torch.nn.Linear(3, 2).to(dev)
    with torch.no_grad():
        bad.weight[0, 0] = float("nan")
    assert bench.replica_spread_tl([bad], parallel) != 0.0                             # a non-finite checksum reads as NaN
    pend = parallel.all_reduce_async(x.clone())
    pend.wait()
    assert lib.tl_allreduce(None, x.data_ptr(), x.data_ptr(), 4, 0, st) != 0 and b"allreduce" in lib.tl_last_error()
    assert lib.tl_allreduce(comm, x.data_ptr(), x.data_ptr(), 4, 7, st) != 0
    assert lib.tl_comm_init(None, 0, 1, None) != 0 and lib.tl_comm_destroy(None) != 0
    parallel.tl_comm_destroy()
    assert parallel._TL is None
    q.put(out)
    dist.destroy_process_group()


def test_exchange_step_over_the_c_abi_rccl_handle_single_rank():
    """SURVEY 8b's RCCL handle (tl_comm_* / tl_allreduce / tl_reduce_scatter / tl_all_gather): a training run whose exchange
    step goes through it ends on the single-process parameters; RCCL refuses two ranks per device, so one rank with the
    exchange forced on is what a one-GPU box can run."""
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev)
    model.train()
    for b in batches:
        tr.train_step(*b)
    ref = {k: v.detach().cpu() for k, v in model.named_parameters()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_tl_worker, args=(31500 + (os.getpid() % 1000), q))
    p.start()
    params, stats = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    torch.manual_seed(0)
    from decode_tonal_langauge_amd.models import SynthesisModelCNN
    init = {k: v.detach().clone() for k, v in SynthesisModelCNN(80, 8, 100, dropout=0.0).named_parameters()}
    for k in ref:
        upd = (ref[k] - init[k]).double()
        err = float((torch.from_numpy(params[k]).double() - ref[k].double()).norm() / max(float(upd.norm()), 1e-30))
        assert err < 2e-2, (k, err)
    assert abs(float(stats[0]) - float(tr._stats[0])) < 1e-3 * abs(float(tr._stats[0]))


def test_exchange_step_over_rccl_single_rank():
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev)
    model.train()
    for b in batches:
        tr.train_step(*b)
    ref = {k: v.detach().cpu() for k, v in model.named_parameters()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(30500 + (os.getpid() % 1000), q))
    p.start()
    params, stats = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    torch.manual_seed(0)
    from decode_tonal_langauge_amd.models import SynthesisModelCNN
    init = {k: v.detach().clone() for k, v in SynthesisModelCNN(80, 8, 100, dropout=0.0).named_parameters()}
    for k in ref:
        upd = (ref[k] - init[k]).double()
        err = float((torch.from_numpy(params[k]).double() - ref[k].double()).norm() / max(float(upd.norm()), 1e-30))
        assert err < 2e-2, (k, err)
    assert abs(float(stats[0]) - float(tr._stats[0])) < 1e-3 * abs(float(tr._stats[0]))


def test_two_rank_training_unsharded_lstm_reduces_the_factor_rows():
    """TONAL_LSTM_SHARD=0: every rank keeps the whole label LSTM.  The trainer still knows the label table, so row
    (step, table row) of the W_hh gradient factors means the same thing on every rank and the global factor is one
    small all-reduce of the dgates rows (no gather, no torch.unique, no host synchronisation) - the two ranks must
    again end on the single-process parameters."""
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev)
    model.train()
    for b in batches:
        tr.train_step(*b)
    ref = {k: v.detach().cpu() for k, v in model.named_parameters()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, (8, 8), 0.0, "0")) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    from decode_tonal_langauge_amd.models import SynthesisModelCNN
    init = {k: v.detach().clone() for k, v in SynthesisModelCNN(80, 8, 100, dropout=0.0).named_parameters()}
    for rank, params, stats, sharded in res:
        assert not sharded
        for k in ref:
            upd = (ref[k] - init[k]).double()
            err = float((torch.from_numpy(params[k]).double() - ref[k].double()).norm() / max(float(upd.norm()), 1e-30))
            assert err < 2e-2, (rank, k, err)
    for k in ref:
        assert (res[0][1][k] == res[1][1][k]).all(), k


def _two_rank_run(port, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q), kwargs=kw) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_training_when_the_sharded_lstm_falls_back_reduces_every_gradient_once():
    """A ten-entry tone dynamic: (L - 1) U = 72 factor rows exceed what the row-sharded LSTM (and the fused low-rank
    NAdam kernel) takes, so the engine runs the whole LSTM on every rank for the step although the trainer asked for the
    sharded form.  The flat gradient layout was fixed for the sharded form and the output layer's all-reduce is already
    in flight when that is known: every gradient must still be reduced exactly ONCE (round 4 reduced the output layer
    twice on this path: 63 MB of gradients N times too large, silently)."""
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev, tone_map=LONG_TONE_MAP)
    model.train()
    for b in batches:
        tr.train_step(*b)
    ref = {k: v.detach().cpu() for k, v in model.named_parameters()}
    res = _two_rank_run(33500 + (os.getpid() % 1000), tone_map=LONG_TONE_MAP)
    torch.manual_seed(0)
    from decode_tonal_langauge_amd.models import SynthesisModelCNN
    init = {k: v.detach().clone() for k, v in SynthesisModelCNN(80, 8, 100, dropout=0.0).named_parameters()}
    for rank, params, stats, sharded in res:
        assert not sharded, "72 factor rows: the engine must have fallen back to the whole LSTM per rank"
        for k in ref:
            upd = (ref[k] - init[k]).double()
            err = float((torch.from_numpy(params[k]).double() - ref[k].double()).norm() / max(float(upd.norm()), 1e-30))
            assert err < 2e-2, (rank, k, err)
    for k in ref:
        assert (res[0][1][k] == res[1][1][k]).all(), k


def test_lstm_shard_can_be_switched_between_steps():
    """trainer.set_lstm_shard (bench.py --lstm-shard auto probes both forms during warm-up): sharded step, whole-LSTM step,
    sharded step.  The NAdam state of W_hh follows the switch (moments cut to the rank's rows / all-gathered, step count and
    mu product kept - ADVICE round 5), so the three steps stay on the single-process trajectory: compared with one process
    at the tolerance of the two-rank test, and bit-identical between the ranks after the re-assembly."""
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev, sizes=(8, 8, 8))
    model.train()
    for b in batches:
        tr.train_step(*b)
    ref = {k: v.detach().cpu() for k, v in model.named_parameters()}
    torch.manual_seed(0)
    from decode_tonal_langauge_amd.models import SynthesisModelCNN
    init = {k: v.detach().clone() for k, v in SynthesisModelCNN(80, 8, 100, dropout=0.0).named_parameters()}
    res = _two_rank_run(34500 + (os.getpid() % 1000), sizes=(8, 8, 8), switch={1: False, 2: True})
    import numpy as np
    for rank, params, stats, sharded in res:
        assert sharded
        assert all(np.isfinite(v).all() for v in params.values())
        for k in ref:
            upd = (ref[k] - init[k]).double()
            err = float((torch.from_numpy(params[k]).double() - ref[k].double()).norm() / max(float(upd.norm()), 1e-30))
            assert err < 2e-2, (rank, k, err)          # relative to the size of the three-step update
    for k in res[0][1]:
        assert (res[0][1][k] == res[1][1][k]).all(), k


def _bench(extra_env, *flags, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_two_rank_launch_at_the_headline_shape():
    """`python bench.py --gpus 2` at the north-star shape (C4's first point), both ranks on the test GPU over gloo
    (TONAL_BENCH_SHARE_GPU=1; RCCL refuses two ranks per device): the launcher starts the ranks, the global batch of
    256 is sharded 128 + 128, the exchange step is timed, and rank 0 prints the one JSON line of the contract."""
    import math
    r, line = _bench({"TONAL_BENCH_SHARE_GPU": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                     "--no-extras")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line is not None and line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    cfg = line["config"]
    assert cfg["backend"] == "gloo" and cfg["per_gpu_batch"] == 128 and cfg["global_batch"] == 256
    assert cfg["parallelism"] == "dp2" and line["scaling"] == "strong"
    assert math.isfinite(cfg["exchange_ms_per_step"]) and cfg["exchange_ms_per_step"] > 0
    assert line["value"] > 0 and abs(line["value"] - 256 / (line["ms_per_step"] * 1e-3)) < 1e-2 * line["value"]
    # the run verifies itself: identical replicas before and after, both LSTM forms probed, the first step's loss reported
    chk = line["dp_check"]
    assert chk["ok"] and chk["init_checksum_spread"] == 0.0 and chk["param_checksum_spread"] == 0.0
    assert chk["ranks_seen"] == 2 and chk["backend"] == "gloo"
    assert set(chk["lstm"]["probe_ms_per_step"]) == {"sharded", "whole"} and chk["lstm"]["requested"] == "auto"
    assert math.isfinite(line["step0"]["loss"]) and line["step0"]["loss"] > 0
    if line["step0"].get("golden"):           # the committed 1-GPU value of the same seeded global batch
        assert line["step0"]["ok"], line["step0"]


def test_bench_under_torch_distributed_run():
    """The driver's launch form for N > 1 - ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`` - with three ranks on the test GPU over gloo (TONAL_BENCH_SHARE_GPU=1): RANK /
    LOCAL_RANK / WORLD_SIZE come from the launcher, rank 0 prints the one JSON line, the replicas end identical."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TONAL_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2",
                        "--warmup", "1", "--batch", "9", "--channels", "8", "--timepoints", "100", "--no-cpu-baseline",
                        "--no-extras"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["config"]["parallelism"] == "dp3" and line["config"]["backend"] == "gloo"
    assert line["config"]["per_gpu_batch"] == 3 and line["config"]["global_batch"] == 9
    assert line["dp_check"]["ok"] and line["dp_check"]["ranks_seen"] == 3 and line["dp_check"]["param_checksum_spread"] == 0.0


def test_bench_ends_non_zero_when_a_rank_dies():
    r, line = _bench({"TONAL_BENCH_SHARE_GPU": "1", "TONAL_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "1",
                     "--warmup", "0", "--batch", "8", "--channels", "8", "--timepoints", "100", "--no-cpu-baseline",
                     "--no-extras", timeout=600)
    assert r.returncode != 0 and line is None
    assert "exits on request" in r.stderr


def _signal_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from argparse import Namespace
    from copy import deepcopy
    from decode_tonal_langauge_amd import parallel
    from decode_tonal_langauge_amd.preprocess import preprocessor
    from tests import golden_inputs as gi
    parallel.init_from_env(backend="gloo")
    x = gi.chain_input()
    outs = []
    for rows in (6, 5):                               # 3 + 3 rows, then 3 + 2 (the last shard padded)
        out, fs = preprocessor.preprocess_signal(x[:rows].copy(), deepcopy(gi.CHAIN_STEPS), Namespace(signal_freq=1000),
                                                 shard_channels=True)
        outs.append((out, fs))
    q.put((rank, outs))
    torch.distributed.destroy_process_group()


def test_channel_sharded_signal_chain_equals_single_process():
    """SURVEY 8e, signal path: downsample -> CAR -> band extraction (two entries) -> z-score with the recording's
    channels split over two ranks (both on the test GPU, gloo): channel-local steps without communication, shards
    gathered in front of the common-average step and at the end.  Same kernels on the same numbers: identical output."""
    import numpy as np
    from argparse import Namespace
    from copy import deepcopy
    from decode_tonal_langauge_amd.preprocess import preprocessor
    from tests import golden_inputs as gi
    x = gi.chain_input()
    refs = [preprocessor.preprocess_signal(x[:rows].copy(), deepcopy(gi.CHAIN_STEPS), Namespace(signal_freq=1000))
            for rows in (6, 5)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_signal_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outs in res:
        for (out, fs), (ref, ref_fs) in zip(outs, refs):
            assert fs == ref_fs == 400 and isinstance(out, np.ndarray) and out.shape == ref.shape
            assert np.array_equal(out, ref, equal_nan=True), (rank, float(np.nanmax(np.abs(out - ref))))


def _tl_worker_multi(rank, world, port, q):
    """One rank per GPU through the C-ABI RCCL handle (TONAL_DIST_BACKEND=tl): the collectives that are the identity with one
    rank - the in-place row gather, the zero-fill-and-sum broadcast, the asynchronous all-reduce on the handle's side stream
    beside blocking collectives on the current stream - and two train steps."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), TONAL_DIST_BACKEND="tl")
    from decode_tonal_langauge_amd import parallel
    import torch.distributed as dist
    r, w, local = parallel.init_from_env()
    assert (r, w, local) == (rank, world, rank) and parallel.tl_active()
    dev = torch.device(f"cuda:{rank}")
    # in-place row gather: rank r owns rows [2 r, 2 r + 2)
    full = torch.arange(2 * world * 3, device=dev, dtype=torch.float32).view(2 * world, 3)
    mine = torch.zeros_like(full)
    mine[2 * rank:2 * rank + 2] = full[2 * rank:2 * rank + 2]
    assert torch.equal(parallel.all_gather_param_rows_(mine, 2 * rank, 2), full)
    try:
        parallel.all_gather_param_rows_(mine, 0 if rank else 2, 2)            # a shard that is not the rank's own block
        raise AssertionError("misplaced row shard accepted")
    except ValueError:
        pass
    # broadcast: rank 0's values everywhere
    lin = torch.nn.Linear(5, 3).to(dev)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    parallel.broadcast_parameters_(lin, src=0)
    assert float(lin.weight.min()) == 1.0 and float(lin.weight.max()) == 1.0
    # asynchronous sum beside a blocking MAX on the current stream
    a = torch.full((1 << 20,), float(rank + 1), device=dev)
    pend = parallel.all_reduce_async(a)
    b = torch.full((7,), float(rank), device=dev)
    parallel.all_reduce_(b, op=dist.ReduceOp.MAX)
    pend.wait()
    torch.cuda.synchronize()
    assert float(a.min()) == float(a.max()) == world * (world + 1) / 2 and float(b.max()) == world - 1
    model, tr, batches = _build(dev)
    model.train()
    for bt in batches:
        tr.train_step(*bt)
    tr.sync_parameters()
    torch.cuda.synchronize()
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.replica_spread_tl([model], parallel) == 0.0                           # identical replicas, seen through the handle
    odd = torch.nn.Linear(3, 2).to(dev)
    with torch.no_grad():
        odd.weight.fill_(float(rank))
    assert bench.replica_spread_tl([odd], parallel) > 0.0                              # ... and a diverged one is caught
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}))
    parallel.tl_comm_destroy()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_tl_handle_with_two_gpus():
    """the multi-rank path of the C-ABI RCCL handle.  Skipped on the one-GPU boxes this suite normally meets."""
    dev = torch.device("cuda:0")
    model, tr, batches = _build(dev)
    model.train()
    for b in batches:
        tr.train_step(*b)
    ref = {k: v.detach().cpu() for k, v in model.named_parameters()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_tl_worker_multi, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    from decode_tonal_langauge_amd.models import SynthesisModelCNN
    init = {k: v.detach().clone() for k, v in SynthesisModelCNN(80, 8, 100, dropout=0.0).named_parameters()}
    for rank, params in res:
        for k in ref:
            upd = (ref[k] - init[k]).double()
            err = float((torch.from_numpy(params[k]).double() - ref[k].double()).norm() / max(float(upd.norm()), 1e-30))
            assert err < 2e-2, (rank, k, err)
    for k in ref:
        assert (res[0][1][k] == res[1][1][k]).all(), k
