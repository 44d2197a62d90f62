"""CPU, world_size 2 (gloo): preprocess_signal(shard_channels=True) - rows split over the ranks for the channel-local
steps, all-gathered in front of a step that mixes channels and at the end - returns on every rank exactly what one
process returns.  The steps are test modules on CPU tensors (tests/shard_steps); the real kernels take the same path in
tests/test_gpu_dp.py."""
import os
from argparse import Namespace

import numpy as np
import torch
import torch.multiprocessing as mp

STEPS = [{"module": "tests.shard_steps.scale_rows"}, {"module": "tests.shard_steps.two_bands"},
         {"module": "tests.shard_steps.car_rereference", "params": {"expect_rows": 10}},
         {"module": "tests.shard_steps.scale_rows"}, {"module": "tests.shard_steps.two_bands"}]


def _data():
    return torch.from_numpy(np.random.default_rng(0).standard_normal((5, 16)))      # 5 rows over 2 ranks: 3 + 2 (padded)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from decode_tonal_langauge_amd import parallel
    from decode_tonal_langauge_amd.preprocess.preprocessor import preprocess_signal
    parallel.init_from_env(backend="gloo")
    out, fs = preprocess_signal(_data(), STEPS, Namespace(signal_freq=400), shard_channels=True)
    q.put((rank, out.numpy(), fs))
    torch.distributed.destroy_process_group()


def test_channel_sharded_dispatch_equals_single_process_world2_gloo():
    from decode_tonal_langauge_amd.preprocess.preprocessor import preprocess_signal
    ref, fs = preprocess_signal(_data(), STEPS, Namespace(signal_freq=400))
    assert ref.shape == (20, 16)
    same, _ = preprocess_signal(_data(), STEPS, Namespace(signal_freq=400), shard_channels=True)   # no process group: no-op
    assert torch.equal(same, ref)
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
    assert sorted(r for r, _, _ in res) == [0, 1]
    for rank, out, f in res:
        assert f == 400 and out.shape == (20, 16)
        assert np.array_equal(out, ref.numpy()), rank
