"""N>1 path on CPU: two gloo ranks shard a batch of independent MSAs, run their shards
(with the CPU oracle standing in for the device kernels -- this test covers the sharding,
weight broadcast and merge-list gather, not the kernels) and gather the merge lists."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neuralnj_amd import sharding, synth, utils, weights


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from oracle_lib import Oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfgs = utils.shipped_config()
    cfgs.model.num_enc_layers = 1
    packed = torch.from_numpy(weights.pack(cfgs, weights.seeded_state(cfgs, 7 if rank == 0 else 99)))
    sharding.broadcast_weights(packed, dist)                   # rank 1 started with other weights
    codes = synth.synth_codes_tree(total, 6, 48, seed=3)
    lo, hi = sharding.shard_bounds(total, world, rank)
    o = Oracle(cfgs, packed.numpy())
    o.set_threads(2)
    r = o.rollout_argmax(synth.codes_to_onehot(codes[lo:hi]).astype(np.float32), None)
    allm = sharding.gather_merges(torch.from_numpy(r["merges"]), total, dist)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # the bench's max-over-ranks timing reduce
    assert t.item() == world
    np.save(os.path.join(out_dir, f"m{rank}.npy"), allm.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 4])
def test_two_rank_sharding_matches_single_process(tmp_path, total):
    from oracle_lib import Oracle
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    cfgs = utils.shipped_config()
    cfgs.model.num_enc_layers = 1
    o = Oracle(cfgs, weights.pack(cfgs, weights.seeded_state(cfgs, 7)))
    codes = synth.synth_codes_tree(total, 6, 48, seed=3)
    ref = o.rollout_argmax(synth.codes_to_onehot(codes).astype(np.float32), None)["merges"]
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"m{r}.npy"), ref)


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2, 2))]
    params[0].grad = torch.full((3, 5), float(rank + 1))
    params[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
    if rank == 0:
        params[2].grad = torch.ones(2, 2)                      # rank 1 has no gradient for this tensor
    sharding.allreduce_gradients(params, dist)
    np.save(os.path.join(out_dir, f"g{rank}.npy"), np.concatenate([p.grad.numpy().reshape(-1) for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_allreduce(tmp_path):
    """The one exchange step of the Finetune mode: gradients of all parameters summed over the ranks through one flat
    bucket; a tensor without a gradient on some rank contributes zeros."""
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = np.concatenate([np.full(15, 3.0), np.arange(7) * 3.0, np.ones(4)])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"g{r}.npy"), want)
