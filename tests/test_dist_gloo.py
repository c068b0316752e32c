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


@pytest.mark.gpu
def test_bench_two_ranks_on_the_device_match_single_rank_runs(tmp_path):
    """the REAL `bench.py --gpus 2` under torch.distributed.run, both ranks computing on the HIP library
    (one-GPU rehearsal: NNJ_BENCH_SHARE_GPU=1 puts both ranks on cuda:0, the tiny collectives run over gloo) -- the weight
    broadcast, per-rank shards, per-rank oracle verification, the all-reduced flags, the max-over-ranks timing and the
    all-gathered merge lists, which must equal two single-rank runs of the same shards.  The launcher starts before
    anything in the children touches the GPU; this process initialises no GPU context either."""
    import json
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shape = ["--batch", "16", "--taxa", "20", "--sites", "256", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
             "--no-compat", "--no-single-msa"]
    env = dict(os.environ, NNJ_BENCH_SHARE_GPU="1", NNJ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = tmp_path / "two.npy"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(repo, "bench.py"),
                        "--gpus", "2", "--dump-merges", str(two)] + shape,
                       cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    # the same from a BARE shell -- no launcher, WORLD_SIZE unset: bench.py starts the two ranks itself
    # (as a child process, before anything touches the GPU) and relays rank 0's line; n_gpus must be what was asked for
    bare_env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    two_b = tmp_path / "two_bare.npy"
    rb = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dump-merges", str(two_b)] + shape,
                        cwd=repo, env=bare_env, capture_output=True, text=True, timeout=900)
    assert rb.returncode == 0, rb.stderr[-2000:]
    out_lines = [x for x in rb.stdout.splitlines() if x.strip()]
    assert len(out_lines) == 1, out_lines                              # ONE JSON line on stdout
    bare = json.loads(out_lines[0])
    assert bare["n_gpus"] == 2 and len(bare["per_rank"]["trees_per_sec"]) == 2
    assert np.array_equal(np.load(two_b), np.load(two))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert len(line["per_rank"]["trees_per_sec"]) == 2 and min(line["per_rank"]["trees_per_sec"]) > 0
    assert line["value"] <= sum(line["per_rank"]["trees_per_sec"]) * 1.0001      # whole-job rate over the SLOWEST rank's time
    v = line["verified"]
    assert v["ok"] and v["all_ranks_ok"] and v["all_ranks_rf_gate_ok"] and v["ranks_verified"] == 2
    got = np.load(two)
    assert got.shape == (2, 16, 19, 2)
    for rank in range(2):
        one = tmp_path / f"one{rank}.npy"
        r1 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--data-rank", str(rank),
                             "--no-verify", "--no-profile", "--dump-merges", str(one)] + shape,
                            cwd=repo, env=dict(os.environ), capture_output=True, text=True, timeout=900)
        assert r1.returncode == 0, r1.stderr[-2000:]
        assert np.array_equal(np.load(one)[0], got[rank]), f"rank {rank}'s gathered merge lists differ from its single-rank run"


def _run_bench(args, env_extra=None, drop=()):
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK") + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + args, cwd=repo, env=env, capture_output=True,
                          text=True, timeout=900)


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` from a bare shell must never print an n_gpus: 1 line.  Here (any
    box): asking for more GPUs than the node shows is an error before anything is launched or measured."""
    r = _run_bench(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 64" in r.stderr and "refusing" in r.stderr
    assert not [x for x in r.stdout.splitlines() if x.startswith("{")]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """A launcher whose WORLD_SIZE disagrees with --gpus (either way, including WORLD_SIZE=1) is refused before any
    GPU call: the printed line would describe another run than the one asked for."""
    for world, gpus in (("1", "2"), ("2", "1"), ("2", "4")):
        r = _run_bench(["--gpus", gpus, "--steps", "1", "--warmup", "0"],
                       env_extra={"WORLD_SIZE": world, "RANK": "0", "LOCAL_RANK": "0"})
        assert r.returncode != 0
        assert f"--gpus {gpus} but WORLD_SIZE={world}" in r.stderr, r.stderr[-500:]


@pytest.mark.gpu
def test_bench_rccl_branch_runs_at_world_one(tmp_path):
    """the `"nccl"` (= RCCL) branch of bench.py executed on the one-GPU box.  NNJ_BENCH_FORCE_DIST=1
    makes a world-1 run go through init_process_group("nccl", device_id=...), the weight broadcast, both all_gathers, the
    all_reduce(MAX / MIN) of the timing and the verification flags, and the barriers -- all on DEVICE tensors; the line
    must say so and the merge lists must equal those of the plain run."""
    import json
    shape = ["--batch", "16", "--taxa", "20", "--sites", "256", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
             "--no-compat", "--no-single-msa"]
    a, b = tmp_path / "rccl.npy", tmp_path / "plain.npy"
    r = _run_bench(["--gpus", "1", "--dump-merges", str(a)] + shape,
                   env_extra={"NNJ_BENCH_FORCE_DIST": "1", "NNJ_BENCH_BACKEND": "nccl", "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
                   drop=("NNJ_BENCH_SHARE_GPU",))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 1
    pr = line["per_rank"]
    assert pr["backend"] == "nccl" and pr["distinct_devices"] == 1 and pr["device_ordinal"] == [0]
    v = line["verified"]
    assert v["ok"] and v["all_ranks_ok"] and v["all_ranks_rf_gate_ok"] and v["ranks_verified"] == 1
    r2 = _run_bench(["--gpus", "1", "--no-verify", "--no-profile", "--dump-merges", str(b)] + shape)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert np.array_equal(np.load(a), np.load(b))
