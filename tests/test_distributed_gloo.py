"""N>1 path on CPU: two gloo ranks, each holding half of the robots' transitions, must produce the
same parameters as one process over the whole batch (flat-bucket gradient all-reduce + global
advantage statistics).  Pure torch on CPU tensors -- the env is not involved: worlds are
sharded with no data-path collective."""
import os
import socket
import tempfile

import numpy as np
import torch
import torch.multiprocessing as mp

import util as U  # noqa: F401  (sys.path)

T, N, B = 4, 8, 512


def _memory(seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    obss = r(T, N, 3, B) - 0.5
    goals = r(T, N, 2) * 10 - 5
    speeds = r(T, N, 2) * 2 - 1
    actions = r(T, N, 2) * 2 - 1
    logprobs = r(T, N, 1) * 0.2 - 2.0
    targets = torch.randn(T, N, generator=g)
    advs = torch.randn(T, N, generator=g) * 3 + 1
    return obss, goals, speeds, actions, logprobs, targets, None, None, advs


def _policy():
    from mrca.net import CNNPolicy
    torch.manual_seed(123)
    return CNNPolicy(3, 2)


def _local_batches(n_local):
    # two epochs x two minibatches, fixed so that the union over ranks is reproducible
    idx = torch.arange(n_local)
    return [idx[: n_local // 2], idx[n_local // 2:]]


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mrca import ppo
    from mrca.trainer import broadcast_parameters
    pol = _policy()
    if rank != 0:  # prove the broadcast: perturb non-root replicas first
        with torch.no_grad():
            for p in pol.parameters():
                p.add_(0.5)
    broadcast_parameters(pol, dist)
    opt = torch.optim.SGD(pol.parameters(), lr=1e-2)
    fg = ppo.FlatGrads(pol.parameters())
    mem = _memory()
    half = N // world
    local = tuple(None if m is None else m[:, rank * half:(rank + 1) * half].contiguous() for m in mem)
    ppo.ppo_update_stage1(policy=pol, optimizer=opt, batch_size=0, memory=local, epoch=2, coeff_entropy=5e-4,
                          clip_value=0.1, num_step=T, num_env=half, frames=3, obs_size=B, act_size=2,
                          index_batches=lambda n: _local_batches(n), dist=dist, flat_grads=fg)
    flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        assert torch.equal(gathered[0], gathered[1]), "replicas diverged"
        torch.save(flat, out)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_update_equals_single_process():
    from mrca import ppo
    out = os.path.join(tempfile.mkdtemp(), "params.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)

    pol = _policy()
    opt = torch.optim.SGD(pol.parameters(), lr=1e-2)
    mem = _memory()
    half = N // 2

    def union_batches(n):
        # rank r's local flat index j*half + i  <->  global flat index j*N + r*half + i
        outb = []
        for lb in _local_batches(T * half):
            j, i = lb // half, lb % half
            outb.append(torch.cat([j * N + r * half + i for r in range(2)]))
        return outb

    ppo.ppo_update_stage1(policy=pol, optimizer=opt, batch_size=0, memory=mem, epoch=2, coeff_entropy=5e-4,
                          clip_value=0.1, num_step=T, num_env=N, frames=3, obs_size=B, act_size=2,
                          index_batches=union_batches)
    want = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    assert float((got - want).abs().max()) < 2e-6, float((got - want).abs().max())
    # and the update did something
    assert float((want - torch.cat([p.detach().reshape(-1) for p in _policy().parameters()])).abs().max()) > 1e-5


def test_flat_grads_is_one_bucket():
    from mrca import ppo
    pol = _policy()
    fg = ppo.FlatGrads(pol.parameters())
    assert fg.flat.numel() == 2172101          # 8.69 MB fp32, one all-reduce per step
    base = fg.flat.data_ptr()
    off = 0
    for p in fg.params:
        assert p.grad.data_ptr() == base + 4 * off
        off += p.numel()


# ------------------------------------------------------------------------------------------------
# Stage 2: the filter leaves a different number of rows on each rank; both ranks must still take the SAME number of
# optimiser steps (one gradient all-reduce each), otherwise the collectives pair up wrongly and RCCL hangs.
def _worker_stage2(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    from mrca import ppo
    from mrca.trainer import broadcast_parameters
    pol = _policy()
    broadcast_parameters(pol, dist)
    opt = torch.optim.SGD(pol.parameters(), lr=1e-2)
    fg = ppo.FlatGrads(pol.parameters())
    mem = _memory()
    half = N // world
    local = tuple(None if m is None else m[:, rank * half:(rank + 1) * half].contiguous() for m in mem)
    n_local = T * half                                  # 16 rows per rank
    # rank 0 drops 1 row (15 kept -> 3 minibatches of 4 with drop_last), rank 1 drops 7 (9 kept -> 2 minibatches)
    filt = torch.arange(1) if rank == 0 else torch.arange(7)
    log = []
    ppo.ppo_update_stage2(policy=pol, optimizer=opt, batch_size=4, memory=local, filter_index=filt, epoch=2,
                          coeff_entropy=5e-4, clip_value=0.1, num_step=T, num_env=half, frames=3, obs_size=B,
                          act_size=2, dist=dist, flat_grads=fg, log=log)
    steps = torch.tensor([len(log)])
    both = [torch.zeros_like(steps) for _ in range(world)]
    dist.all_gather(both, steps)
    flat = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    # without flat_grads a multi-rank update must refuse to run (it would train unsynchronised replicas)
    refused = False
    try:
        ppo.ppo_update_stage1(policy=pol, optimizer=opt, batch_size=4, memory=local, epoch=1, num_step=T, num_env=half,
                              frames=3, obs_size=B, act_size=2, dist=dist, flat_grads=None)
    except ValueError:
        refused = True
    if rank == 0:
        assert n_local == 16
        assert int(both[0]) == int(both[1]) == 2 * 2, (both, "2 epochs x min(3, 2) minibatches on every rank")
        assert torch.equal(gathered[0], gathered[1]), "replicas diverged"
        assert refused
        torch.save(torch.tensor([1]), out)
    dist.destroy_process_group()


def test_two_rank_stage2_update_with_unequal_filtering_stays_in_step():
    out = os.path.join(tempfile.mkdtemp(), "ok.pt")
    mp.spawn(_worker_stage2, args=(2, _free_port(), out), nprocs=2, join=True)
    assert os.path.exists(out)


def _worker_kl(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mrca import ppo
    pol = _policy()
    opt = torch.optim.Adam(pol.parameters(), lr=1e-3)
    fg = ppo.FlatGrads(pol.parameters())
    mem = _memory(seed=rank)       # different data per rank: the local KLs differ, the decision must not
    half = N // world
    local = tuple(None if m is None else m[:, :half].contiguous() for m in mem)
    ctl = ppo.KLAdaptiveLR(target=1e-9, lr_min=1e-7)   # any movement exceeds this target: lr must shrink everywhere
    ppo.ppo_update_stage1(policy=pol, optimizer=opt, batch_size=8, memory=local, epoch=3, num_step=T, num_env=half,
                          frames=3, obs_size=B, act_size=2, dist=dist, flat_grads=fg, kl_ctl=ctl)
    lr = torch.tensor([opt.param_groups[0]["lr"], ctl.last_kl], dtype=torch.float64)
    both = [torch.zeros_like(lr) for _ in range(world)]
    dist.all_gather(both, lr)
    if rank == 0:
        assert torch.equal(both[0], both[1]), both
        assert abs(float(both[0][0]) - 1e-3 / 1.5 ** 3) < 1e-12
        torch.save(torch.tensor([1]), out)
    dist.destroy_process_group()


def test_kl_adaptive_lr_takes_the_same_decision_on_every_rank():
    out = os.path.join(tempfile.mkdtemp(), "ok.pt")
    mp.spawn(_worker_kl, args=(2, _free_port(), out), nprocs=2, join=True)
    assert os.path.exists(out)


# ------------------------------------------------------------------------------------------------
# The multi-rank TRAIN CLI end to end on two gloo ranks (the C oracle standing in for the device env): sharded worlds,
# flat-bucket gradient all-reduce, KL decisions, the collective stop, per-rank generator states in the checkpoint.
def _worker_train_cli(rank, world, port, workdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), MRCA_TRAIN_BACKEND="gloo")
    import cpu_train as ct
    from mrca import train, vec_env
    vec_env.VecStageWorld = ct.CpuEnv            # this worker process only
    vec_env.gae = ct._gae_cpu
    torch.set_num_threads(2)
    os.chdir(workdir)
    train.main(["--stage", "2", "--worlds", "1", "--updates", "3", "--save-every", "2", "--batch-size", "64", "--horizon", "8",
                "--epoch", "2", "--kl-target", "0.01", "--kl-stop", "2.0", "--max-grad-norm", "1.0", "--logstd-min", "-1.2",
                "--stock-policy-path", "--policy-dir", "policy", "--seed", "3"])


def test_train_cli_two_ranks_stage2_end_to_end():
    work = tempfile.mkdtemp()
    mp.spawn(_worker_train_cli, args=(2, _free_port(), work), nprocs=2, join=True)
    st = torch.load(os.path.join(work, "policy", "stage2_2.pth.state"), weights_only=False)
    assert st["global_update"] == 2 and len(st["generators"]) == 2                  # one noise generator state per rank
    assert not torch.equal(st["generators"][0], st["generators"][1])
    sd = torch.load(os.path.join(work, "policy", "last.pth"))
    assert float(sd["logstd"].min()) >= -1.2 and all(torch.isfinite(v).all() for v in sd.values())
    assert os.path.exists(os.path.join(work, "policy", "stage2_3.pth"))            # the run's last state is saved too
