"""GPU: bench.py's multi-rank path (barrier, max-over-ranks, whole-job aggregate) exercised with two
ranks sharing the single GPU of the test box over gloo (test hook; production = one rank per GPU, RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

import util as U

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_same_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MRCA_BENCH_SAME_DEVICE="1", MRCA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_port()), os.path.join(U.ROOT, "bench.py"), "--gpus", "2", "--steps", "50",
           "--warmup", "10", "--scenario", "stage1", "--worlds", "16", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 50 and j["scaling"] == "weak"
    assert j["config"]["robots_per_gpu"] == 16 * 32
    # whole-job aggregate: robots on all ranks x steps / max-rank time
    assert abs(j["value"] - 2 * 16 * 32 * 50 / (j["ms_per_step"] * 1e-3 * 50)) / j["value"] < 1e-6
    assert len(j["per_rank_agent_steps_per_s"]) == 2 and j["collective_backend"] == "gloo"
    # the learner's collectives are on a measured path too: PPO updates with the flat gradient all-reduce
    tr = j["train_side_figure"]
    assert tr["collective"]["world_size"] == 2 and tr["collective"]["gradient_bucket_bytes"] == 2172101 * 4
    assert tr["collective"]["optimizer_steps"] >= 2 and tr["value"] > 0


def test_plain_python_bench_gpus_2_starts_its_own_two_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun must start two ranks itself (never a 1-GPU figure labelled as one): with
    the same-device / gloo hook on this 1-GPU box the line must say n_gpus 2 and the collective must have seen 2 ranks."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MRCA_BENCH_SAME_DEVICE="1", MRCA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(U.ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5",
                          "--worlds", "16", "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and len(j["per_rank_agent_steps_per_s"]) == 2


def test_plain_python_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus 8` on a box with fewer GPUs: one clear line on stderr, a non-zero exit code, NO JSON line."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    want = torch.cuda.device_count() + 7
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MRCA_BENCH_SAME_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(U.ROOT, "bench.py"), "--gpus", str(want), "--steps", "5",
                          "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "refusing" in out.stderr and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_bench_env_mode_as_hipgraph_counts_exactly_the_steps_asked_for():
    """--graph: the two-launch tick replayed as hipGraphs (16 ticks per graph + a remainder graph); the line must account
    for exactly --steps ticks and say so."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = subprocess.run([sys.executable, os.path.join(U.ROOT, "bench.py"), "--steps", "37", "--warmup", "21", "--worlds", "8",
                          "--no-cpu-baseline", "--no-extra", "--graph"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert j["steps"] == 37 and j["config"]["tick_as_hipgraph"] is True
    assert abs(j["value"] - 8 * 32 * 37 / (j["ms_per_step"] * 1e-3 * 37)) / j["value"] < 1e-6
    assert j["roofline"]["launches_timed"] == 64          # the kernel averages come from the eager pass after the region


def test_bench_multi_gpu_default_workload_is_the_single_gpu_one():
    """Weak scaling of ONE per-GPU workload: N > 1 defaults to configs[1] per GPU (128 Stage-1 rinks x 32 robots), like
    N = 1, so that value(N) / (N x value(1)) is a scaling efficiency; configs[3]'s per-GPU workload (187 Stage-2 worlds x
    44 robots) is ``--scenario stage2``."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MRCA_BENCH_SAME_DEVICE="1", MRCA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra, robots, prefix in (([], 128 * 32, "stage1: 128 worlds"), (["--scenario", "stage2"], 187 * 44, "stage2: 187 worlds")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_port()), os.path.join(U.ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
               "--warmup", "5", "--no-cpu-baseline", "--no-extra"] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert j["config"]["robots_per_gpu"] == robots and j["config"]["workload"].startswith(prefix)
        assert j["scaling"] == "weak" and j["n_gpus"] == 2


def test_bench_two_ranks_over_rccl():
    """One rank per GPU over RCCL (backend "nccl"): needs >= 2 visible devices, skipped on a 1-GPU box."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_port()), os.path.join(U.ROOT, "bench.py"), "--gpus", "2", "--steps", "50",
           "--warmup", "10", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["collective_backend"] == "nccl"
    assert j["train_side_figure"]["collective"]["backend"] == "nccl"
    assert j["train_side_figure"]["collective"]["world_size"] == 2


def test_bench_single_rank_json_contract():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = subprocess.run([sys.executable, os.path.join(U.ROOT, "bench.py"), "--steps", "40", "--warmup", "5",
                          "--worlds", "8", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the kernel averages are the launches' own begin / end stamps (one launch over all worlds each, in an eager pass after
    # the timed region).  Under the default run-ahead schedule a step is SHORTER than their sum -- the move launches run beside
    # the ray casts, two world ranges overlap -- so the only relation is to the one-chain schedule, checked below
    assert r["kernel_avg_us"] > 0 and r["move_kernel_avg_us"] > 0
    assert r["sustained"]["bytes_per_agent_step"] == 2140 and r["sustained"]["frac"] > 0
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and "workload" in j["config"]
    assert "run-ahead" in j["config"]["schedule"]
    out = subprocess.run([sys.executable, os.path.join(U.ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--worlds", "8",
                          "--no-cpu-baseline", "--no-extra", "--schedule", "chained", "--chains", "1"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    c = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    rc = c["roofline"]
    # one chain, in order: the two launches of a tick fit inside the step they are part of
    assert (rc["kernel_avg_us"] + rc["move_kernel_avg_us"]) * 1e-3 <= c["ms_per_step"] * 1.02


@pytest.mark.gpu
def test_two_envs_on_two_devices_in_one_process():
    """Two envs, the policy ops and the GAE kernel on two GPUs of ONE process while the caller's current device stays
    cuda:0: every entry point launches where its buffers live (per-device launch state, device guards keyed on the
    pointers).  Skips below two GPUs -- the first multi-GPU box runs it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import __graft_entry__ as g
    g.build()
    from mrca import policy_ops, ppo, scenario as S
    from mrca.net import CNNPolicy
    from mrca.vec_env import VecStageWorld
    torch.cuda.set_device(0)
    sc = S.stage1(num_worlds=4, robots_per_world=24, seed=3)
    envs = [VecStageWorld(sc, device=f"cuda:{d}") for d in (0, 1)]
    gen = torch.Generator().manual_seed(0)
    for e in envs:
        e.reset()
    for _ in range(5):
        a = torch.stack([torch.rand(sc.num_robots, generator=gen), torch.rand(sc.num_robots, generator=gen) * 2 - 1], 1)
        for e in envs:
            e.step(a.to(e.device).contiguous())
    for f in ("pose", "scan", "obs", "reward", "done", "local_goal"):
        assert torch.equal(getattr(envs[0], f).cpu(), getattr(envs[1], f).cpu()), f
    torch.manual_seed(1)
    pols = [CNNPolicy(3, 2).to(f"cuda:{d}") for d in (0, 1)]
    pols[1].load_state_dict(pols[0].state_dict())
    outs = []
    for e, p in zip(envs, pols):                       # current device is still cuda:0
        ring, head = e.policy_obs()
        lo, hi = (torch.tensor(b, device=e.device) for b in ((0.0, -1.0), (1.0, 1.0)))
        outs.append([t.cpu() for t in p.act_fused(ring, e.local_goal, e.speed, None, lo, hi, head=head)])
        t, adv = ppo.generate_train_data(torch.rand(8, e.N, device=e.device), 0.99, torch.rand(8, e.N, device=e.device),
                                         torch.rand(e.N, device=e.device),
                                         torch.zeros(8, e.N, dtype=torch.uint8, device=e.device), 0.95)
        assert t.device == e.device and bool(torch.isfinite(t).all())
        p.fused_train = True
        v, lp, ent = p.evaluate_actions(e.obs, e.local_goal, e.speed, torch.rand(e.N, 2, device=e.device))
        (lp.mean() + v.pow(2).mean()).backward()
        assert bool(torch.isfinite(p.act_fea_cv1.weight.grad).all())
    for a, b in zip(*outs):
        assert torch.allclose(a, b, atol=1e-6)
    assert torch.cuda.current_device() == 0
    for e in envs:
        e.close()
