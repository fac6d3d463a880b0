"""bench.py's ONE JSON line, assembled on the host from what a run measured (bench.assemble_line: a pure function) -- so that
the first 8-GPU run cannot lose its curve to a formatting bug: a fake 8-rank run goes through the same code the real one
will, and the driver's contract keys, the SCALE evidence (rccl_ranks_seen, eight devices, per-rank rates, the gradient
bucket of the train side figure) and the side figures are checked on the parsed line.  No GPU, no oracle."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-collision-avoidance_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402
from mrca import scenario as S  # noqa: E402

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _args(**kw):
    a = argparse.Namespace(mode="env", steps=20, warmup=5, scenario="stage1", worlds=128, robots_per_world=32, fidelity=False,
                           no_graph=False, no_cpu_baseline=False, policy_dtype="f32", policy_path="fused", update_dtype="f32",
                           update_path="fused")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _line(world_size, extra, **kw):
    sc = S.stage1(num_worlds=128, robots_per_world=32, seed=1000)
    N = sc.num_robots
    elapsed = 20 * 31.3e-6
    per_rank = [N * 20 / (elapsed * (1 + 0.01 * r)) for r in range(world_size)] if world_size > 1 else None
    return bench.assemble_line(
        args=_args(**kw), sc=sc, N=N, world_size=world_size, value=N * world_size * 20 / elapsed, elapsed=elapsed,
        ray_ms=64 * 20.6e-3, mv_ms=64 * 10.8e-3, launches=64, kernel_timing_note="test",
        sched={"graph": False, "native": True, "chains": 2}, extra=extra, per_rank=per_rank,
        ranks_seen=world_size if world_size > 1 else None,
        devices=[f"rank {r}: cuda:{r} AMD Instinct MI355X (gfx950:sramecc+:xnack-, 256 CUs)" for r in range(world_size)]
        if world_size > 1 else None, backend="nccl" if world_size > 1 else None,
        cpu_baseline_fn=lambda *a, **k: {"value": 97.3e3, "unit": "agent-steps/s", "cores": 16, "kind": "port", "sample": "test"})


def test_single_gpu_line_has_the_contract_keys_roofline_and_cpu_baseline():
    extra = {"graph_prime_ticks": 0, "rollout_side_figure": {"value": 16e6, "unit": "agent-steps/s"},
             "stage2_side_figure": {"value": 2.0e8, "unit": "agent-steps/s", "robots": 8228},
             "fidelity_side_figure": {"value": 8.0e7, "unit": "agent-steps/s"},
             "reference_shaped_obs_side_figure": {"value": 1.0e8, "unit": "agent-steps/s"}}
    d = json.loads(json.dumps(_line(1, extra)))            # through JSON: what the driver parses
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "agent-steps/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert abs(d["value"] - 4096 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1.0
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["chains"] == 2
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["kernel_avg_us"] - 20.6) < 1e-6 and abs(r["kernel_sum_us"] - 31.4) < 1e-6
    assert "traffic" in r                                   # null or bytes: bench.pmc_traffic refuses a stale file
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 16 and "reference_structural_cap" in c
    for k in extra:
        assert k in d, k
    assert "per_rank_agent_steps_per_s" not in d and "rccl_ranks_seen" not in d


def test_eight_rank_line_carries_the_scale_evidence():
    extra = {"train_side_figure": {"value": 1.5e7, "unit": "agent-steps/s",
                                   "collective": {"backend": "nccl", "world_size": 8, "gradient_bucket_bytes": 2172101 * 4,
                                                  "optimizer_steps": 8}},
             "stage2_side_figure": {"value": 8 * 1.9e8, "per_rank_value": 1.9e8, "robots": 8228, "robots_all_ranks": 65824,
                                    "unit": "agent-steps/s"}}
    d = json.loads(json.dumps(_line(8, extra)))
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 8 and d["rccl_ranks_seen"] == 8 and d["collective_backend"] == "nccl"
    assert len(d["devices"]) == 8 and all("MI355X" in x for x in d["devices"]) and len(set(d["devices"])) == 8
    assert len(d["per_rank_agent_steps_per_s"]) == 8 and all(v > 0 for v in d["per_rank_agent_steps_per_s"])
    assert abs(d["value"] - 8 * 4096 * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 8.0      # whole job: all ranks' robots / max time
    assert "cpu_baseline" not in d                          # rank 0 at N = 1 only
    assert d["train_side_figure"]["collective"]["gradient_bucket_bytes"] == 8688404       # 2 172 101 fp32 parameters = 8.69 MB
    assert d["stage2_side_figure"]["robots_all_ranks"] >= 65536                              # BASELINE configs[3]
    assert d["roofline"]["value_at_kernel_sum"] > d["value"] / 2


def test_flop_model_and_the_side_figures_rooflines():
    """the rollout and train side figures carry a FLOP roofline of their own -- fp32 MFMA peak, FLOPs
    counted from the layer shapes of model/net.py:19-33 (checked here against an actual CNNPolicy's parameter shapes)."""
    from mrca.net import CNNPolicy
    pol = CNNPolicy(frames=3, action_space=2, beams=512)
    sd = pol.state_dict()
    l1, l2 = 255, 128
    macs = 0
    for tower in ("act", "crt"):
        w1, w2 = sd[f"{tower}_fea_cv1.weight"], sd[f"{tower}_fea_cv2.weight"]
        macs += w1.numel() * l1 + w2.numel() * l2 + sd[f"{tower}_fc1.weight"].numel() + sd[f"{tower}_fc2.weight"].numel()
    macs += sd["actor1.weight"].numel() + sd["actor2.weight"].numel() + sd["critic.weight"].numel()
    assert bench.policy_flops() == 2 * macs == 6390656
    r = bench.flop_roofline(16.8e6, bench.policy_flops(), "test", bench._kernel_rates()[0])
    assert r["bound"] == "fp32_mfma" and r["peak"] == 157.3 and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - 16.8e6 * 6390656 / 1e12) < 1e-9 and abs(r["frac"] - r["achieved"] / 157.3) < 1e-12
    assert 0.6 < r["frac"] < 0.75                                       # round 5's 16.8 M = 68 % of the fp32 roof
    assert all(set(k) >= {"kernel", "avg_us", "tflops", "frac", "source"} and os.path.exists(os.path.join(ROOT, k["source"]))
               for k in r["kernels"])
    t = bench.flop_roofline(2.4e6, bench.policy_flops() * (1 + 3 * 2), "test", bench._kernel_rates()[1])
    assert 0.65 < t["frac"] < 0.72                                      # round 5's 2.40 M = 68 %
    extra = {"rollout_side_figure": {"value": 16.8e6, "unit": "agent-steps/s", "roofline": r},
             "train_side_figure": {"value": 2.4e6, "unit": "agent-steps/s", "roofline": t, "collective": None,
                                   "hparams": {"horizon": 128, "epoch": 2, "minibatch_per_rank": 16384, "dtype": "f32"}}}
    d = json.loads(json.dumps(_line(1, extra)))
    assert d["train_side_figure"]["roofline"]["bound"] == "fp32_mfma" and d["rollout_side_figure"]["roofline"]["kernels"]


def test_the_action_pool_is_as_deep_as_the_region_it_serves():
    """actions i.i.d. per step -- the pool holds one batch per tick of the region (bench.main / env_side_figure ask
    for warm-up + steps entries, at most 1024), and tick k takes entry k % depth"""
    import torch
    pool = bench.action_pool(64, torch.device("cpu"), 3, depth=120)
    assert len(pool) == 120 and all(p.shape == (64, 2) and p.dtype == torch.float32 for p in pool)
    flat = torch.stack(pool)
    assert (flat[..., 0] >= 0).all() and (flat[..., 0] < 1).all() and (flat[..., 1] >= -1).all() and (flat[..., 1] < 1).all()
    assert len({tuple(p[0].tolist()) for p in pool}) == 120            # no entry repeats


def test_side_figures_have_their_scenarios_and_a_process_of_their_own():
    """bench.py --side-figure NAME: the three env side figures are measured by fresh processes (DESIGN.md 5.10 "streams"); the
    scenario of each is what the documents quote -- configs[2] (187 Stage-2 worlds x 44 robots = 8228), the same worlds at
    Stage's resolutions, the same worlds with the reference-shaped views formed every tick."""
    sc, lazy = bench.side_scenario("stage2_side_figure", "stage1", 128, 32, 1000)
    assert sc.num_robots == 8228 and sc.robots_per_world == 44 and lazy is True
    sc, lazy = bench.side_scenario("fidelity_side_figure", "stage1", 128, 32, 1000)
    assert sc.num_robots == 4096 and getattr(sc, "collision_raster", 0.0) == 0.2 and abs(sc.grid.cell - 0.2) < 1e-9 and lazy is True
    sc, lazy = bench.side_scenario("reference_shaped_obs_side_figure", "stage1", 128, 32, 1000)
    assert sc.num_robots == 4096 and getattr(sc, "collision_raster", 0.0) == 0.0 and lazy is False
    import pytest
    with pytest.raises(SystemExit):
        bench.side_scenario("no_such_figure", "stage1", 128, 32, 1000)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--side-figure" in sys.argv[1:]' in src and "subprocess.run(cmd" in src


def test_the_gradient_bucket_of_the_policy_is_8_69_MB():
    """what the train side figure's collective will report at N > 1: one flat fp32 bucket of every parameter of CNNPolicy"""
    import torch
    from mrca import ppo
    from mrca.net import CNNPolicy
    pol = CNNPolicy(frames=3, action_space=2, beams=512)
    fg = ppo.FlatGrads(pol.parameters())
    assert fg.flat.numel() == 2172101 and fg.flat.dtype == torch.float32 and fg.flat.numel() * 4 == 8688404


class _FakeEnv:
    """what TickSchedule touches of an env when the native schedule runs: W, device, step_many"""
    device = None

    def __init__(self, W):
        self.W, self.calls = W, []

    def step_many(self, pool, first, count, chains):
        self.calls.append((first, count, chains))


def test_world_ranges_of_the_schedule_cover_every_world_once():
    for W in (1, 2, 5, 128, 187):
        for P in (1, 2, 3, 4, 8):
            sched = bench.TickSchedule(_FakeEnv(W), list(range(16)), chains=P, native=True)
            ranges = sched.ranges
            assert sched.chains == min(P, W) == len(ranges)
            assert sum(n for _f, n in ranges) == W and ranges[0][0] == 0
            assert all(ranges[i][0] + ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1)) and min(n for _f, n in ranges) >= 1


def test_the_schedule_runs_exactly_the_ticks_it_is_asked_for():
    """chunks(): pieces of at most ticks_per_graph ticks that tile [first, first + count) in order -- the timed region is EXACTLY
    --steps ticks whatever the chunking -- and the native schedule hands each piece to ONE mrca_step_many call"""
    for first, count in ((0, 5), (5, 20), (100, 1000), (0, 64), (63, 2), (7, 129), (0, 0)):
        env = _FakeEnv(128)
        sched = bench.TickSchedule(env, list(range(16)), chains=2, native=True)
        pieces = list(sched.chunks(first, count))
        assert sum(m for _k, m in pieces) == count and all(0 < m <= sched.ticks_per_graph for _k, m in pieces)
        assert all(pieces[i][0] + pieces[i][1] == pieces[i + 1][0] for i in range(len(pieces) - 1))
        assert not pieces or (pieces[0][0] == first and pieces[-1][0] + pieces[-1][1] == first + count)
        sched.run(first, count)
        assert env.calls == [(k, m, 2) for k, m in pieces]
