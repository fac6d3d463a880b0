"""GPU: the vectorised trainer (rollout buffer, HIP GAE, PPO update, checkpoint round trip)."""
import numpy as np
import pytest
import torch

import util as U
from util import S, O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    from mrca import vec_env
    return vec_env


def test_stage1_trainer_one_update_and_gae_consistency(hip):
    from mrca.trainer import HParams, Stage1Trainer
    sc = S.stage1(num_worlds=4, robots_per_world=8, seed=2)
    env = hip.VecStageWorld(sc)
    hp = HParams(horizon=16, batch_size=128, epoch=2)
    tr = Stage1Trainer(env, hp=hp, seed=1)
    before = torch.cat([p.detach().reshape(-1).clone() for p in tr.policy.parameters()])
    tr.run(15)
    assert tr.global_update == 0
    buf = tr.buffer
    # what the buffer holds at t is the state the action was computed from
    stacks = buf.obs_rows().materialise()
    assert torch.isfinite(stacks[:15]).all() and float(stacks[:15].abs().max()) <= 0.5
    tr.run(1)                                   # 16th tick triggers the update
    assert tr.global_update == 1 and len(tr.loss_log) == 2 * 4     # 512 samples / 128 x 2 epochs
    after = torch.cat([p.detach().reshape(-1) for p in tr.policy.parameters()])
    assert float((after - before).abs().max()) > 0
    assert all(torch.isfinite(torch.stack(x)).all() for x in tr.loss_log)
    # GAE of the stored rollout == oracle restatement of generate_train_data (fp32, bitwise)
    with torch.no_grad():
        _m, last_v = tr.policy.mean_value(env.obs, env.local_goal, env.speed)
    tg, adv = hip.gae(buf.reward, buf.value, last_v.reshape(-1).contiguous(), buf.done, hp.gamma, hp.lam)
    t32, a32 = O.gae(buf.reward.cpu().numpy(), buf.value.cpu().numpy(), last_v.reshape(-1).cpu().numpy(),
                     buf.done.cpu().numpy(), hp.gamma, hp.lam, np.float32)
    assert (tg.cpu().numpy().view(np.uint32) == t32.view(np.uint32)).all()
    env.close()


def test_stage2_trainer_filtered_update(hip):
    from mrca.trainer import HParams, Stage1Trainer
    sc = S.stage2(num_worlds=2, seed=3)
    env = hip.VecStageWorld(sc)
    hp = HParams(horizon=24, batch_size=256, epoch=1)
    tr = Stage1Trainer(env, hp=hp, seed=1, stage2=True)
    tr.run(24)
    assert tr.global_update == 1 and len(tr.loss_log) >= 1
    env.close()


def test_checkpoint_roundtrip_reference_keys(hip, tmp_path):
    """policy/*.pth compatibility: a state_dict saved here has exactly the reference's keys
    (ppo_stage1.py:122-124,185-191; model/net.py:19-33)."""
    from mrca.net import CNNPolicy
    pol = CNNPolicy(3, 2).cuda()
    f = tmp_path / "stage1_test.pth"
    torch.save(pol.state_dict(), f)
    sd = torch.load(f)
    want = {"logstd", "critic.weight", "critic.bias", "actor1.weight", "actor1.bias", "actor2.weight", "actor2.bias"}
    for tw in ("act", "crt"):
        for layer in ("fea_cv1", "fea_cv2", "fc1", "fc2"):
            want |= {f"{tw}_{layer}.weight", f"{tw}_{layer}.bias"}
    assert set(sd.keys()) == want
    pol2 = CNNPolicy(3, 2).cuda()
    pol2.load_state_dict(sd)
    x = torch.rand(5, 3, 512, device="cuda") - 0.5
    g, s = torch.rand(5, 2, device="cuda"), torch.rand(5, 2, device="cuda")
    with torch.no_grad():
        a = pol.mean_value(x, g, s)
        b = pol2.mean_value(x, g, s)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_train_cli_checkpoint_and_resume(hip, tmp_path, monkeypatch):
    """python -m mrca.train: log streams, reference-named checkpoints, exact resume state."""
    import os
    from mrca import train
    monkeypatch.chdir(tmp_path)
    args = ["--stage", "1", "--worlds", "2", "--robots-per-world", "6", "--updates", "2", "--save-every", "1",
            "--batch-size", "256"]
    train.main(args)
    assert os.path.exists(tmp_path / "policy" / "Stage1_1") and os.path.exists(tmp_path / "policy" / "Stage1_2.state")
    logs = os.listdir(tmp_path / "log")
    d = tmp_path / "log" / logs[0]
    assert {"output.log", "cal.log", "ppo.log"} <= set(os.listdir(d))
    lines = [ln for ln in open(d / "ppo.log").read().splitlines() if ln.strip()]
    assert len(lines) == 2 * 2 * 6          # 2 updates x 2 epochs x ceil(1536/256)
    assert all(len(ln.split(",")) == 3 for ln in lines)
    # resume: the reference looks for policy/stage1_2.pth (ppo_stage1.py:185)
    os.replace(tmp_path / "policy" / "Stage1_2", tmp_path / "policy" / "stage1_2.pth")
    os.replace(tmp_path / "policy" / "Stage1_2.state", tmp_path / "policy" / "stage1_2.pth.state")
    train.main(["--stage", "1", "--worlds", "2", "--robots-per-world", "6", "--updates", "1", "--save-every", "1",
                "--batch-size", "256"])
    assert os.path.exists(tmp_path / "policy" / "Stage1_3")      # global_update continued from 2


@pytest.mark.parametrize("fused,via_run", [(False, False), (True, False), (True, True)])
def test_graph_captured_tick_fills_the_buffer_consistently(hip, fused, via_run):
    """The rollout tick replayed as a hipGraph (device-side buffer row, registered noise generator): every stored row
    must be self-consistent -- the stored log-probability is the policy's density of the stored action at the stored
    state, the stored value its value -- and rows, rewards and done flags must line up with the env's own fields.
    ``via_run``: through ``Stage1Trainer.run`` -- one replay of the eight-tick graph (its noise drawn in one launch) and
    three of the one-tick graph."""
    from mrca.trainer import HParams, Stage1Trainer
    sc = S.stage1(num_worlds=4, robots_per_world=8, seed=5)
    env = hip.VecStageWorld(sc)
    hp = HParams(horizon=12, batch_size=128, epoch=1, graph_tick=True, rollout_fused=fused)
    tr = Stage1Trainer(env, hp=hp, seed=2)
    tr.start()
    if via_run:
        tr.run(11)
        assert tr._graph_run is not None
    else:
        for _ in range(11):
            tr.tick()
    torch.cuda.synchronize()
    buf = tr.buffer
    assert tr.global_update == 0 and int(tr._t_idx) == 11 and tr.t == 11
    stacks = buf.obs_rows().materialise()
    with torch.no_grad():
        for t in range(11):
            v, lp, _ent = tr.policy.evaluate_actions(stacks[t], buf.goal[t], buf.speed[t], buf.action[t])
            assert float((lp - buf.logprob[t]).abs().max()) < 2e-5, t
            assert float((v.view(-1) - buf.value[t]).abs().max()) < 1e-4, t
    assert torch.equal(buf.reward[10], env.reward) and torch.equal(buf.done[10], env.done)
    assert float(buf.action[:11].std()) > 0.3                    # the noise generator advances between replays
    assert not torch.equal(buf.action[3], buf.action[4])
    # consecutive rows of a robot that did not restart share two of their three frames
    keep = ~(buf.done[4].bool())
    assert torch.equal(stacks[5][keep][:, :2], stacks[4][keep][:, 1:])
    tr.tick()                                                    # 12th tick: the update runs, the row counter rewinds
    assert tr.global_update == 1 and int(tr._t_idx) == 0
    tr.tick()
    assert int(tr._t_idx) == 1
    env.close()


def test_single_frame_buffer_equals_the_stack_buffer(hip):
    """Two trainers on identical envs and seeds, one storing the whole 3-frame stack per tick, the other one frame per
    tick + the row indices: every minibatch the update would draw must be identical, restarts included."""
    from mrca.trainer import HParams, Stage1Trainer
    sc = S.stage1(num_worlds=4, robots_per_world=16, seed=8)
    envs = [hip.VecStageWorld(sc) for _ in range(2)]
    trs = [Stage1Trainer(e, hp=HParams(horizon=40, batch_size=512, epoch=1, single_frame_buffer=sf), seed=4)
           for e, sf in zip(envs, (False, True))]
    for tr in trs:
        tr.start()
    for _ in range(39):
        for tr in trs:
            tr.tick()
    torch.cuda.synchronize()
    a, b = trs[0].buffer, trs[1].buffer
    assert int(a.done[:39].sum()) > 0                                   # robots did restart inside the horizon
    assert torch.equal(a.action[:39], b.action[:39])
    full = b.obs_rows().materialise()
    assert torch.equal(full[:39], a.obs[:39])
    idx = torch.randperm(39 * sc.num_robots, device="cuda")[:500]
    assert torch.equal(b.obs_rows()[idx], a.obs_rows()[idx])
    keep = torch.rand(40 * sc.num_robots, device="cuda") < 0.7
    keep[39 * sc.num_robots:] = False
    sub = torch.arange(int(keep.sum()), device="cuda")[::7]
    assert torch.equal(b.obs_rows()[keep][sub], a.obs_rows()[keep][sub])
    for e in envs:
        e.close()


def test_library_stores_fill_the_buffer_exactly_as_the_torch_stores(hip):
    """Two trainers on identical envs and seeds, ticks replayed as hipGraphs: one buffer written by the env's library
    (mrca_rollout_store_state / _outcome: two launches per tick, the row counter moved on by the second), the other through
    PyTorch's index_copy_ / cat / where chain.  Every tensor of the buffer must be identical, restarts included; a counter
    outside the horizon stores nothing and is reported by env.check()."""
    from mrca.trainer import HParams, Stage1Trainer
    sc = S.stage1(num_worlds=4, robots_per_world=16, seed=8)
    envs = [hip.VecStageWorld(sc) for _ in range(2)]
    trs = [Stage1Trainer(e, hp=HParams(horizon=40, batch_size=512, epoch=1, graph_tick=True, rollout_fused=True), seed=4) for e in envs]
    assert trs[0].buffer.env_bound and trs[1].buffer.env_bound
    trs[1].buffer._env = None                                            # this one: the torch stores
    for tr in trs:
        tr.start()
    for _ in range(39):                # (one tick per replay on both sides: a replay of eight draws its noise differently)
        for tr in trs:
            tr.tick()
    torch.cuda.synchronize()
    a, b = trs[0].buffer, trs[1].buffer
    assert int(trs[0]._t_idx) == int(trs[1]._t_idx) == 39
    assert int(a.done[:39].sum()) > 0                                    # robots did restart inside the horizon
    for name in ("goal", "speed", "action", "logprob", "value", "reward", "done", "fidx"):
        assert torch.equal(getattr(a, name)[:39], getattr(b, name)[:39]), name
    assert torch.equal(a.frames[:41], b.frames[:41]) and torch.equal(a._cur, b._cur)
    assert torch.equal(a.obs_rows().materialise()[:39], b.obs_rows().materialise()[:39])
    # a row counter outside [0, horizon): nothing is stored, the env's status word says so
    before = a.value.clone()
    bad = torch.tensor([40], dtype=torch.int64, device="cuda")
    envs[0].rollout_store_state(a._rows, bad, a.action[0], a.logprob[0], a.value[0])
    torch.cuda.synchronize()
    assert torch.equal(a.value, before) and int(bad) == 40
    with pytest.raises(RuntimeError, match="tick counter"):
        envs[0].check()
    envs[0].check()                                                      # cleared by the failed check
    for e in envs:
        e.close()
