"""Host-side learner logic that needs no GPU: the trainer on the C oracle env (tests/cpu_train.CpuEnv) --
one-frame-per-tick rollout buffer == full-stack buffer, mixed-scenario env concatenation, KL-adaptive learning rate,
floor on the policy's log std."""
import numpy as np
import pytest
import torch

import cpu_train as ct          # tests/cpu_train.py: CpuEnv + the torch GAE stand-in for the HIP kernel
import util as U  # noqa: F401
from util import S


@pytest.fixture(autouse=True)
def _cpu_gae(monkeypatch):
    """The HIP GAE kernel's binding -> the torch loop, for the duration of one test only."""
    from mrca import vec_env
    monkeypatch.setattr(vec_env, "gae", ct._gae_cpu)


def _trainers(sc, horizon, **kw):
    from mrca.trainer import HParams, Stage1Trainer
    out = []
    for single in (False, True):
        env = ct.CpuEnv(sc)
        tr = Stage1Trainer(env, hp=HParams(horizon=horizon, batch_size=256, epoch=1, single_frame_buffer=single, **kw),
                           seed=4)
        tr.start()
        out.append(tr)
    return out


def test_single_frame_buffer_equals_the_stack_buffer_on_the_oracle_env():
    torch.set_num_threads(2)
    sc = S.stage1(num_worlds=2, robots_per_world=12, seed=8)
    a, b = _trainers(sc, horizon=60)
    for _ in range(59):
        a.tick()
        b.tick()
    A, B = a.buffer, b.buffer
    assert int(A.done[:59].sum()) > 0                      # robots restarted inside the horizon
    assert torch.equal(A.action[:59], B.action[:59])
    assert torch.equal(B.obs_rows().materialise()[:59], A.obs[:59])
    idx = torch.randperm(59 * sc.num_robots)[:300]
    assert torch.equal(B.obs_rows()[idx], A.obs_rows()[idx])
    keep = torch.rand(60 * sc.num_robots) < 0.7            # the Stage-2 filter path: boolean mask, then indices
    keep[59 * sc.num_robots:] = False
    sub = torch.arange(int(keep.sum()))[::5]
    assert torch.equal(B.obs_rows()[keep][sub], A.obs_rows()[keep][sub])
    assert B.frames.numel() * 3 < A.obs.numel() * 1.2      # a third of the memory (plus two rows)
    a.tick()                                               # the update runs; the next horizon starts from row 0 again
    b.tick()
    a.tick()
    b.tick()
    assert a.global_update == b.global_update == 1


def test_concat_env_steps_every_part_and_keeps_the_order():
    from mrca.multi_env import ConcatEnv
    parts = [ct.CpuEnv(S.stage2(num_worlds=1, seed=1)), ct.CpuEnv(S.circle_n(10, 8.0, train=True))]
    ref = [ct.CpuEnv(S.stage2(num_worlds=1, seed=1)), ct.CpuEnv(S.circle_n(10, 8.0, train=True))]
    env = ConcatEnv(parts)
    env.reset()
    for e in ref:
        e.reset()
    assert env.N == 54 and env.bounds == [(0, 44), (44, 54)]
    g = torch.Generator().manual_seed(0)
    for _ in range(5):
        act = torch.stack([torch.rand(54, generator=g), torch.rand(54, generator=g) * 2 - 1], 1)
        env.step(act)
        ref[0].step(act[:44].contiguous())
        ref[1].step(act[44:].contiguous())
    for k in ("obs", "local_goal", "reward", "done", "live", "fresh"):
        assert torch.equal(getattr(env, k)[:44], getattr(ref[0], k)), k
        assert torch.equal(getattr(env, k)[44:], getattr(ref[1], k)), k


def test_kl_adaptive_lr_and_logstd_floor():
    from mrca import ppo
    from mrca.net import CNNPolicy
    torch.manual_seed(0)
    pol = CNNPolicy(3, 2)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
    ctl = ppo.KLAdaptiveLR(target=0.01, lr_min=1e-6, lr_max=3e-4, stop_factor=2.0)
    ctl.step(opt, torch.tensor(0.001), 1, None)            # KL well below target: lr up, capped
    assert abs(opt.param_groups[0]["lr"] - 1.5e-4) < 1e-12
    ctl.step(opt, torch.tensor(0.001), 1, None)
    ctl.step(opt, torch.tensor(0.001), 1, None)
    assert opt.param_groups[0]["lr"] == 3e-4
    ctl.step(opt, torch.tensor(0.05), 1, None)             # above 2 x target: lr down
    assert abs(opt.param_groups[0]["lr"] - 2e-4) < 1e-12
    assert ctl.should_stop(torch.tensor(0.03), None) and not ctl.should_stop(torch.tensor(0.015), None)
    # the floor: an update that would push logstd down leaves it at the floor
    T, N = 4, 8
    g = torch.Generator().manual_seed(1)
    mem = (torch.rand(T, N, 3, 512, generator=g) - 0.5, torch.rand(T, N, 2, generator=g), torch.rand(T, N, 2, generator=g),
           torch.rand(T, N, 2, generator=g), torch.full((T, N, 1), -2.0), torch.randn(T, N, generator=g), None, None,
           torch.randn(T, N, generator=g))
    with torch.no_grad():
        pol.logstd.fill_(-1.2)
    big = torch.optim.SGD(pol.parameters(), lr=1.0)
    ppo.ppo_update_stage1(policy=pol, optimizer=big, batch_size=16, memory=mem, epoch=2, coeff_entropy=5e-4,
                          clip_value=0.1, num_step=T, num_env=N, frames=3, obs_size=512, act_size=2, logstd_min=-1.2)
    assert float(pol.logstd.detach().min()) >= -1.2


def test_rollout_cache_follows_the_parameters_whoever_changes_them():
    """The fused rollout path reads tower-major COPIES of the parameters.  They are rebuilt lazily whenever any parameter's
    in-place version, storage or device differs from what the copies were made from -- an optimiser step outside the
    trainer, a ``copy_`` (broadcast), ``load_state_dict`` -- and IN PLACE (a captured tick keeps reading the same addresses)."""
    import torch
    from mrca.net import CNNPolicy
    pol = CNNPolicy(3, 2)
    rc = pol._rollout_cache()
    addr = rc["fc1_w"].data_ptr()
    assert torch.equal(rc["w1"][0], pol.act_fea_cv1.weight) and pol._rollout_cache() is rc
    with torch.no_grad():
        pol.act_fea_cv1.weight.add_(1.0)                       # e.g. an optimiser step the trainer did not make
        pol.crt_fc1.bias.copy_(torch.full_like(pol.crt_fc1.bias, 3.0))       # e.g. a parameter broadcast
    assert not torch.equal(rc["w1"][0], pol.act_fea_cv1.weight)                # stale until somebody asks
    rc2 = pol._rollout_cache()
    assert rc2 is rc and rc["fc1_w"].data_ptr() == addr                        # refreshed in place
    assert torch.equal(rc["w1"][0], pol.act_fea_cv1.weight) and float(rc["fc1_b"][1, 0, 0]) == 3.0
    sd = {k: v.clone() * 0.5 for k, v in pol.state_dict().items()}
    pol.load_state_dict(sd)
    assert torch.equal(pol._rollout_cache()["w2"][1], pol.crt_fea_cv2.weight)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-2)
    pol.logstd.grad = torch.ones_like(pol.logstd)
    opt.step()
    assert torch.equal(pol._rollout_cache()["logstd"], pol.logstd.detach().reshape(-1))


def test_ring_head_travels_opaquely_and_normalize_scans_is_the_ieee_quotient():
    """policy_ops.RingHead (what VecStageWorld.policy_obs hands out since ABI 4: head slots + "the ring holds RAW ranges")
    slices like a tensor; normalize_scans on the host is |x| / 6 - 0.5 with the correctly rounded quotient (the sign bit of a
    ring entry says what the beam hit), i.e. the kernel's norm_obs."""
    import numpy as np
    import torch
    from mrca import policy_ops as P
    h = P.RingHead(torch.tensor([0, 1, 2, 1], dtype=torch.uint8), raw=True)
    sl = h[1:3]
    assert isinstance(sl, P.RingHead) and sl.raw and sl.slots.tolist() == [1, 2]
    assert P.unwrap_head(h) == (h.slots, True) and P.unwrap_head(h.slots) == (h.slots, False) and P.unwrap_head(None) == (None, False)
    x = torch.tensor([0.0, -0.0, 1.0, -2.5, 5.9999995, -6.0, 3.3333333], dtype=torch.float32)
    want = np.abs(x.numpy()) / np.float32(6.0) - np.float32(0.5)
    assert np.array_equal(P.normalize_scans(x).numpy(), want) and want.dtype == np.float32


def test_recorded_gemm_choices_are_a_no_op_without_a_gpu(monkeypatch):
    import torch
    from mrca import gemm_tuning
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    assert gemm_tuning.use_recorded_choices() is False
    rows = [ln.split(",") for ln in open(gemm_tuning.DEFAULT_FILE) if ln.startswith("Gemm")]
    assert len(rows) >= 10 and all(len(r) == 4 and float(r[3]) > 0 for r in rows)


def test_one_frame_buffer_row_indices_follow_the_deque_rule():
    """The rule the one-frame buffer's row indices follow -- and that csrc/mrca_rollout_store.hip restates on the device
    (tests/test_gpu_trainer.py holds the two bit-identical): the newest frame of tick t lives in row t + F - 1; tick 0 of a
    horizon sees rows 0 .. F-1; a robot that restarted sees F times its fresh scan (ppo_stage1.py:59-60: the deque is refilled
    with the first observation); everybody else drops its oldest row and appends the new one (ppo_stage1.py:87-89)."""
    from mrca import ppo
    T, N, F, B = 9, 5, 3, 8
    rng = np.random.default_rng(3)
    buf = ppo.RolloutBuffer(T, N, F, B, torch.device("cpu"), single_frame=True)
    first = torch.arange(N * F * B, dtype=torch.float32).view(N, F, B)
    buf.begin_horizon(first)
    want = np.zeros((T, N, F), dtype=np.int64)
    cur = np.tile(np.arange(F), (N, 1))
    fresh_log = []
    for t in range(T):
        fresh = rng.random(N) < 0.3
        fresh_log.append(fresh)
        row = t + F - 1
        if t == 0:
            cur = np.tile(np.arange(F), (N, 1))
        else:
            shifted = np.concatenate([cur[:, 1:], np.full((N, 1), row)], axis=1)
            cur = np.where(fresh[:, None], row, shifted)
        want[t] = cur
        newest = torch.full((N, B), float(100 + t))
        z2, z1 = torch.zeros(N, 2), torch.zeros(N, 1)
        buf.store_state_at(torch.tensor([t]), None, z2, z2, z2, z1, z1, torch.from_numpy(fresh.astype(np.uint8)), newest=newest)
    assert np.array_equal(buf.fidx.numpy(), want)
    assert torch.equal(buf.frames[:F - 1], first[:, :F - 1].transpose(0, 1))          # the older frames of the first tick's stack
    assert all(float(buf.frames[t + F - 1, 0, 0]) == 100 + t for t in range(T))
    stacks = buf.obs_rows().materialise()                                            # [T, N, F, B]
    t, n = 4, int(np.argmax(fresh_log[4])) if fresh_log[4].any() else 0
    if fresh_log[4].any():
        assert float(stacks[t, n, 0, 0]) == float(stacks[t, n, F - 1, 0]) == 104.0    # restarted: F copies of its fresh scan


def test_frame_rows_lazy_index_names_the_rows_the_gather_reads():
    """FrameRows.lazy (the fused update, mrca/ppo.py): an integer index returns a FrameTable -- the flattened frame store and, per
    sample, the rows of its three frames -- whose gather() is the [mb, 3, 512] tensor the eager index returns, also behind a
    boolean keep-mask (Stage-2 filtering)."""
    import torch
    from mrca import ppo
    T, N = 5, 7
    g = torch.Generator().manual_seed(0)
    frames = torch.rand(T + 2, N, 512, generator=g)
    fidx = torch.randint(0, T + 2, (T, N, 3), generator=g)
    eager, lazy = ppo.FrameRows(frames, fidx), ppo.FrameRows(frames, fidx, lazy=True)
    index = torch.randperm(T * N, generator=g)[:13]
    table = lazy[index]
    assert table.rows.dtype == torch.int32 and tuple(table.rows.shape) == (13, 3) and tuple(table.shape) == (13, 3, 512)
    assert torch.equal(table.gather(), eager[index])
    keep = torch.rand(T * N, generator=g) > 0.4
    sub_e, sub_l = eager[keep], lazy[keep]
    assert sub_l.lazy and not sub_e.lazy
    idx2 = torch.arange(int(keep.sum()))[::2]
    assert torch.equal(sub_l[idx2].gather(), sub_e[idx2])
