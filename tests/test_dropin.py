"""The drop-in boundary at the level the reference's scripts see it: StageWorld(beam, index,
num_env) + rospy + mpi4py, run as rank threads (mrca.spmd).

CPU (here): the facade is driven by the oracle backend (test-only injection) --
  * the UNCHANGED /root/reference/ppo_stage1.py runs 24 ranks through >= one PPO update
    (skipped where the reference checkout is absent, i.e. on the GPU box);
  * a stand-in script written against the same surface checks values.
GPU: the same stand-in script runs on the HIP backend and must reproduce the oracle-backed log."""
import glob
import json
import os
import tempfile

import numpy as np
import pytest
import torch

import util as U

REF = U.reference_dir() or "/nonexistent"     # a checkout ($MRCA_REFERENCE) or the staged archive (tools/stage_reference.sh)
MINI = os.path.join(U.ROOT, "tests", "dropin_script_mini.py")


def _run_mini(backend_factory, n=6):
    from mrca import spmd, stage_world
    stage_world.set_backend_factory(backend_factory)
    out = os.path.join(tempfile.mkdtemp(), "mini")
    os.environ["MINI_OUT"], os.environ["MINI_NUM_ENV"] = out, str(n)
    try:
        errs = spmd.run_script(MINI, n, max_ticks=200)
    finally:
        stage_world.set_backend_factory(None)
    assert not errs, errs
    logs = [json.load(open(f"{out}.{r}.json")) for r in range(n)]
    assert all(len(lg) > 10 for lg in logs)
    return logs


def test_mini_script_on_oracle_backend():
    logs = _run_mini(U.OracleBackend)
    # speed input = the command of the tick just executed (stageros.cpp:543-558)
    for r, lg in enumerate(logs):
        for ep, step, rew, term, res, pose, speed in lg:
            assert abs(speed[0] - 0.8) < 1e-6 or term
            assert res in ("0", "Reach Goal", "Crashed", "Time out")
    # first tick of the first episode: progress reward 2.5 * (d0 - d1) in (-0.2, 0.2], or a crash
    assert all(abs(lg[0][2]) <= 0.2001 or lg[0][4] == "Crashed" for lg in logs)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ppo_stage1.py")), reason="reference checkout absent")
def test_unchanged_ppo_stage1_runs_through_an_update(monkeypatch):
    from mrca import spmd, stage_world
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)   # no GPU in this container
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    stage_world.set_backend_factory(U.OracleBackend)
    tmp = tempfile.mkdtemp()
    try:
        errs = spmd.run_script(os.path.join(REF, "ppo_stage1.py"), 24, max_ticks=132, chdir=tmp)
    finally:
        stage_world.set_backend_factory(None)
    assert not errs, errs
    ppo_logs = glob.glob(os.path.join(tmp, "log", "*", "ppo.log"))
    assert ppo_logs, "model/ppo.py's logger did not write"
    lines = [ln for ln in open(ppo_logs[0]).read().splitlines() if ln.strip()]
    assert len(lines) >= 6                      # 2 epochs x 3 minibatches of 1024 (ppo_stage1.py:28-29)
    vals = np.array([[float(x) for x in ln.split(",")] for ln in lines[:6]])
    assert np.isfinite(vals).all()
    out_logs = glob.glob(os.path.join(tmp, "log", "*", "output.log"))
    assert out_logs


def _reference_logs(tmp):
    """The three log streams of a run (ppo_stage1.py:140-161, model/ppo.py:10-19) in an order-free form: the rank threads
    of one tick write their episode lines in whatever order they are scheduled, so output.log is split per env (each
    env's own sequence is deterministic) and cal.log -- bare episode returns -- is sorted; ppo.log has one writer."""
    import re

    def lines(name):
        hits = glob.glob(os.path.join(tmp, "log", "*", name))
        return [ln.rstrip("\n") for ln in open(hits[0])] if hits else []
    per_env = {}
    other = []
    for ln in lines("output.log"):
        msg = ln.split(" - INFO - ", 1)[-1]            # drop "%(asctime)s - %(levelname)s - "
        m = re.match(r"Env (\d+),", msg)
        (per_env.setdefault(int(m.group(1)), []) if m else other).append(msg)
    return {"output_per_env": per_env, "output_other": other, "cal": sorted(lines("cal.log")), "ppo": lines("ppo.log")}


def _detach_reference_loggers():
    """The scripts attach FileHandlers to module-level loggers; a second run in this process must not write into the
    first run's files."""
    import logging
    for name in ("mylogger", "loggercal", "loggerppo"):
        lg = logging.getLogger(name)
        for h in list(lg.handlers):
            h.close()
            lg.removeHandler(h)


class _Recording:
    """A backend of the SharedWorld protocol that forwards to the real one and keeps (a) a running SHA-256 over what the
    scripts can see of EVERY tick and (b) a host copy of the world right after tick ``last_tick``.  (The state after the
    runtime's shutdown is not comparable: ranks whose episode ended at the last tick may or may not get their
    ``reset_pose`` teleport in before they notice the shutdown.)"""
    FIELDS = ("pose", "goal", "reward", "done", "result", "first_result", "crashed", "scan", "obs", "speed", "speed_gt",
              "local_goal")

    def __init__(self, inner, last_tick):
        import hashlib
        self.inner, self.last_tick, self.ticks = inner, last_tick, 0
        self.sha = hashlib.sha256()
        self.snapshot = None

    def reset(self, mask, poses, goals):
        self.inner.reset(mask, poses, goals)

    def step(self, actions):
        self.inner.step(actions)
        self.ticks += 1
        if self.ticks <= self.last_tick:
            state = {k: np.array(self.inner.field(k)) for k in self.FIELDS}
            self.sha.update(np.ascontiguousarray(actions, np.float32).tobytes())
            for k in self.FIELDS:
                self.sha.update(np.ascontiguousarray(state[k]).tobytes())
            if self.ticks == self.last_tick:
                self.snapshot = state

    def field(self, name):
        return self.inner.field(name)


def _run_unchanged(script, nprocs, max_ticks, factory, seed, checkpoint=None, hold="0"):
    """One run of an UNCHANGED reference script as rank threads with its own model/*.py on cuda (real ``.cuda()`` calls).
    -> (logs, host state of the batched world after the last tick, digest of every tick)."""
    import shutil
    from mrca import spmd, stage_world
    made = []

    def recording(sc):
        made.append(_Recording(factory(sc), max_ticks))
        return made[-1]
    _detach_reference_loggers()
    os.environ["MRCA_HOLD_VELOCITY"] = hold
    stage_world.set_seed(seed)
    stage_world.set_backend_factory(recording)
    torch.manual_seed(seed)                # policy initialisation, action noise (cuda), minibatch sampler (cpu)
    np.random.seed(seed)
    torch.backends.cudnn.benchmark = False        # MIOpen: no per-run algorithm search, deterministic algorithms only --
    torch.backends.cudnn.deterministic = True     # two runs of the reference's OWN model must agree bit for bit
    tmp = tempfile.mkdtemp()
    if checkpoint:
        os.makedirs(os.path.join(tmp, "policy"))
        shutil.copy(checkpoint, os.path.join(tmp, "policy", "stage2.pth"))
    try:
        errs = spmd.run_script(os.path.join(REF, script), nprocs, max_ticks=max_ticks, chdir=tmp)
    finally:
        stage_world.set_backend_factory(None)
        stage_world.set_seed(0)
        os.environ.pop("MRCA_HOLD_VELOCITY", None)
        _detach_reference_loggers()
    assert not errs, errs
    assert made[-1].snapshot is not None, f"the run stopped after {made[-1].ticks} of {max_ticks} ticks"
    return _reference_logs(tmp), made[-1].snapshot, made[-1].sha.hexdigest()


_UNCHANGED = {   # script -> (rank threads, ticks: one full PPO update + a few ticks acting on the UPDATED policy)
    "ppo_stage1.py": (24, 134),       # HORIZON 128 (ppo_stage1.py:25)
    "ppo_stage2.py": (44, 134),       # HORIZON 128 (ppo_stage2.py:25), group-synchronous episodes, filtered update
    "circle_test.py": (50, 160),
}


def _need_the_reference_scripts():
    """The GPU legs below ARE the north-star sentence ("ppo_stage1.py drops in unchanged"): where neither a checkout
    ($MRCA_REFERENCE, /root/reference) nor the staged archive (tests/_reference.tgz, packed by tools/stage_reference.sh /
    __graft_entry__.build() wherever a checkout exists; git-ignored, travels with the working tree) is present they FAIL --
    a clean clone must not lose them to a silent skip."""
    if not os.path.exists(os.path.join(REF, "ppo_stage1.py")):
        pytest.fail("the reference's unchanged scripts are not available on this box: no checkout ($MRCA_REFERENCE) and no "
                    "tests/_reference.tgz -- run tools/stage_reference.sh (or __graft_entry__.build()) where /root/reference "
                    "exists and ship the working tree, archive included")


@pytest.mark.gpu
@pytest.mark.parametrize("script", sorted(_UNCHANGED))
def test_unchanged_script_on_the_hip_backend_equals_the_oracle_backend(script):
    """The literal north-star sentence on the MI355X: the UNCHANGED ppo_stage1.py / ppo_stage2.py / circle_test.py (their own
    model/ppo.py and model/net.py, real .cuda() policy) run twice with the same seeds -- the batched world behind the
    StageWorld surface once the C oracle, once HipBackend (libmrca_env.so through the C ABI).  The env is bit-exact, so
    everything the scripts see is identical: output.log (per env), cal.log and ppo.log must be EQUAL, and so must the
    final state of the world, field by field, bit by bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _need_the_reference_scripts()
    import __graft_entry__ as g
    g.build()
    from mrca import stage_world
    nprocs, ticks = _UNCHANGED[script]
    ckpt = os.path.join(U.ROOT, "rl-collision-avoidance_amd", "mrca", "data", "policy_r02_stage2_circles.pth") \
        if script == "circle_test.py" else None
    ref_logs, ref_state, ref_sha = _run_unchanged(script, nprocs, ticks, U.COracleBackend, 11, ckpt)
    got_logs, got_state, got_sha = _run_unchanged(script, nprocs, ticks, stage_world.HipBackend, 11, ckpt)
    problems = _compare_runs(ref_logs, ref_state, ref_sha, got_logs, got_state, got_sha)
    if script != "circle_test.py":         # (circle_test.py writes no log: its evidence is the world's state and digest)
        assert ref_logs["ppo"], "the run did not reach a PPO update"
        assert sum(len(v) for v in ref_logs["output_per_env"].values()) >= nprocs // 2, "too few episodes ended"
    assert not problems, f"{script}: " + "; ".join(problems)


def _compare_runs(ref_logs, ref_state, ref_sha, got_logs, got_state, got_sha):
    """Everything that differs between two runs, in one report.  Episode lines: an episode that ends AT the last tick is
    logged by its rank thread only if the thread gets there before it notices the shutdown, so per env the shorter
    sequence must be a PREFIX of the longer one and at most one line shorter; cal.log likewise (a multiset)."""
    from collections import Counter
    problems = []
    for k in ref_state:
        a, b = ref_state[k], got_state[k]
        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
        if not same.all():
            problems.append(f"{k} after the last tick differs at {int((~same).sum())} of {same.size} entries")
    if ref_sha != got_sha:
        problems.append("the digest over actions + state of every tick differs")
    pa, pb = ref_logs["output_per_env"], got_logs["output_per_env"]
    for e in sorted(set(pa) | set(pb)):
        x, y = pa.get(e, []), pb.get(e, [])
        n = min(len(x), len(y))
        if x[:n] != y[:n] or abs(len(x) - len(y)) > 1:
            problems.append(f"output.log of env {e}: {x[max(0, n - 2):n + 1]} vs {y[max(0, n - 2):n + 1]}")
            break
    if ref_logs["output_other"] != got_logs["output_other"]:
        problems.append(f"output.log banner lines: {ref_logs['output_other']} vs {got_logs['output_other']}")
    ca, cb = Counter(ref_logs["cal"]), Counter(got_logs["cal"])
    if sum(((ca - cb) + (cb - ca)).values()) > len(pa) or not (ca - cb == Counter() or cb - ca == Counter()):
        problems.append(f"cal.log: {sum((ca - cb).values())} / {sum((cb - ca).values())} unmatched episode returns")
    if ref_logs["ppo"] != got_logs["ppo"]:
        a, b = ref_logs["ppo"], got_logs["ppo"]
        first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
        problems.append(f"ppo.log: {len(a)} vs {len(b)} lines, first difference at line {first}: {a[first:first + 1]} vs "
                        f"{b[first:first + 1]}")
    return problems


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ppo_stage1.py")), reason="reference checkout absent")
def test_unchanged_ppo_stage1_is_reproducible_run_to_run(monkeypatch):
    """The SPMD runtime's deterministic schedule (mrca/spmd.py: one rank at a time, lowest ready rank next): two runs of the
    UNCHANGED ppo_stage1.py with the same seeds on the SAME backend must agree in every tick's actions and state, in every
    log line and in the PPO losses.  (With rank threads left to the OS scheduler they differed from the second tick on:
    whichever rank's reset_pose teleport came first re-cast its lidar against a different world.)  This is what makes the
    oracle-vs-HIP comparison of the GPU leg meaningful."""
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)   # no GPU in this container
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    torch.set_num_threads(1)          # (a fixed reduction order in the CPU GEMMs of the reference's policy)
    try:
        a = _run_unchanged("ppo_stage1.py", 24, 40, U.COracleBackend, 5)
        b = _run_unchanged("ppo_stage1.py", 24, 40, U.COracleBackend, 5)
    finally:
        torch.set_num_threads(max(1, os.cpu_count() or 1))
    assert not _compare_runs(*a, *b)
    assert a[2] == b[2] and sum(len(v) for v in a[0]["output_per_env"].values()) > 20


@pytest.mark.gpu
def test_mini_script_hip_backend_matches_oracle_backend():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as g
    g.build()
    ref = _run_mini(U.OracleBackend)
    got = _run_mini(None)       # product default: HipBackend
    # every facade owns a generator seeded from (base seed, rank), so the two runs draw the same poses / goals
    # whatever the thread schedule: the HIP-backed log must equal the oracle-backed one EXACTLY -- episode,
    # step, reward, terminal, result string, ground-truth pose and speed of every rank at every step
    assert got == ref
    assert any(row[3] for lg in got for row in lg), "no terminal event in the run: the comparison is too easy"


def test_get_reward_and_terminate_honours_the_callers_step_counter():
    """stage_world1.py:206: `if t > 150` uses the CALLER's t."""
    from mrca import spmd, stage_world  # noqa: F401
    stage_world.set_backend_factory(U.OracleBackend)
    try:
        env = stage_world.Stage1World(512, index=0, num_env=2)
        env.reset_pose()
        env.generate_goal_point()
        env.control_vel([0.0, 0.0])
        env.world.tick()
        r0, term0, res0 = env.get_reward_and_terminate(1)
        r1, term1, res1 = env.get_reward_and_terminate(151)
        assert (term0, res0) == (False, 0) and (term1, res1) == (True, "Time out") and r0 == r1
    finally:
        stage_world.set_backend_factory(None)


@pytest.mark.parametrize("hold", [False, True])
def test_a_rank_that_stops_commanding(monkeypatch, hold):
    """ppo_stage2.py:72-74: a finished rank sends no cmd_vel.  Default: its robot idles.  MRCA_HOLD_VELOCITY=1: Stage's
    SetSpeed persistence (stageros.cpp:272-280) -- the robot drives on at its last command and get_self_speed shows it,
    also right after reset_pose's teleport (stage_world1.py:106-108: the odom twist)."""
    from mrca import stage_world
    monkeypatch.setenv("MRCA_HOLD_VELOCITY", "1" if hold else "0")
    stage_world.set_backend_factory(U.OracleBackend)
    try:
        a, b = (stage_world.Stage1World(512, index=i, num_env=2) for i in range(2))
        a.control_pose([-2.0, 0.0, 0.0])
        b.control_pose([2.0, 3.0, 0.0])
        a.control_vel([0.5, 0.2])
        b.control_vel([0.5, 0.0])
        a.world.tick()
        x1 = a.get_self_stateGT()[0]
        assert abs(x1 - (-1.95)) < 1e-6
        for _ in range(4):
            b.control_vel([0.5, 0.0])        # rank 0 says nothing any more
            a.world.tick()
        x5, sp = a.get_self_stateGT()[0], a.get_self_speed()
        if hold:
            assert x5 > x1 + 0.19 and np.allclose(sp, [0.5, 0.2])
        else:
            assert x5 == x1 and np.allclose(sp, [0.0, 0.0])
        a.control_pose([-4.0, -4.0, 1.0])
        assert np.allclose(a.get_self_speed(), [0.5, 0.2] if hold else [0.0, 0.0])
    finally:
        stage_world.set_backend_factory(None)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ppo_stage2.py")), reason="reference checkout absent")
@pytest.mark.parametrize("hold", ["0", "1"])
def test_unchanged_ppo_stage2_runs(monkeypatch, hold):
    """44 ranks of the UNCHANGED ppo_stage2.py (group-synchronous episodes, liveflag, bcast,
    get_group_terminal with the py2 `reduce`) on the drop-ins -- with the default idle rule for ranks that stop
    commanding and with Stage's velocity persistence (MRCA_HOLD_VELOCITY=1)."""
    from mrca import spmd, stage_world
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setenv("MRCA_HOLD_VELOCITY", hold)
    stage_world.set_backend_factory(U.OracleBackend)
    tmp = tempfile.mkdtemp()
    try:
        errs = spmd.run_script(os.path.join(REF, "ppo_stage2.py"), 44, max_ticks=30, chdir=tmp)
    finally:
        stage_world.set_backend_factory(None)
    assert not errs, errs


def test_get_group_terminal_matches_reference_rule():
    from mrca import ppo
    t = [False] * 44
    for i in range(6, 10):
        t[i] = True
    assert ppo.get_group_terminal(t, 7) and not ppo.get_group_terminal(t, 3) and not ppo.get_group_terminal(t, 12)
    t[5] = True
    assert not ppo.get_group_terminal(t, 0)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "circle_test.py")), reason="reference checkout absent")
def test_unchanged_circle_test_with_our_checkpoint(monkeypatch):
    """50 ranks of the UNCHANGED circle_test.py; it loads policy/stage2.pth (missing from the reference
    checkout), so a checkpoint SAVED BY THIS REPO's CNNPolicy is put there: the reference's own
    CNNPolicy.load_state_dict must accept it (checkpoint compatibility in the reference -> direction)."""
    from mrca import spmd, stage_world
    from mrca.net import CNNPolicy
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "policy"))
    torch.manual_seed(5)
    torch.save(CNNPolicy(3, 2).state_dict(), os.path.join(tmp, "policy", "stage2.pth"))
    stage_world.set_backend_factory(U.OracleBackend)
    try:
        errs = spmd.run_script(os.path.join(REF, "circle_test.py"), 50, max_ticks=12, chdir=tmp)
    finally:
        stage_world.set_backend_factory(None)
    assert not errs, errs


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "circle_test.py")), reason="reference checkout absent")
def test_unchanged_circle_test_with_the_trained_checkpoint_reaches_every_goal(monkeypatch):
    """The reference's deliverable end to end: its UNCHANGED circle_test.py (50 ranks, its own CNNPolicy and
    generate_action_no_sampling, model/ppo.py:84-107) loads the checkpoint this repo trained
    (mrca/data/policy_r02_stage2_circles.pth as policy/stage2.pth) and drives the 50 robots of the drop-in circle world:
    every robot's first terminal event must be "Reach Goal"."""
    import shutil
    from mrca import spmd, stage_world
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "policy"))
    shutil.copy(os.path.join(U.ROOT, "rl-collision-avoidance_amd", "mrca", "data", "policy_r02_stage2_circles.pth"),
                os.path.join(tmp, "policy", "stage2.pth"))
    made = []

    def factory(sc):
        made.append(U.COracleBackend(sc))
        return made[-1]
    stage_world.set_backend_factory(factory)
    try:
        errs = spmd.run_script(os.path.join(REF, "circle_test.py"), 50, max_ticks=700, chdir=tmp)
    finally:
        stage_world.set_backend_factory(None)
    assert not errs, errs
    first = np.asarray(made[-1].env.first_result)
    assert (first == 1).mean() >= 0.9, np.bincount(first, minlength=4)       # measured: 50 of 50 reach their goals
