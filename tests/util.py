"""Shared test helpers: scenario -> oracle config, and the host emulation harness
(tests/host_emul) that runs the product's per-lane arithmetic on the CPU."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-collision-avoidance_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import mrca_oracle as O  # noqa: E402
from mrca import scenario as S  # noqa: E402

_ref_dir = None


def reference_dir():
    """A directory holding the reference's unchanged Python scripts, or None: $MRCA_REFERENCE / /root/reference where a
    checkout exists (this container), otherwise the git-ignored archive tools/stage_reference.sh packed
    (tests/_reference.tgz -- it travels to the GPU box with the working tree) unpacked into a temporary directory."""
    global _ref_dir
    if _ref_dir is None:
        cand = os.environ.get("MRCA_REFERENCE", "/root/reference")
        tgz = os.path.join(ROOT, "tests", "_reference.tgz")
        if os.path.exists(os.path.join(cand, "ppo_stage1.py")):
            _ref_dir = cand
        elif os.path.exists(tgz):
            import tarfile
            import tempfile
            d = tempfile.mkdtemp(prefix="mrca_reference_")
            with tarfile.open(tgz) as t:
                t.extractall(d)
            _ref_dir = d
        else:
            _ref_dir = ""
    return _ref_dir or None


STATE_FIELDS = ["pose", "speed", "speed_gt", "goal", "init_pose", "scan", "obs", "local_goal", "reward", "prev_dist",
                "done", "result", "first_result", "crashed", "live", "t", "episode"]


def oracle_env(sc, dtype=np.float32, first_world=0, num_worlds=None):
    """Oracle for worlds [first_world, first_world+num_worlds) of scenario ``sc`` (all by default)."""
    gm = O.GridMap(sc.grid.bits, sc.grid.width, sc.grid.height, sc.grid.cell, sc.grid.x0, sc.grid.y0)
    cfg = O.OracleConfig(sc.num_worlds if num_worlds is None else num_worlds, sc.robots_per_world, gm,
                         timeout=sc.timeout, w_thresh=sc.w_thresh,
                         pre_dist_zero=sc.pre_dist_zero, auto_reset=sc.auto_reset, seed=sc.seed,
                         reset_mode=sc.reset_mode, init_table=sc.init_table, goal_table=sc.goal_table,
                         group_id=sc.group_id, beams=sc.beams, frames=sc.frames, first_world=first_world,
                         collision_raster=getattr(sc, "collision_raster", 0.0),
                         hold_velocity=getattr(sc, "hold_velocity", False))
    cfg.goal_mode = np.asarray(sc.goal_mode, np.int32)
    return O.OracleEnv(cfg, dtype)


def random_actions(rng, n):
    """v~U(0,1), omega~U(-1,1): the clipped action range (ppo_stage1.py:170)."""
    return np.stack([rng.uniform(0, 1, n), rng.uniform(-1, 1, n)], 1).astype(np.float32)


def small_grid(cell=0.05, size=20.0, ring_radius=None, blocks=()):
    """Synthetic square arena with boundary walls, optional circular wall and box obstacles."""
    n = int(round(size / cell))
    occ = np.zeros((n, n), bool)
    occ[0, :] = occ[-1, :] = occ[:, 0] = occ[:, -1] = True
    yy, xx = np.mgrid[0:n, 0:n]
    cx = (xx + 0.5) * cell - size / 2
    cy = (yy + 0.5) * cell - size / 2
    if ring_radius:
        r = np.sqrt(cx * cx + cy * cy)
        occ |= (r >= ring_radius) & (r <= ring_radius + 3 * cell)
    for (bx0, by0, bx1, by1) in blocks:
        occ |= (cx >= bx0) & (cx <= bx1) & (cy >= by0) & (cy <= by1)
    return S.GridData.from_dense(occ, cell, -size / 2, -size / 2)


# ------------------------------------------------------------------------------------------------
class _EmulEnvStruct(C.Structure):
    _fields_ = ([(k, C.c_int32) for k in ("N", "R", "W", "B", "F")] +
                [(k, C.c_void_p) for k in ("pose", "speed", "speed_gt", "goal", "init_pose", "scan", "obs",
                                           "local_goal", "reward", "prev_dist", "done", "result", "first_result",
                                           "crashed", "live", "fresh", "t", "episode", "reset_mode", "goal_mode",
                                           "group_id", "init_table", "goal_table", "beam_cos", "beam_sin",
                                           "map_bits")] +
                [("x0", C.c_float), ("y0", C.c_float), ("cell", C.c_float)] +
                [(k, C.c_int32) for k in ("width", "height", "wpr", "timeout")] +
                [("w_thresh", C.c_float)] +
                [(k, C.c_int32) for k in ("pre_dist_zero", "auto_reset", "num_groups")] +
                [("key0", C.c_uint32), ("key1", C.c_uint32), ("hold_velocity", C.c_int32)])


_emul_lib = None


def emul_lib():
    """Build (once) and load tests/host_emul/libmrca_emul.so with g++ -ffp-contract=off."""
    global _emul_lib
    if _emul_lib is not None:
        return _emul_lib
    src = os.path.join(ROOT, "tests", "host_emul", "emul.cpp")
    hdrs = [os.path.join(ROOT, "rl-collision-avoidance_amd", "csrc", h)
            for h in ("mrca_device.h", "mrca_host.h", "mrca_policy_layout.h")]
    out = os.path.join(ROOT, "tests", "host_emul", "libmrca_emul.so")
    if (not os.path.exists(out)) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                               "-Wno-unknown-pragmas", src, "-o", out])
    _emul_lib = C.CDLL(out)
    return _emul_lib


class EmulEnv:
    """numpy-backed state + the product's mrca_device.h arithmetic driven by host loops."""

    def __init__(self, sc):
        self.lib = emul_lib()
        self.sc = sc
        N, R, B, F = sc.num_robots, sc.robots_per_world, sc.beams, sc.frames
        self.N = N
        f32, u8, i32 = np.float32, np.uint8, np.int32
        self.pose = np.ascontiguousarray(np.tile(np.asarray(sc.init_table, f32).reshape(-1, 3), (sc.num_worlds, 1)), f32)
        self.speed = np.zeros((N, 2), f32)
        self.speed_gt = np.zeros((N, 2), f32)
        self.goal = np.zeros((N, 2), f32)
        self.init_pose = np.zeros((N, 3), f32)
        self.scan = np.zeros((N, B), f32)
        self.obs = np.zeros((N, F, B), f32)
        self.local_goal = np.zeros((N, 2), f32)
        self.reward = np.zeros(N, f32)
        self.prev_dist = np.zeros(N, f32)
        self.done = np.zeros(N, u8)
        self.result = np.zeros(N, u8)
        self.first_result = np.zeros(N, u8)
        self.crashed = np.zeros(N, u8)
        self.live = np.ones(N, u8)
        self.fresh = np.zeros(N, u8)
        self.t = np.ones(N, i32)
        self.episode = np.zeros(N, i32)
        self._reset_mode = np.ascontiguousarray(sc.reset_mode, i32)
        self._goal_mode = np.ascontiguousarray(sc.goal_mode, i32)
        self._group_id = np.ascontiguousarray(sc.group_id, i32)
        self._init_table = np.ascontiguousarray(sc.init_table, f32)
        self._goal_table = np.ascontiguousarray(sc.goal_table, f32)
        b = -np.pi / 2 + np.arange(B, dtype=np.float64) * (np.pi / (B - 1))
        self._bcos = np.cos(b).astype(f32)
        self._bsin = np.sin(b).astype(f32)
        self._bits = np.ascontiguousarray(sc.grid.bits, np.uint32)
        st = _EmulEnvStruct()
        st.N, st.R, st.W, st.B, st.F = N, R, sc.num_worlds, B, F
        for k in ("pose", "speed", "speed_gt", "goal", "init_pose", "scan", "obs", "local_goal", "reward", "prev_dist",
                  "done", "result", "first_result", "crashed", "live", "fresh", "t", "episode"):
            setattr(st, k, getattr(self, k).ctypes.data)
        st.reset_mode, st.goal_mode, st.group_id = (self._reset_mode.ctypes.data, self._goal_mode.ctypes.data,
                                                    self._group_id.ctypes.data)
        st.init_table, st.goal_table = self._init_table.ctypes.data, self._goal_table.ctypes.data
        st.beam_cos, st.beam_sin, st.map_bits = self._bcos.ctypes.data, self._bsin.ctypes.data, self._bits.ctypes.data
        st.x0, st.y0, st.cell = sc.grid.x0, sc.grid.y0, sc.grid.cell
        st.width, st.height, st.wpr = sc.grid.width, sc.grid.height, sc.grid.words_per_row
        st.timeout, st.w_thresh = sc.timeout, sc.w_thresh
        st.pre_dist_zero, st.auto_reset = int(sc.pre_dist_zero), sc.auto_reset
        st.num_groups = int(self._group_id.max()) + 1
        st.key0, st.key1 = sc.seed & 0xFFFFFFFF, (sc.seed >> 32) & 0xFFFFFFFF
        st.hold_velocity = int(bool(getattr(sc, "hold_velocity", False)))
        self._st = st

    def reset(self, mask=None, poses=None, goals=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p = None if poses is None else np.ascontiguousarray(poses, np.float32)
        g = None if goals is None else np.ascontiguousarray(goals, np.float32)
        self.lib.emul_reset(C.byref(self._st), C.c_void_p(m.ctypes.data) if m is not None else None,
                            C.c_void_p(p.ctypes.data) if p is not None else None,
                            C.c_void_p(g.ctypes.data) if g is not None else None)

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32)
        self.lib.emul_step(C.byref(self._st), C.c_void_p(a.ctypes.data))


def assert_state_equal(a, b, fields=STATE_FIELDS, what=""):
    """Bit-exact comparison of two env states (numpy arrays or objects exposing them)."""
    for k in fields:
        x = np.asarray(getattr(a, k))
        y = np.asarray(getattr(b, k))
        if x.dtype.kind == "f":
            same = (x.view(np.uint32) == y.astype(np.float32).view(np.uint32)) if x.dtype == np.float32 else (x == y)
        else:
            same = x.astype(np.int64) == y.astype(np.int64)
        if not same.all():
            bad = np.argwhere(~same)
            i = tuple(bad[0])
            raise AssertionError(f"{what}: field {k} differs at {len(bad)} of {same.size} entries; first {i}: "
                                 f"{x[i]!r} vs {y[i]!r}")


def assert_hits_equal(env, ora, what=""):
    """What every beam hit: the device's flag (its MRCA_F_HIT_BITS plane: another robot) against the oracle's."""
    got = env.hit_robot.cpu().numpy()
    want = np.asarray(ora.hit_robot).astype(bool)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError(f"{what}: hit_robot differs at {len(bad)} of {got.size} beams; first {tuple(bad[0])}")


class HostView:
    """Host copy of a VecStageWorld's state (optionally a slice of robots)."""

    def __init__(self, env, lo=0, hi=None, fields=STATE_FIELDS):
        for k in fields:
            setattr(self, k, getattr(env, k)[lo:hi].cpu().numpy())


class OracleBackend:
    """Checker backend for the per-index facade (tests only): the NumPy oracle behind the
    SharedWorld backend protocol of mrca/stage_world.py."""

    def __init__(self, sc, dtype=np.float32):
        self.env = oracle_env(sc, dtype)

    def reset(self, mask, poses, goals):
        self.env.reset(mask, poses, goals)

    def step(self, actions):
        self.env.step(actions)

    def field(self, name):
        return np.asarray(getattr(self.env, name))


# ------------------------------------------------------------------------------------------------
class _OcEnvStruct(C.Structure):
    _fields_ = _EmulEnvStruct._fields_ + [("first_world", C.c_int32), ("raster_res", C.c_float), ("hit_robot", C.c_void_p)]


_oc_lib = None


def oracle_c_lib():
    """oracle/libmrca_oracle_c.so (plain-C restatement of the oracle, OpenMP)."""
    global _oc_lib
    if _oc_lib is None:
        d = os.path.join(ROOT, "oracle")
        so = os.path.join(d, "libmrca_oracle_c.so")
        src = os.path.join(d, "mrca_oracle_c.c")
        if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", d])
        _oc_lib = C.CDLL(so)
        _oc_lib.oc_max_threads.restype = C.c_int
    return _oc_lib


class COracleEnv(EmulEnv):
    """The C oracle behind the same numpy-array state as EmulEnv."""

    def __init__(self, sc, first_world=0):
        self.lib = oracle_c_lib()
        EmulEnv.__init__(self, sc)
        self.lib = oracle_c_lib()
        st = _OcEnvStruct()
        for name, _t in _EmulEnvStruct._fields_:
            setattr(st, name, getattr(self._st, name))
        st.first_world = first_world
        st.raster_res = float(getattr(sc, "collision_raster", 0.0))
        self.hit_robot = np.zeros((sc.num_robots, sc.beams), np.uint8)     # 1: the beam returned from another robot
        st.hit_robot = self.hit_robot.ctypes.data
        self._st = st

    def reset(self, mask=None, poses=None, goals=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p = None if poses is None else np.ascontiguousarray(poses, np.float32)
        g = None if goals is None else np.ascontiguousarray(goals, np.float32)
        self.lib.oc_reset(C.byref(self._st), C.c_void_p(m.ctypes.data) if m is not None else None,
                          C.c_void_p(p.ctypes.data) if p is not None else None,
                          C.c_void_p(g.ctypes.data) if g is not None else None)

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32)
        self.lib.oc_step(C.byref(self._st), C.c_void_p(a.ctypes.data))


class COracleBackend:
    """The C oracle behind the SharedWorld backend protocol of mrca/stage_world.py (tests only; much faster than the
    NumPy OracleBackend for the 50-robot circle)."""

    def __init__(self, sc):
        self.env = COracleEnv(sc)

    def reset(self, mask, poses, goals):
        self.env.reset(mask, poses, goals)

    def step(self, actions):
        self.env.step(actions)

    def field(self, name):
        return np.asarray(getattr(self.env, name))
