"""NumPy oracle of the multi-robot Stage tick  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the reported CPU baseline.  The product path
(``rl-collision-avoidance_amd/``) never imports it and has no CPU fallback.

PARITY STATUS
  * env tick, libstage half (kinematics / collision / ray cast): **parity unpinned**.  The arithmetic lives in
    libstage (un-vendored, no version pin: stage_ros-add_pose_and_crash/package.xml:25,35;
    "Stage 4.1.1" per stageros.cpp:346,814), which is not under /root/reference, and the
    reference holds no golden vectors for it.  This file *restates* the published
    behaviour with the clean synchronous semantics fixed in DESIGN.md "Oracle decisions".
    The four qualitative rostest assertions the reference does hold
    (stage_ros-add_pose_and_crash/test/cmdpose_tests.py:87-203) are restated in
    tests/test_oracle_invariants.py.
    PINNED since round 6 by the one libstage-rendered artefact in the checkout, doc/stage2.gif (Stage's own GUI,
    23 frames of worlds/stage2.world; tools/make_golden_gif.py -> tests/golden/stage_gui_stage2.npz,
    tests/test_golden_gui.py): bitmap bounding box -> ``size`` scaling, image row 0 = +y, floorplan centred on its pose;
    polygon obstacles rescaled to ``size`` on their pose; the 0.44 x 0.38 footprint centred on the pose; cmd_pose = teleport
    onto the table poses; 'Reach Goal' inside 0.5 m of the robot's own table goal; displacement per tick = v x 0.1 s with
    v <= 1 as a hard ceiling; heading in degrees CCW from +x with motion along it.
    STILL UNPINNED (nothing in a GUI picture shows them): the order of operations inside one tick, Stage's raster
    collision test (fidelity mode's shared-cell rule), the quantisation of ranges to raster cells.
  * env tick, Python half (reward / terminal, observation, local goal, episode set-up, reset
    distributions): follows the reference's Python line by line (citations on each function) and is
    PINNED by golden vectors made by running the reference's own stage_world1.py / stage_world2.py /
    circle_world.py (tools/make_golden_env.py -> tests/golden/env_python_*.npz; fp64 mode == reference
    to 1e-12, tests/test_golden_env.py).
  * GAE / filter-index / policy: pinned against the reference's own importable functions
    (tests/golden/, tools/make_golden.py).

Two arithmetic modes, same operation order:
  dtype=float64 : the "clean maths" specification (north-star: NumPy re-implementation, 1e-5)
  dtype=float32 : every operation rounded to fp32 in the SAME order as the HIP kernels, built
                  only from IEEE +,-,*,/,sqrt, comparisons and the polynomial sincos below, so
                  the GPU result can be compared bit-for-bit, flags included.
"""
import numpy as np

# --------------------------------------------------------------------------------------------
# constants (cited there)
BEAMS = 512                 # stage1.world:14 samples 512
FOV = np.pi                 # stage1.world:12 fov 180
RANGE_MAX = 6.0             # stage1.world:13 range [0 6]
DT = 0.1                    # Stage default interval_sim (no override in worlds/*.world:1-5)
HALF_LEN = 0.22             # stage1.world:83 size [0.44 0.38 0.22]
HALF_WID = 0.19
GOAL_RADIUS = 0.5           # stage_world1.py:34
R_ARRIVE = 15.0             # stage_world1.py:195
R_CRASH = -15.0             # stage_world1.py:200
K_PROGRESS = 2.5            # stage_world1.py:187
K_OMEGA = -0.1              # stage_world1.py:204

RESULT_NONE, RESULT_REACH, RESULT_CRASH, RESULT_TIMEOUT = 0, 1, 2, 3

RESET_TABLE, RESET_DISC, RESET_REGION = 0, 1, 2      # per-robot reset rule
AUTO_NONE, AUTO_ROBOT, AUTO_GROUP = 0, 1, 2           # episode structure
BIG_WORLD = 64              # worlds with more robots use the distance-culled passes (_collide_big / _raycast_big)
CULL_COLLIDE = 0.6          # > 2 x circumradius 0.2907: rectangles further apart cannot overlap
CULL_LIDAR = 6.3            # > 6 m + circumradius: robots further away cannot return a range below 6 m
MAX_TRIES_POSE = 64
MAX_TRIES_GOAL = 256
STREAM_POSE, STREAM_GOAL = 0, 1


# --------------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al., SC'11 -- Random123).  Pinned by the Random123 known-answer
# vectors in tests/test_oracle_philox.py.
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)


def philox4x32(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.asarray(k0, dtype=np.uint32)
    k1 = np.asarray(k1, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for r in range(10):
            if r:
                k0 = k0 + _W0
                k1 = k1 + _W1
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
    return c0, c1, c2, c3


def u01(x, dtype):
    """uint32 -> [0,1): top 24 bits * 2^-24 (exact in fp32)."""
    return (x >> np.uint32(8)).astype(dtype) * dtype(1.0 / 16777216.0)


# --------------------------------------------------------------------------------------------
# deterministic sincos: Cody-Waite reduction by pi/2 + Cephes single-precision polynomials,
# written with separately rounded * and + only (no fused multiply-add).
_TWO_OVER_PI = 0.6366197723675814
_DP1 = 1.5703125
_DP2 = 4.837512969970703125e-4
_DP3 = 7.54978995489188e-8
_S1, _S2, _S3 = -1.9515295891e-4, 8.3321608736e-3, -1.6666654611e-1
_C1, _C2, _C3 = 2.443315711809948e-5, -1.388731625493765e-3, 4.166664568298827e-2


def sincos(th, dtype):
    """Returns (sin, cos).  fp64 mode uses libm; fp32 mode uses the shared polynomial."""
    if dtype is np.float64:
        return np.sin(th), np.cos(th)
    f = dtype
    th = np.asarray(th, dtype=f)
    k = np.rint(th * f(_TWO_OVER_PI))
    r = ((th - k * f(_DP1)) - k * f(_DP2)) - k * f(_DP3)
    z = r * r
    s = ((f(_S1) * z + f(_S2)) * z + f(_S3)) * z * r + r
    c = ((f(_C1) * z + f(_C2)) * z + f(_C3)) * (z * z) + (f(1.0) - f(0.5) * z)
    q = k.astype(np.int32) & 3
    sin = np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c)))
    cos = np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s)))
    return sin.astype(f), cos.astype(f)


def wrap_angle(th, dtype):
    """(-pi, pi]  --  the GT yaw convention after the quaternion round trip
    (stageros.cpp:575-583 -> stage_world1.py:88-91)."""
    f = dtype
    pi = f(np.pi)
    two_pi = f(2.0 * np.pi)
    th = np.where(th > pi, th - two_pi, th)
    th = np.where(th <= -pi, th + two_pi, th)
    return th.astype(f)


def beam_table(dtype, beams=BEAMS):
    """Robot-frame beam directions: bearing_i = -fov/2 + i*fov/(beams-1)
    (stageros.cpp:495-497).  Computed in float64, rounded once to the working type; the
    product uploads the identical table."""
    b = -FOV / 2.0 + np.arange(beams, dtype=np.float64) * (FOV / (beams - 1))
    return np.cos(b).astype(dtype), np.sin(b).astype(dtype)


# --------------------------------------------------------------------------------------------
class GridMap:
    """Occupancy grid, cells outside [0,W)x[0,H) are free."""

    def __init__(self, bits, width, height, cell, x0, y0):
        self.bits = np.ascontiguousarray(bits, dtype=np.uint32)
        self.width, self.height = int(width), int(height)
        self.wpr = self.bits.shape[1]
        self.cell, self.x0, self.y0 = float(cell), float(x0), float(y0)

    @classmethod
    def empty(cls, width, height, cell, x0, y0):
        return cls(np.zeros((height, (width + 31) // 32), dtype=np.uint32), width, height, cell, x0, y0)

    def set_cell(self, ix, iy, v=True):
        if v:
            self.bits[iy, ix >> 5] |= np.uint32(1 << (ix & 31))
        else:
            self.bits[iy, ix >> 5] &= np.uint32(~(1 << (ix & 31)) & 0xFFFFFFFF)

    def occupied(self, ix, iy):
        ix = np.asarray(ix)
        iy = np.asarray(iy)
        inb = (ix >= 0) & (ix < self.width) & (iy >= 0) & (iy < self.height)
        cx = np.where(inb, ix, 0)
        cy = np.where(inb, iy, 0)
        w = self.bits[cy, cx >> 5]
        return inb & (((w >> (cx & 31).astype(np.uint32)) & np.uint32(1)) != 0)


def grid_march(gm, ox, oy, dx, dy, tmax, dtype):
    """First occupied cell along the ray o + t*d, 0 <= t < tmax (metres; d is unit length).

    Cell sequence = exact grid traversal driven by CLOSED-FORM boundary times
        tx(b) = (float(b) - fx) * (1/dx),  ty likewise        (cell units)
    so the visited sequence does not depend on how an implementation walks it.  Returns the
    entry distance of the first occupied cell (0 if the start cell is occupied), else tmax.
    Ties tx == ty step in y first.  [libstage quantises ranges to its raster the same way --
    SURVEY Appendix B; restated, uncited.]
    """
    f = dtype
    shape = np.broadcast(ox, oy, dx, dy, tmax).shape
    ox, oy, dx, dy, tmax = (np.broadcast_to(np.asarray(a, dtype=f), shape).ravel() for a in (ox, oy, dx, dy, tmax))
    n = ox.size
    inv_cell = f(1.0) / f(gm.cell)
    fx = (ox - f(gm.x0)) * inv_cell
    fy = (oy - f(gm.y0)) * inv_cell
    ix = np.floor(fx).astype(np.int32)
    iy = np.floor(fy).astype(np.int32)
    tmax_c = tmax * inv_cell
    inf = f(np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv_dx = np.where(dx != 0, f(1.0) / dx, inf)
        inv_dy = np.where(dy != 0, f(1.0) / dy, inf)
    sx = np.where(dx > 0, 1, -1).astype(np.int32)
    sy = np.where(dy > 0, 1, -1).astype(np.int32)
    bx = np.where(dx > 0, ix + 1, ix).astype(np.int32)
    by = np.where(dy > 0, iy + 1, iy).astype(np.int32)
    with np.errstate(invalid="ignore"):
        tx = np.where(dx != 0, (bx.astype(f) - fx) * inv_dx, inf)
        ty = np.where(dy != 0, (by.astype(f) - fy) * inv_dy, inf)
    out = tmax.copy()
    start_occ = gm.occupied(ix, iy)
    out[start_occ] = f(0.0)
    act = np.nonzero(~start_occ & (tmax_c > 0))[0]
    ix, iy, bx, by, tx, ty = ix[act], iy[act], bx[act], by[act], tx[act], ty[act]
    fx, fy, inv_dx, inv_dy, sx, sy, tmc = fx[act], fy[act], inv_dx[act], inv_dy[act], sx[act], sy[act], tmax_c[act]
    dxnz, dynz = (dx[act] != 0), (dy[act] != 0)
    cell = f(gm.cell)
    while act.size:
        stepx = tx < ty
        t = np.where(stepx, tx, ty)
        ix = np.where(stepx, ix + sx, ix)
        bx = np.where(stepx, bx + sx, bx)
        iy = np.where(stepx, iy, iy + sy)
        by = np.where(stepx, by, by + sy)
        with np.errstate(invalid="ignore"):
            ntx = np.where(dxnz, (bx.astype(f) - fx) * inv_dx, inf)
            nty = np.where(dynz, (by.astype(f) - fy) * inv_dy, inf)
        tx = np.where(stepx, ntx, tx)
        ty = np.where(stepx, ty, nty)
        beyond = t >= tmc
        hit = ~beyond & gm.occupied(ix, iy)
        out[act[hit]] = (t[hit] * cell).astype(f)
        keep = ~(beyond | hit)
        act = act[keep]
        ix, iy, bx, by, tx, ty = ix[keep], iy[keep], bx[keep], by[keep], tx[keep], ty[keep]
        fx, fy, inv_dx, inv_dy, sx, sy, tmc = fx[keep], fy[keep], inv_dx[keep], inv_dy[keep], sx[keep], sy[keep], tmc[keep]
        dxnz, dynz = dxnz[keep], dynz[keep]
    return out.reshape(shape)


# --------------------------------------------------------------------------------------------
def footprint_corners(x, y, s, c, dtype):
    """Corners of the 0.44 x 0.38 rectangle (stage1.world:83), order (+,+),(-,+),(-,-),(+,-)."""
    f = dtype
    hx = np.array([HALF_LEN, -HALF_LEN, -HALF_LEN, HALF_LEN], dtype=f)
    hy = np.array([HALF_WID, HALF_WID, -HALF_WID, -HALF_WID], dtype=f)
    cx = x[..., None] + (hx * c[..., None] - hy * s[..., None])
    cy = y[..., None] + (hx * s[..., None] + hy * c[..., None])
    return cx.astype(f), cy.astype(f)


def static_hit(gm, x, y, s, c, dtype):
    """Robot outline vs occupancy grid: march each of the 4 edges from corner k towards corner
    k+1 (direction is the body axis, length 0.44 / 0.38); any occupied cell on the way is a
    collision.  [libstage tests the cells under a model's outline -- Appendix B; restated.]"""
    f = dtype
    cx, cy = footprint_corners(x, y, s, c, f)
    edx = np.stack([-c, s, c, -s], axis=-1).astype(f)
    edy = np.stack([-s, -c, s, c], axis=-1).astype(f)
    elen = np.array([2 * HALF_LEN, 2 * HALF_WID, 2 * HALF_LEN, 2 * HALF_WID], dtype=f)
    elen = np.broadcast_to(elen, cx.shape)
    t = grid_march(gm, cx, cy, edx, edy, elen, f)
    return (t < elen).any(axis=-1)


def walk_cells(inv_res, ox, oy, dx, dy, tmax, f):
    """Cells visited by the closed-form grid walk of ``grid_march`` on a raster of 1/inv_res metres aligned at the
    world origin: the start cell, then every cell entered at t < tmax (scalar inputs of dtype f)."""
    fx, fy = f(ox * inv_res), f(oy * inv_res)
    ix, iy = int(np.floor(fx)), int(np.floor(fy))
    tmax_c = f(tmax * inv_res)
    out = [(ix, iy)]
    if not tmax_c > 0:
        return out
    inf = f(np.inf)
    xnz, ynz = dx != 0, dy != 0
    inv_dx = f(f(1.0) / dx) if xnz else inf
    inv_dy = f(f(1.0) / dy) if ynz else inf
    sx, sy = (1 if dx > 0 else -1), (1 if dy > 0 else -1)
    bx, by = (ix + 1 if dx > 0 else ix), (iy + 1 if dy > 0 else iy)
    tx = f((f(bx) - fx) * inv_dx) if xnz else inf
    ty = f((f(by) - fy) * inv_dy) if ynz else inf
    while True:
        if tx < ty:
            t = tx
            ix += sx
            bx += sx
            tx = f((f(bx) - fx) * inv_dx)
        else:
            t = ty
            iy += sy
            by += sy
            ty = f((f(by) - fy) * inv_dy) if ynz else inf
        if t >= tmax_c:
            return out
        out.append((ix, iy))


def outline_cells(res, x, y, s, c, f):
    """Fidelity mode: the raster cells (side ``res``) under the outline of the 0.44 x 0.38 footprint -- the cells the
    grid walk visits along its four edges (corner k -> corner k+1 as in ``static_hit``).  [Stage maps a model's outline
    into its world raster and reports a collision when a cell also holds another model -- SURVEY Appendix B; restated.]"""
    inv_res = f(f(1.0) / f(res))
    hx = [HALF_LEN, -HALF_LEN, -HALF_LEN, HALF_LEN]
    hy = [HALF_WID, HALF_WID, -HALF_WID, -HALF_WID]
    ex = [-c, s, c, -s]
    ey = [-s, -c, s, c]
    el = [2 * HALF_LEN, 2 * HALF_WID, 2 * HALF_LEN, 2 * HALF_WID]
    cells = set()
    for k in range(4):
        cx = f(x + f(f(f(hx[k]) * c) - f(f(hy[k]) * s)))
        cy = f(y + f(f(f(hx[k]) * s) + f(f(hy[k]) * c)))
        cells.update(walk_cells(inv_res, cx, cy, f(ex[k]), f(ey[k]), f(el[k]), f))
    return cells


class RasterCells:
    """A set of raster cells (side ``res``, aligned at the world origin) behind the occupancy protocol ``grid_march``
    walks: fidelity mode's view of the OTHER robots for a lidar (their outline cells)."""

    def __init__(self, cells, res):
        self.cell, self.x0, self.y0 = float(res), 0.0, 0.0
        cells = np.asarray(sorted(cells), np.int64).reshape(-1, 2)
        self.ix0, self.iy0 = (int(cells[:, 0].min()), int(cells[:, 1].min())) if len(cells) else (0, 0)
        w = int(cells[:, 0].max()) - self.ix0 + 1 if len(cells) else 1
        h = int(cells[:, 1].max()) - self.iy0 + 1 if len(cells) else 1
        self.dense = np.zeros((h, w), bool)
        if len(cells):
            self.dense[cells[:, 1] - self.iy0, cells[:, 0] - self.ix0] = True

    def occupied(self, ix, iy):
        jx = np.asarray(ix, np.int64) - self.ix0
        jy = np.asarray(iy, np.int64) - self.iy0
        inb = (jx >= 0) & (jx < self.dense.shape[1]) & (jy >= 0) & (jy < self.dense.shape[0])
        return inb & self.dense[np.where(inb, jy, 0), np.where(inb, jx, 0)]


def obb_overlap(xi, yi, si, ci, xj, yj, sj, cj, dtype):
    """Separating-axis test of two 0.44 x 0.38 rectangles; touching counts as overlap."""
    f = dtype
    hx, hy = f(HALF_LEN), f(HALF_WID)
    tx = xj - xi
    ty = yj - yi
    a0 = np.abs(ci * cj + si * sj)
    a1 = np.abs(si * cj - ci * sj)
    ex = hx + (hx * a0 + hy * a1)
    ey = hy + (hx * a1 + hy * a0)
    sep = (np.abs(tx * ci + ty * si) > ex) | (np.abs(ty * ci - tx * si) > ey) | \
          (np.abs(tx * cj + ty * sj) > ex) | (np.abs(ty * cj - tx * sj) > ey)
    return ~sep


def ray_box(ox, oy, dx, dy, xj, yj, sj, cj, dtype):
    """Entry distance of ray (o,d) into robot j's rectangle, +inf on a miss.
    Robots are lidar-visible: ranger_return 0.5 (stage1.world:95)."""
    f = dtype
    hx, hy = f(HALF_LEN), f(HALF_WID)
    inf = f(np.inf)
    tiny = f(1e-12)
    rx = ox - xj
    ry = oy - yj
    lx = rx * cj + ry * sj
    ly = ry * cj - rx * sj
    ldx = dx * cj + dy * sj
    ldy = dy * cj - dx * sj

    def slab(lo, ld, h):
        par = np.abs(ld) < tiny
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inv = f(1.0) / np.where(par, f(1.0), ld)
            ta = (-h - lo) * inv
            tb = (h - lo) * inv
        t0 = np.where(par, np.where(np.abs(lo) > h, inf, -inf), np.minimum(ta, tb))
        t1 = np.where(par, np.where(np.abs(lo) > h, -inf, inf), np.maximum(ta, tb))
        return t0, t1

    t0x, t1x = slab(lx, ldx, hx)
    t0y, t1y = slab(ly, ldy, hy)
    tin = np.maximum(t0x, t0y)
    tout = np.minimum(t1x, t1y)
    hit = (tin <= tout) & (tout >= 0)
    return np.where(hit, np.maximum(tin, f(0.0)), inf).astype(f)


# --------------------------------------------------------------------------------------------
class OracleConfig:
    """Scenario description; mirrors include/mrca_env.h ``mrca_config`` field for field."""

    def __init__(self, num_worlds, robots_per_world, grid, *, timeout=150, w_thresh=1.05,
                 pre_dist_zero=False, auto_reset=AUTO_ROBOT, seed=0, reset_mode=None,
                 init_table=None, goal_table=None, group_id=None, beams=BEAMS, frames=3, first_world=0,
                 collision_raster=0.0, hold_velocity=False):
        self.W, self.R = int(num_worlds), int(robots_per_world)
        self.grid = grid
        self.timeout, self.w_thresh = int(timeout), float(w_thresh)
        self.pre_dist_zero, self.auto_reset, self.seed = bool(pre_dist_zero), int(auto_reset), int(seed)
        R = self.R
        self.reset_mode = np.full(R, RESET_DISC, np.int32) if reset_mode is None else np.asarray(reset_mode, np.int32)
        self.goal_mode = self.reset_mode.copy()
        self.init_table = np.zeros((R, 3)) if init_table is None else np.asarray(init_table, np.float64)
        self.goal_table = np.zeros((R, 2)) if goal_table is None else np.asarray(goal_table, np.float64)
        self.group_id = np.zeros(R, np.int32) if group_id is None else np.asarray(group_id, np.int32)
        self.beams, self.frames = int(beams), int(frames)
        self.first_world = int(first_world)   # simulate worlds [first_world, first_world+W) of a larger batch
        # fidelity: Stage keeps the last SetSpeed (stageros.cpp:272-280) -- a robot that is no longer commanded (dead,
        # ppo_stage2.py:72-74 sends nothing for it) keeps driving at it, and get_self_speed (the odom twist) still shows
        # it after reset_pose's teleport.  False: dead robots idle, the speed input restarts at 0
        self.hold_velocity = bool(hold_velocity)
        self.collision_raster = float(collision_raster)   # fidelity mode: > 0 = Stage-like raster collision between robots


class OracleEnv:
    """Synchronous batched restatement of one Stage tick + the StageWorld getters.

    Tick order (DESIGN.md "Oracle decisions" i-x):
      latch action -> integrate (explicit Euler, heading at tick start) -> collide in robot
      order within a world, revert + stall on hit -> GT velocity -> reward / terminal ->
      episode bookkeeping (auto reset) -> ray cast at the final pose -> observation stack.
    """

    def __init__(self, cfg, dtype=np.float32):
        self.cfg = cfg
        self.f = f = np.float32 if dtype in (np.float32, "float32") else np.float64
        self.N = N = cfg.W * cfg.R
        self.local = (np.arange(N) % cfg.R).astype(np.int32)
        # before the first reset the robots stand where the world file puts them (the table's rows ARE the agent lines of
        # worlds/stage2.world, tools/make_maps.py): stage_world2.py:250-268 keeps a region-sampled start 7 m away from
        # the robot's CURRENT position, its first one included
        self.pose = cfg.init_table[self.local].astype(f)
        self.speed = np.zeros((N, 2), f)        # odom velocity = last latched command (stageros.cpp:543-558)
        self.speed_gt = np.zeros((N, 2), f)     # finite-difference GT velocity (stageros.cpp:585-590)
        self.goal = np.zeros((N, 2), f)
        self.init_pose = np.zeros((N, 3), f)
        self.prev_dist = np.zeros(N, f)
        self.t = np.ones(N, np.int32)
        self.episode = np.zeros(N, np.int32)
        self.crashed = np.zeros(N, np.int8)
        self.live = np.ones(N, np.int8)
        self.reward = np.zeros(N, f)
        self.done = np.zeros(N, np.int8)
        self.result = np.zeros(N, np.int8)
        self.first_result = np.zeros(N, np.int8)
        self.scan = np.full((N, cfg.beams), RANGE_MAX, f)
        self.obs = np.zeros((N, cfg.frames, cfg.beams), f)
        self.local_goal = np.zeros((N, 2), f)
        self.bcos, self.bsin = beam_table(f, cfg.beams)

    # ---------------------------------------------------------------- resets
    def _key(self):
        return np.uint32(self.cfg.seed & 0xFFFFFFFF), np.uint32((self.cfg.seed >> 32) & 0xFFFFFFFF)

    def _draw(self, gid, episode, k, stream):
        k0, k1 = self._key()
        gid = gid + self.cfg.first_world * self.cfg.R
        return philox4x32(gid.astype(np.uint32), episode.astype(np.uint32),
                          np.full(gid.shape, k, np.uint32), np.full(gid.shape, stream, np.uint32), k0, k1)

    def _sample_pose(self, idx):
        """stage_world1.py:251-260 (disc r<=9, theta~U(0,2pi)); stage_world2.py:250-268 (region,
        >=7 m from the current position); tables model/utils.py:6-23,41-52."""
        f = self.f
        cfg = self.cfg
        mode = cfg.reset_mode[self.local[idx]]
        out = cfg.init_table[self.local[idx]].astype(f)
        out[:, 2] = wrap_angle(out[:, 2], f)
        pend = mode != RESET_TABLE
        cur = self.pose[idx]
        for k in range(MAX_TRIES_POSE):
            if not pend.any():
                break
            r0, r1, r2, _ = self._draw(idx, self.episode[idx], k, STREAM_POSE)
            ua, ub, uc = u01(r0, f), u01(r1, f), u01(r2, f)
            # disc
            xd = f(-9.0) + f(18.0) * ua
            yd = f(-9.0) + f(18.0) * ub
            okd = np.sqrt(xd * xd + yd * yd) <= f(9.0)
            # stage-2 region
            xr = f(9.0) + f(10.0) * ua
            yr = np.where(ub <= f(0.4), -(ub * f(10.0) + f(1.0)), -(ub * f(10.0) + f(9.0)))
            ddx = xr - cur[:, 0]
            ddy = yr - cur[:, 1]
            okr = ~(np.sqrt(ddx * ddx + ddy * ddy) < f(7.0))
            disc = mode == RESET_DISC
            x = np.where(disc, xd, xr)
            y = np.where(disc, yd, yr)
            ok = np.where(disc, okd, okr)
            th = wrap_angle(f(2.0 * np.pi) * uc, f)
            last = k == MAX_TRIES_POSE - 1
            take = pend & (ok | last)
            out[take, 0], out[take, 1], out[take, 2] = x[take], y[take], th[take]
            pend = pend & ~take
        return out

    def _sample_goal(self, idx):
        """stage_world1.py:262-274 (disc r<=9, 8..10 m from init pose); stage_world2.py:270-287
        (region, >=7 m from the robot); tables model/utils.py:25-38,54-63."""
        f = self.f
        cfg = self.cfg
        mode = cfg.goal_mode[self.local[idx]]
        out = cfg.goal_table[self.local[idx]].astype(f)
        pend = mode != RESET_TABLE
        cur = self.pose[idx]
        for k in range(MAX_TRIES_GOAL):
            if not pend.any():
                break
            r0, r1, _, _ = self._draw(idx, self.episode[idx], k, STREAM_GOAL)
            ua, ub = u01(r0, f), u01(r1, f)
            xd = f(-9.0) + f(18.0) * ua
            yd = f(-9.0) + f(18.0) * ub
            do = np.sqrt(xd * xd + yd * yd)
            gx = xd - cur[:, 0]
            gy = yd - cur[:, 1]
            dg = np.sqrt(gx * gx + gy * gy)
            okd = ~((do > f(9.0)) | (dg > f(10.0)) | (dg < f(8.0)))
            xr = f(9.0) + f(10.0) * ua
            yr = np.where(ub <= f(0.4), -(ub * f(10.0) + f(1.0)), -(ub * f(10.0) + f(9.0)))
            rx = xr - cur[:, 0]
            ry = yr - cur[:, 1]
            okr = ~(np.sqrt(rx * rx + ry * ry) < f(7.0))
            disc = mode == RESET_DISC
            x = np.where(disc, xd, xr)
            y = np.where(disc, yd, yr)
            ok = np.where(disc, okd, okr)
            last = k == MAX_TRIES_GOAL - 1
            take = pend & (ok | last)
            out[take, 0], out[take, 1] = x[take], y[take]
            pend = pend & ~take
        return out

    def _begin_episode(self, idx, poses=None, goals=None):
        """reset_pose + generate_goal_point (stage_world1.py:171-177,213-223)."""
        f = self.f
        if idx.size == 0:
            return
        self.pose[idx] = self._sample_pose(idx) if poses is None else np.asarray(poses, f)
        self.init_pose[idx] = self.pose[idx]
        self.goal[idx] = self._sample_goal(idx) if goals is None else np.asarray(goals, f)
        dx = self.goal[idx, 0] - self.pose[idx, 0]
        dy = self.goal[idx, 1] - self.pose[idx, 1]
        d = np.sqrt(dx * dx + dy * dy)
        # stage_world2.py:170-171 / circle_world.py:166-167 start pre_distance at 0 (quirk)
        self.prev_dist[idx] = f(0.0) if self.cfg.pre_dist_zero else d
        self.t[idx] = 1
        self.crashed[idx] = 0
        self.live[idx] = 1
        if not self.cfg.hold_velocity:
            self.speed[idx] = 0
        self.speed_gt[idx] = 0

    def reset(self, mask=None, poses=None, goals=None):
        idx = np.arange(self.N) if mask is None else np.nonzero(np.asarray(mask))[0]
        self.episode[idx] += 1
        if poses is not None:
            poses = np.asarray(poses, self.f).reshape(self.N, 3)[idx]
        if goals is not None:
            goals = np.asarray(goals, self.f).reshape(self.N, 2)[idx]
        self._begin_episode(idx, poses, goals)
        fresh = np.zeros(self.N, bool)
        fresh[idx] = True
        self.done[idx] = 0
        self.result[idx] = 0
        self.reward[idx] = 0
        self.first_result[idx] = 0
        # only the robots that were reset are re-observed; the others keep the scan of the last
        # tick until the next one (their cached LaserScan in the reference, stage_world1.py:97-101)
        self._observe(fresh, only_fresh=True)

    # ---------------------------------------------------------------- tick
    def step(self, actions):
        f = self.f
        cfg = self.cfg
        N, R = self.N, cfg.R
        act = np.asarray(actions, dtype=f).reshape(N, 2)
        act = np.where(np.isfinite(act), act, f(0.0)).astype(f)   # a non-finite command idles the robot
        live = self.live.astype(bool)
        held = self.speed if cfg.hold_velocity else np.zeros_like(self.speed)
        v = np.where(live, act[:, 0], held[:, 0]).astype(f)
        w = np.where(live, act[:, 1], held[:, 1]).astype(f)
        self.speed[:, 0], self.speed[:, 1] = v, w

        # -- integrate (explicit Euler, heading at tick start)
        x, y, th = self.pose[:, 0].copy(), self.pose[:, 1].copy(), self.pose[:, 2].copy()
        s, c = sincos(th, f)
        d = v * f(DT)
        nx = x + d * c
        ny = y + d * s
        nth = wrap_angle(th + w * f(DT), f)
        ns, nc = sincos(nth, f)
        moving = (v != 0) | (w != 0)
        shit = static_hit(cfg.grid, nx, ny, ns, nc, f)

        # -- collide in robot order within each world (Gauss-Seidel, like Stage's model loop)
        cx, cy, cs, cc = x.copy(), y.copy(), s.copy(), c.copy()   # current poses / headings
        cth = th.copy()
        moved = np.zeros(N, bool)
        base = np.arange(cfg.W) * R
        if cfg.collision_raster > 0:
            self._collide_raster(nx, ny, nth, ns, nc, moving, shit, cx, cy, cth, cs, cc, moved)
        elif R > BIG_WORLD:
            self._collide_big(nx, ny, nth, ns, nc, moving, shit, cx, cy, cth, cs, cc, moved)
        for i in range(R if (R <= BIG_WORLD and cfg.collision_raster <= 0) else 0):
            ii = base + i
            hit = shit[ii].copy()
            for j in range(R):
                if j == i:
                    continue
                jj = base + j
                hit |= obb_overlap(nx[ii], ny[ii], ns[ii], nc[ii], cx[jj], cy[jj], cs[jj], cc[jj], f)
            mv = moving[ii]
            ok = mv & ~hit
            cx[ii] = np.where(ok, nx[ii], cx[ii])
            cy[ii] = np.where(ok, ny[ii], cy[ii])
            cth[ii] = np.where(ok, nth[ii], cth[ii])
            cs[ii] = np.where(ok, ns[ii], cs[ii])
            cc[ii] = np.where(ok, nc[ii], cc[ii])
            moved[ii] = ok
            self.crashed[ii] = np.where(mv, hit.astype(np.int8), self.crashed[ii])
        self.pose[:, 0], self.pose[:, 1], self.pose[:, 2] = cx, cy, cth

        # -- GT velocity by finite difference = commanded if the move succeeded (stageros.cpp:585-590)
        self.speed_gt[:, 0] = np.where(moved, np.abs(v), f(0.0))
        self.speed_gt[:, 1] = np.where(moved, w, f(0.0))

        # -- reward / terminal (stage_world1.py:180-211)
        gx = self.goal[:, 0] - cx
        gy = self.goal[:, 1] - cy
        dist = np.sqrt(gx * gx + gy * gy).astype(f)
        rg = (self.prev_dist - dist) * f(K_PROGRESS)
        reach = dist < f(GOAL_RADIUS)
        rg = np.where(reach, f(R_ARRIVE), rg)
        crash = self.crashed == 1
        rc = np.where(crash, f(R_CRASH), f(0.0))
        aw = np.abs(self.speed_gt[:, 1])
        rw = np.where(aw > f(cfg.w_thresh), f(K_OMEGA) * aw, f(0.0))
        tout = self.t > cfg.timeout
        result = np.where(reach, RESULT_REACH, RESULT_NONE)
        result = np.where(crash, RESULT_CRASH, result)
        result = np.where(tout, RESULT_TIMEOUT, result).astype(np.int8)
        done = (reach | crash | tout)
        reward = ((rg + rc) + rw).astype(f)
        self.reward = np.where(live, reward, self.reward).astype(f)
        self.done = np.where(live, done.astype(np.int8), self.done)
        self.result = np.where(live, result, self.result)
        self.prev_dist = np.where(live, dist, self.prev_dist).astype(f)
        self.t = np.where(live, self.t + 1, self.t).astype(np.int32)
        newly = live & done & (self.first_result == 0)
        self.first_result = np.where(newly, result, self.first_result).astype(np.int8)

        # -- episode bookkeeping
        fresh = np.zeros(N, bool)
        if cfg.auto_reset == AUTO_ROBOT:
            idx = np.nonzero(live & done)[0]
            self.episode[idx] += 1
            self._begin_episode(idx)
            fresh[idx] = True
        elif cfg.auto_reset == AUTO_GROUP:
            # ppo_stage2.py:72-107: a finished robot stops acting until its whole group is done
            self.live = np.where(live & done, 0, self.live).astype(np.int8)
            dn = self.done.reshape(cfg.W, R).astype(bool)
            gid = cfg.group_id
            grp_done = np.zeros((cfg.W, R), bool)
            for g in np.unique(gid):
                m = gid == g
                grp_done[:, m] = dn[:, m].all(axis=1, keepdims=True)
            idx = np.nonzero(grp_done.ravel())[0]
            self.episode[idx] += 1
            # robot order matters for RESET_REGION (distance to the *current* pose): sequentially
            # identical because each robot only looks at its own pose.
            self._begin_episode(idx)
            fresh[idx] = True
        self._observe(fresh)
        return self.obs, self.local_goal, self.speed, self.reward, self.done, self.result

    # ---------------------------------------------------------------- fidelity mode
    def _collide_raster(self, nx, ny, nth, ns, nc, moving, shit, cx, cy, cth, cs, cc, moved):
        """The robot-order pass with Stage's raster rule: robot i, at its provisional pose, collides with robot j iff
        their OUTLINES SHARE A RASTER CELL of ``collision_raster`` metres (j at the pose it has when it is i's turn)."""
        f, cfg = self.f, self.cfg
        R, res = cfg.R, cfg.collision_raster
        for w in range(cfg.W):
            cur = [outline_cells(res, cx[w * R + j], cy[w * R + j], cs[w * R + j], cc[w * R + j], f) for j in range(R)]
            for i in range(R):
                n = w * R + i
                if not moving[n]:
                    continue
                mine = outline_cells(res, nx[n], ny[n], ns[n], nc[n], f)
                hit = bool(shit[n]) or any((j != i) and not mine.isdisjoint(cur[j]) for j in range(R))
                if not hit:
                    cx[n], cy[n], cth[n], cs[n], cc[n] = nx[n], ny[n], nth[n], ns[n], nc[n]
                    moved[n] = True
                    cur[i] = mine
                self.crashed[n] = 1 if hit else 0

    # ---------------------------------------------------------------- worlds with more than 64 robots
    def _collide_big(self, nx, ny, nth, ns, nc, moving, shit, cx, cy, cth, cs, cc, moved):
        """The SAME robot-order pass for worlds too large for the all-pairs loop (a single 500 / 50 000-robot
        circle).  Only pairs are skipped that cannot overlap: two 0.44 x 0.38 rectangles whose
        centres are further apart than 2 x circumradius (0.5815 m) are separated on every axis, so robot i only
        needs the robots whose OLD or PROVISIONAL centre lies within CULL_COLLIDE = 0.6 m of its provisional
        centre (a robot is at one of the two when i is tested).  Candidates come from two k-d trees (float64
        copies of the fp32 coordinates: the cull itself is exact arithmetic on the same numbers); robots without
        any candidate commit straight away, the others are walked in index order exactly like the loop above."""
        from scipy.spatial import cKDTree
        f, cfg = self.f, self.cfg
        R = cfg.R
        for w in range(cfg.W):
            sl = slice(w * R, (w + 1) * R)
            old = np.stack([cx[sl], cy[sl]], 1).astype(np.float64)
            new = np.stack([nx[sl], ny[sl]], 1).astype(np.float64)
            t_old, t_new = cKDTree(old), cKDTree(new)
            near_old = t_old.query_ball_point(new, CULL_COLLIDE)
            near_new = t_new.query_ball_point(new, CULL_COLLIDE)
            for i in range(R):
                n = w * R + i
                if not moving[n]:
                    continue
                hit = bool(shit[n])
                cand = sorted(set(near_old[i]) | set(near_new[i]))
                for j in cand:
                    if j == i or hit:
                        continue
                    m = w * R + j
                    hit = bool(obb_overlap(nx[n], ny[n], ns[n], nc[n], cx[m], cy[m], cs[m], cc[m], f))
                if not hit:
                    cx[n], cy[n], cth[n], cs[n], cc[n] = nx[n], ny[n], nth[n], ns[n], nc[n]
                    moved[n] = True
                self.crashed[n] = 1 if hit else 0

    def _raycast_big(self):
        """Ray cast for worlds with more than 64 robots: a robot further than 6 m + circumradius (CULL_LIDAR =
        6.3 m) from the sensor cannot return a range below 6 m, so only the others are slab-tested."""
        from scipy.spatial import cKDTree
        f, cfg = self.f, self.cfg
        N, R, B = self.N, cfg.R, cfg.beams
        x, y, th = self.pose[:, 0], self.pose[:, 1], self.pose[:, 2]
        s, c = sincos(th, f)
        dx = c[:, None] * self.bcos[None, :] - s[:, None] * self.bsin[None, :]
        dy = s[:, None] * self.bcos[None, :] + c[:, None] * self.bsin[None, :]
        ox = np.broadcast_to(x[:, None], (N, B))
        oy = np.broadcast_to(y[:, None], (N, B))
        rng = grid_march(cfg.grid, ox, oy, dx, dy, f(RANGE_MAX), f)
        rng_map = rng.copy()
        self._hit_new = np.zeros((N, B), bool)
        for w in range(cfg.W):
            sl = slice(w * R, (w + 1) * R)
            pts = np.stack([x[sl], y[sl]], 1).astype(np.float64)
            tree = cKDTree(pts)
            pairs = tree.query_pairs(CULL_LIDAR, output_type="ndarray")
            if len(pairs) == 0:
                continue
            a = np.concatenate([pairs[:, 0], pairs[:, 1]]) + w * R      # sensor
            b = np.concatenate([pairs[:, 1], pairs[:, 0]]) + w * R      # target
            t = ray_box(x[a, None], y[a, None], dx[a], dy[a], x[b, None], y[b, None], s[b, None], c[b, None], f)
            np.minimum.at(rng, a, t)
        self._hit_new = rng < rng_map
        return np.minimum(rng, f(RANGE_MAX)).astype(f)

    # ---------------------------------------------------------------- sensing
    def raycast(self):
        """512 beams per robot against the grid and the other robots of the same world
        (stageros.cpp:479-516 geometry; libstage ray trace restated)."""
        f = self.f
        cfg = self.cfg
        if cfg.R > BIG_WORLD:
            return self._raycast_big()
        N, R, B = self.N, cfg.R, cfg.beams
        x, y, th = self.pose[:, 0], self.pose[:, 1], self.pose[:, 2]
        s, c = sincos(th, f)
        dx = c[:, None] * self.bcos[None, :] - s[:, None] * self.bsin[None, :]
        dy = s[:, None] * self.bcos[None, :] + c[:, None] * self.bsin[None, :]
        ox = np.broadcast_to(x[:, None], (N, B))
        oy = np.broadcast_to(y[:, None], (N, B))
        rng = grid_march(cfg.grid, ox, oy, dx, dy, f(RANGE_MAX), f)
        # what a beam hit: True = another robot closer than the floorplan (ranger_return 0.5 -> LaserScan intensity 0,
        # stageros.cpp:501-506); the product keeps it as one bit per beam beside its scan ring (MRCA_F_HIT_BITS)
        self._hit_new = np.zeros((N, B), bool)
        if cfg.collision_raster > 0:
            # Fidelity mode: the lidar sees the other robots through the SAME raster they collide on (Stage's ranger walks
            # the world raster the models are mapped into, worlds/stage1.world:3,94-95): the range is the entry distance of
            # the first raster cell along the beam that holds a piece of another robot's outline -- the closed-form grid walk
            # on cells of `collision_raster` metres aligned at the world origin, start cell included (range 0).
            res = cfg.collision_raster
            rng = rng.reshape(N, B).copy()
            for w in range(cfg.W):
                cells = [outline_cells(res, x[w * R + j], y[w * R + j], s[w * R + j], c[w * R + j], f) for j in range(R)]
                for i in range(R):
                    others = set().union(*(cells[j] for j in range(R) if j != i)) if R > 1 else set()
                    if not others:
                        continue
                    n = w * R + i
                    t = grid_march(RasterCells(others, res), ox[n], oy[n], dx[n], dy[n], f(RANGE_MAX), f)
                    self._hit_new[n] = t < rng[n]
                    rng[n] = np.minimum(rng[n], t)
            return np.minimum(rng, f(RANGE_MAX)).reshape(N, B).astype(f)
        xs, ys = x.reshape(cfg.W, R), y.reshape(cfg.W, R)
        ss, cs = s.reshape(cfg.W, R), c.reshape(cfg.W, R)
        rng = rng.reshape(cfg.W, R, B)
        dxw, dyw = dx.reshape(cfg.W, R, B), dy.reshape(cfg.W, R, B)
        for j in range(R):
            tj = ray_box(xs[:, :, None], ys[:, :, None], dxw, dyw,
                         xs[:, j, None, None], ys[:, j, None, None], ss[:, j, None, None], cs[:, j, None, None], f)
            tj[:, j, :] = np.inf
            self._hit_new |= (tj < rng).reshape(N, B)
            rng = np.minimum(rng, tj)
        return np.minimum(rng, f(RANGE_MAX)).reshape(N, B).astype(f)

    def _observe(self, fresh, only_fresh=False):
        f = self.f
        upd = fresh if only_fresh else np.ones(self.N, bool)
        self.scan = np.where(upd[:, None], self.raycast(), self.scan).astype(f)
        self.hit_robot = np.where(upd[:, None], self._hit_new, getattr(self, "hit_robot", np.zeros_like(self._hit_new)))
        # stage_world1.py:122-140: NaN/inf -> 6, identity sub-sampling at 512 beams, scan/6 - 0.5
        new = (self.scan / f(6.0) - f(0.5)).astype(f)
        F = self.cfg.frames
        shifted = np.concatenate([self.obs[:, 1:], new[:, None]], axis=1)
        filled = np.repeat(new[:, None], F, axis=1)      # deque([obs, obs, obs]) ppo_stage1.py:59-60
        nobs = np.where(fresh[:, None, None], filled, shifted)
        self.obs = np.where(upd[:, None, None], nobs, self.obs).astype(f)
        # stage_world1.py:155-160
        s, c = sincos(self.pose[:, 2], f)
        gx = self.goal[:, 0] - self.pose[:, 0]
        gy = self.goal[:, 1] - self.pose[:, 1]
        self.local_goal[:, 0] = np.where(upd, gx * c + gy * s, self.local_goal[:, 0])
        self.local_goal[:, 1] = np.where(upd, gy * c - gx * s, self.local_goal[:, 1])


# --------------------------------------------------------------------------------------------
# learner-side restatements (pinned against the reference's importable functions)
def gae(rewards, values, last_value, dones, gamma, lam, dtype=np.float64):
    """model/ppo.py:122-139 generate_train_data."""
    f = dtype
    T, N = rewards.shape
    v = np.concatenate([np.asarray(values, f).reshape(T, N), np.asarray(last_value, f).reshape(1, N)], 0)
    r = np.asarray(rewards, f)
    nd = f(1.0) - np.asarray(dones, f)
    targets = np.zeros((T, N), f)
    g = np.zeros(N, f)
    for t in range(T - 1, -1, -1):
        delta = r[t] + f(gamma) * v[t + 1] * nd[t] - v[t]
        g = delta + f(gamma) * f(lam) * nd[t] * g
        targets[t] = g + v[t]
    return targets, targets - v[:-1]


def filter_index(dones):
    """model/utils.py:65-78 get_filter_index, including the not-reset-between-envs quirk."""
    T, N = dones.shape
    out = []
    flag = 0
    for i in range(N):
        for j in range(T):
            flag = flag + 1 if dones[j, i] else 0
            if flag >= 2:
                out.append(N * j + i)
    return out
