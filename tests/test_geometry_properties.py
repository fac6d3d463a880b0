"""Geometric building blocks of the tick, product (csrc/mrca_device.h through the host harness) vs first
principles: the beam-interval cull must never drop a beam that hits, the slab test must agree with an
independent float64 polygon clip, the rectangle SAT must be symmetric and agree with dense sampling."""
import ctypes as C

import numpy as np

import util as U
from util import O

HL, HW = 0.22, 0.19


def _P(a):
    return C.c_void_p(a.ctypes.data)


def test_beam_interval_never_culls_a_hit():
    lib = U.emul_lib()
    rng = np.random.default_rng(0)
    n = 4000
    bc, bs = O.beam_table(np.float32, 512)
    # neighbours anywhere within reach, including very close, behind, and on the +-90 degree edges
    r = np.concatenate([rng.uniform(0.45, 6.6, n - 600), rng.uniform(0.4, 0.7, 600)])
    phi = rng.uniform(-np.pi, np.pi, n)
    phi[:400] = rng.choice([-np.pi / 2, np.pi / 2, 0.0, np.pi], 400) + rng.normal(0, 0.02, 400)
    lx = (r * np.cos(phi)).astype(np.float32)
    ly = (r * np.sin(phi)).astype(np.float32)
    thj = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    sj, cj = O.sincos(thj, np.float32)
    lo = np.zeros(n, np.int32)
    hi = np.zeros(n, np.int32)
    lib.emul_beam_interval(n, _P(lx), _P(ly), 512, _P(lo), _P(hi))
    kept = 0
    for i in range(n):
        z = np.zeros(512, np.float32)
        t = np.empty(512, np.float32)
        ax, ay, as_, ac = (np.full(512, v, np.float32) for v in (lx[i], ly[i], sj[i], cj[i]))   # keep alive
        lib.emul_ray_box(512, _P(z), _P(z), _P(bc), _P(bs), _P(ax), _P(ay), _P(as_), _P(ac), _P(t))
        hit = np.nonzero(t < 6.0)[0]
        if hit.size:
            assert lo[i] <= hit.min() and hit.max() <= hi[i], (i, lo[i], hi[i], hit.min(), hit.max())
            kept += 1
            # and the interval is tight enough to be useful: at most a few beams of slack plus the angular size
            assert (hi[i] - lo[i]) <= (hit.max() - hit.min()) + 2 * (3 + int(120 / max(r[i], 0.45)))
    assert kept > 1000


def _clip_ray_rect(o, d, c, th):
    """float64 reference: entry parameter of ray o + t d into the rectangle centred c, heading th."""
    ca, sa = np.cos(th), np.sin(th)
    rel = o - c
    lo = np.array([rel[0] * ca + rel[1] * sa, rel[1] * ca - rel[0] * sa])
    ld = np.array([d[0] * ca + d[1] * sa, d[1] * ca - d[0] * sa])
    t0, t1 = -np.inf, np.inf
    for a, h in ((0, HL), (1, HW)):
        if abs(ld[a]) < 1e-15:
            if abs(lo[a]) > h:
                return np.inf
            continue
        ta, tb = (-h - lo[a]) / ld[a], (h - lo[a]) / ld[a]
        t0, t1 = max(t0, min(ta, tb)), min(t1, max(ta, tb))
    return max(t0, 0.0) if (t0 <= t1 and t1 >= 0) else np.inf


def test_ray_box_matches_float64_clip():
    lib = U.emul_lib()
    rng = np.random.default_rng(1)
    n = 20000
    ox, oy = rng.uniform(-5, 5, (2, n)).astype(np.float32)
    a = rng.uniform(-np.pi, np.pi, n)
    dx, dy = np.cos(a).astype(np.float32), np.sin(a).astype(np.float32)
    xj = (ox + rng.uniform(-4, 4, n)).astype(np.float32)
    yj = (oy + rng.uniform(-4, 4, n)).astype(np.float32)
    thj = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    sj, cj = O.sincos(thj, np.float32)
    t = np.empty(n, np.float32)
    lib.emul_ray_box(n, _P(ox), _P(oy), _P(dx), _P(dy), _P(xj), _P(yj), _P(sj), _P(cj), _P(t))
    to = O.ray_box(ox, oy, dx, dy, xj, yj, sj, cj, np.float32)
    assert (t.view(np.uint32) == to.view(np.uint32)).all()          # product == oracle, bitwise
    bad = 0
    for i in range(0, n, 7):
        ref = _clip_ray_rect(np.array([ox[i], oy[i]], float), np.array([dx[i], dy[i]], float),
                             np.array([xj[i], yj[i]], float), float(thj[i]))
        if np.isinf(ref) != np.isinf(t[i]):
            bad += 1                                  # grazing rays may flip between fp32 and fp64
        elif not np.isinf(ref):
            assert abs(ref - t[i]) < 1e-4
    assert bad <= 3


def _corners(x, y, th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[x + hx * c - hy * s, y + hx * s + hy * c] for hx, hy in ((HL, HW), (-HL, HW), (-HL, -HW), (HL, -HW))])


def _inside(p, x, y, th):
    c, s = np.cos(th), np.sin(th)
    rx, ry = p[:, 0] - x, p[:, 1] - y
    return (np.abs(rx * c + ry * s) <= HL) & (np.abs(ry * c - rx * s) <= HW)


def test_obb_overlap_symmetric_and_agrees_with_sampling():
    lib = U.emul_lib()
    lib.emul_obb.argtypes = [C.c_float] * 8
    rng = np.random.default_rng(2)
    g = np.stack(np.meshgrid(np.linspace(-HL, HL, 23), np.linspace(-HW, HW, 21)), -1).reshape(-1, 2)
    n_ov = 0
    for _ in range(3000):
        xi, yi, xj, yj = rng.uniform(-0.5, 0.5, 4)
        ti, tj = rng.uniform(-np.pi, np.pi, 2)
        si, ci = (float(v) for v in O.sincos(np.float32(ti), np.float32))
        sj, cj = (float(v) for v in O.sincos(np.float32(tj), np.float32))
        a = lib.emul_obb(xi, yi, si, ci, xj, yj, sj, cj)
        b = lib.emul_obb(xj, yj, sj, cj, xi, yi, si, ci)
        assert a == b
        o = O.obb_overlap(np.float32(xi), np.float32(yi), np.float32(si), np.float32(ci), np.float32(xj),
                          np.float32(yj), np.float32(sj), np.float32(cj), np.float32)
        assert bool(o) == bool(a)
        # sample points of rectangle i (its own frame grid) inside rectangle j, and vice versa
        ca, sa = np.cos(ti), np.sin(ti)
        pts_i = np.stack([xi + g[:, 0] * ca - g[:, 1] * sa, yi + g[:, 0] * sa + g[:, 1] * ca], 1)
        cb, sb = np.cos(tj), np.sin(tj)
        pts_j = np.stack([xj + g[:, 0] * cb - g[:, 1] * sb, yj + g[:, 0] * sb + g[:, 1] * cb], 1)
        sampled = _inside(pts_i, xj, yj, tj).any() or _inside(pts_j, xi, yi, ti).any()
        if sampled:
            assert a == 1          # a common point proves overlap; SAT must agree
        n_ov += a
    assert 500 < n_ov < 2900


def test_norm_obs_equals_ieee_division():
    """mrca_device.h:norm_obs (multiply + two FMAs) == RN(RN(x / 6) - 0.5) for floats x in [0, 6]: every 7th bit
    pattern plus the neighbourhoods of 0, of the subnormal boundary, of powers of two and of 6 (the full sweep is
    tools/check_div6.c: 0 mismatches over all 1 086 324 737 values)."""
    import ctypes as C
    lib = U.emul_lib()
    top = np.float32(6.0).view(np.uint32)
    u = np.concatenate([np.arange(0, int(top) + 1, 7, dtype=np.uint32), np.arange(0, 70000, dtype=np.uint32),
                        np.arange(0x007F0000, 0x00810000, dtype=np.uint32),
                        np.arange(int(top) - 70000, int(top) + 1, dtype=np.uint32)] +
                       [np.arange(e - 3000, e + 3000, dtype=np.uint32) for e in range(0x30000000, 0x40800001,
                                                                                       0x00800000)])
    x = u.view(np.float32)
    got = np.empty_like(x)
    lib.emul_norm_obs(C.c_void_p(x.ctypes.data), C.c_int(x.size), C.c_void_p(got.ctypes.data))
    want = (x / np.float32(6.0) - np.float32(0.5)).astype(np.float32)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_outline_edge_as_independent_crossing_events_equals_the_walk():
    """move_kernel tests a footprint edge against its LDS patch with one lane per crossing EVENT of the walk (the start cell,
    every x crossing, every y crossing: walk_event_hits) instead of one lane walking ~13 dependent steps; the oracle and the
    specification (grid_march) walk.  Edge by edge the OR over the events must equal the walk's answer: on the shipped maps at
    both cell sizes, on the circle world's 0.1 m map and on a synthetic map with boxes, for poses scattered over the map,
    poses hugging occupied cells, headings along the raster and centres on raster lines."""
    import ctypes as C
    from util import S
    maps = {"stage1": S.stage1(num_worlds=1, robots_per_world=4).grid, "stage2": S.stage2(num_worlds=1).grid,
            "stage1 r0.2": S.stage1(num_worlds=1, robots_per_world=4, stage_resolution=True).grid,
            "circle": S.circle(num_worlds=1).grid,
            "boxes": U.small_grid(cell=0.1, size=16.0, ring_radius=7.0, blocks=[(-1.0, -1.0, 1.0, 0.5), (3.0, 2.0, 3.3, 2.2)]),
            "fine": U.small_grid(cell=0.025, size=8.0, ring_radius=3.2, blocks=[(0.5, 0.5, 1.5, 1.0)])}
    rng = np.random.default_rng(17)
    for name, grid in maps.items():
        sc = S.stage1(num_worlds=1, robots_per_world=4, grid=grid)
        env = U.EmulEnv(sc)
        env.lib.emul_outline_hits.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        n = 60000
        half = 0.5 * min(grid.width, grid.height) * grid.cell
        x = rng.uniform(-half - 0.5, half + 0.5, n).astype(np.float32)
        y = rng.uniform(-half - 0.5, half + 0.5, n).astype(np.float32)
        th = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
        # a third of the poses next to occupied cells: centre within 0.45 m of a random occupied cell
        occ = np.argwhere(grid.dense())
        m = n // 3
        pick = occ[rng.integers(0, len(occ), m)]
        x[:m] = (grid.x0 + (pick[:, 1] + 0.5) * grid.cell + rng.uniform(-0.45, 0.45, m)).astype(np.float32)
        y[:m] = (grid.y0 + (pick[:, 0] + 0.5) * grid.cell + rng.uniform(-0.45, 0.45, m)).astype(np.float32)
        th[: n // 8] = rng.choice(np.array([0.0, np.pi / 2, -np.pi / 2, np.pi], np.float32), n // 8)     # edges along the raster
        k = n // 6
        x[m: m + k] = (np.round(x[m: m + k] / grid.cell) * grid.cell).astype(np.float32)                  # centres on raster lines
        y[m + k // 2: m + k] = (np.round(y[m + k // 2: m + k] / grid.cell) * grid.cell).astype(np.float32)
        a = np.zeros(n, np.int32)
        b = np.zeros(n, np.int32)
        Q = env.lib.emul_outline_hits(C.byref(env._st), n, x.ctypes.data, y.ctypes.data, th.ctypes.data, a.ctypes.data, b.ctypes.data)
        assert Q == int(np.ceil(np.float32(0.44) * np.float32(1.0 / grid.cell))) + 1 or Q >= 3
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, (name, bad.size, x[bad[:3]], y[bad[:3]], th[bad[:3]], a[bad[:3]], b[bad[:3]])
        assert 0.03 * n < (a != 0).sum() < 0.9 * n, name           # both outcomes occur
