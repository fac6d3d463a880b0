"""The product's skipping ray march (free-rectangle field, csrc/mrca_device.h:grid_march_skip) must return
EXACTLY what the plain cell-by-cell walk returns -- and that walk must equal the NumPy oracle's -- on rays
chosen to hit the awkward cases: origins on cell boundaries, inside walls, outside the map, axis-aligned and
45-degree directions, zero components, short ranges (the robot-outline walks use 0.38 / 0.44 m)."""
import ctypes as C

import numpy as np
import pytest

import util as U
from util import S, O


def _rays(grid, n, rng):
    g = grid
    ox = rng.uniform(g.x0 - 2, g.x0 + g.width * g.cell + 2, n).astype(np.float32)
    oy = rng.uniform(g.y0 - 2, g.y0 + g.height * g.cell + 2, n).astype(np.float32)
    th = rng.uniform(-np.pi, np.pi, n)
    th[:4000] = rng.integers(0, 8, 4000) * np.pi / 4
    dx = np.cos(th).astype(np.float32)
    dy = np.sin(th).astype(np.float32)
    dx[:1000] = np.where(np.abs(dx[:1000]) < 1e-6, 0, dx[:1000])
    dy[:1000] = np.where(np.abs(dy[:1000]) < 1e-6, 0, dy[:1000])
    ox[4000:12000] = (g.x0 + rng.integers(0, g.width, 8000) * np.float32(g.cell)).astype(np.float32)
    oy[8000:16000] = (g.y0 + rng.integers(0, g.height, 8000) * np.float32(g.cell)).astype(np.float32)
    tmax = np.full(n, 6.0, np.float32)
    tmax[::7] = rng.uniform(0, 6, len(tmax[::7])).astype(np.float32)
    tmax[::11] = 0.44
    tmax[::13] = 0.38
    return ox, oy, dx, dy, tmax


@pytest.mark.parametrize("name", ["stage1", "stage2", "circle", "synthetic"])
def test_skipping_march_equals_plain_walk_and_oracle(name):
    if name == "synthetic":
        sc = S.stage1(1, 2, grid=U.small_grid(cell=0.1, size=16.0, ring_radius=7.0, blocks=[(-1, -1, 1, 0.5)]))
    else:
        sc = {"stage1": S.stage1(1, 2), "stage2": S.stage2(1), "circle": S.circle(1)}[name]
    e = U.EmulEnv(sc)
    rng = np.random.default_rng(hash(name) % 1000)
    n = 400000
    ox, oy, dx, dy, tmax = _rays(sc.grid, n, rng)
    a = np.empty(n, np.float32)
    b = np.empty(n, np.float32)
    P = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    e.lib.emul_march(C.byref(e._st), n, P(ox), P(oy), P(dx), P(dy), P(tmax), P(a), P(b))
    same = a.view(np.uint32) == b.view(np.uint32)
    assert same.all(), f"{(~same).sum()} rays differ, first at {np.argwhere(~same)[0]}"
    assert 0.1 < (a < tmax).mean() < 0.9          # the sample has both hits and misses
    # the lock-step K-ray march of the ray-cast kernel (grid_march_skip_n): each ray marched next to partners
    # that finish earlier / later must still return exactly the single-ray result
    for K in (2, 4):
        c = np.empty(n, np.float32)
        e.lib.emul_march_lockstep(C.byref(e._st), n, K, P(ox), P(oy), P(dx), P(dy), P(tmax), P(c))
        same = c.view(np.uint32) == b.view(np.uint32)
        assert same.all(), f"K={K}: {(~same).sum()} rays differ, first at {np.argwhere(~same)[0]}"
    m = 15000
    g = sc.grid
    gm = O.GridMap(g.bits, g.width, g.height, g.cell, g.x0, g.y0)
    o = O.grid_march(gm, ox[:m], oy[:m], dx[:m], dy[:m], tmax[:m], np.float32)
    assert (o.view(np.uint32) == b[:m].view(np.uint32)).all()


def test_free_rectangle_field_is_sound():
    """Every rectangle stored for an empty cell -- four per cell, one per quadrant of the direction of travel, the cell
    in the corner the ray enters through -- must contain no occupied cell (the march's only assumption about the field),
    and occupied cells must be flagged in all four."""
    for sc in (S.stage1(1, 2), S.stage2(1), S.circle(1)):
        e = U.EmulEnv(sc)
        g = sc.grid
        buf = np.zeros(g.width * g.height * 4, np.uint16)
        assert e.lib.emul_free_rect_field(C.byref(e._st), C.c_void_p(buf.ctypes.data), buf.size) == 0
        f = buf.reshape(g.height, g.width, 4)
        occ = g.dense()
        assert ((f == 0xFFFF).all(axis=2) == occ).all() and ((f == 0xFFFF).any(axis=2) == occ).all()
        sat = np.pad(occ.astype(np.int64).cumsum(0).cumsum(1), ((1, 0), (1, 0)))
        ys, xs = np.nonzero(~occ)
        total = 0.0
        for q in range(4):
            sx, sy = (1 if q & 1 else -1), (1 if q & 2 else -1)
            v = f[ys, xs, q].astype(np.int64)
            ex, ey = v & 255, v >> 8
            assert ex.max() <= 254 and ey.max() <= 254
            xa, xb = xs, xs + sx * ex
            ya, yb = ys, ys + sy * ey
            x0 = np.clip(np.minimum(xa, xb), 0, g.width - 1)
            x1 = np.clip(np.maximum(xa, xb), 0, g.width - 1)
            y0 = np.clip(np.minimum(ya, yb), 0, g.height - 1)
            y1 = np.clip(np.maximum(ya, yb), 0, g.height - 1)
            cnt = sat[y1 + 1, x1 + 1] - sat[y0, x1 + 1] - sat[y1 + 1, x0] + sat[y0, x0]
            assert (cnt == 0).all(), q
            total += (ex + ey).mean()
        assert total / 4 > 4                             # the rectangles are not trivial
