"""Fidelity mode's kernels keep a robot's outline as an anchored 8 x 8 bitmap and find a beam's return from another robot's
outline in closed form (mrca_device.h: outline_bits, outline_bits_meet, ray_outline_entry).  Both replace walks that the
oracles still do -- the list of outline cells (oracle/mrca_oracle_c.c:oc_outline_cells) and the cell-by-cell walk through a
window of marked raster cells (oc_raster_march) -- so they are held against exactly those, on the host, before any GPU time:
the same header compiled with g++ (tests/host_emul).  Bit-exact or it does not ship."""
import ctypes as C

import numpy as np
import pytest

import util as U

RES = [0.2, 0.1, 0.25, 0.13, 0.5]


def _lib():
    lib = U.emul_lib()
    lib.emul_outline_bits.argtypes = [C.c_float] * 4 + [C.c_void_p]
    lib.emul_outline_cells.argtypes = [C.c_float] * 4 + [C.c_void_p]
    lib.emul_outline_meet.argtypes = [C.c_void_p, C.c_void_p]
    lib.emul_outline_span.argtypes = [C.c_float]
    lib.emul_ray_outline.argtypes = [C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_void_p]
    return lib


def _poses(rng, n, extent):
    x = rng.uniform(-extent, extent, n).astype(np.float32)
    y = rng.uniform(-extent, extent, n).astype(np.float32)
    th = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    k = n // 4
    # adversarial: axis-aligned headings, corners on raster lines, centres on raster lines
    th[:k] = rng.choice(np.array([0.0, np.pi / 2, -np.pi / 2, np.pi, np.pi / 4], np.float32), k)
    x[k:2 * k] = np.round(x[k:2 * k] * 5) / 5
    y[k:2 * k] = np.round(y[k:2 * k] * 5) / 5
    return x, y, th


def _bits_to_cells(ob):
    ax, ay, lo, hi = int(ob[0]), int(ob[1]), int(np.uint32(ob[2])), int(np.uint32(ob[3]))
    m = lo | (hi << 32)
    return {(ax + b % 8, ay + b // 8) for b in range(64) if (m >> b) & 1}


def _cells(lib, res, x, y, th):
    buf = np.zeros(64, np.int64)
    n = lib.emul_outline_cells(res, x, y, th, buf.ctypes.data)
    assert n <= 40
    return {(int(v >> 32), int(np.int32(np.uint32(v & 0xFFFFFFFF)))) for v in buf[:n]}


@pytest.mark.parametrize("res", RES)
def test_outline_bitmap_is_the_set_of_outline_cells(res):
    """every pose: the bitmap's cells == the cells of the walk the oracles list; none falls outside the window; the
    occupied columns / rows stay within outline_span(res) (what picks the ray cast's 4 x 4 or 8 x 8 variant)"""
    lib = _lib()
    rng = np.random.default_rng(7)
    span = lib.emul_outline_span(res)
    assert span <= 8 and (res < 0.195 or span <= 4)
    for extent in (10.0, 1500.0):
        xs, ys, ths = _poses(rng, 3000, extent)
        ob = np.zeros(4, np.int32)
        for x, y, th in zip(xs, ys, ths):
            assert lib.emul_outline_bits(res, x, y, th, ob.ctypes.data) == 1
            got = _bits_to_cells(ob)
            assert got == _cells(lib, res, x, y, th)
            assert max(c[0] for c in got) - ob[0] < span and max(c[1] for c in got) - ob[1] < span


@pytest.mark.parametrize("res", RES)
def test_bitmaps_meet_iff_the_cell_lists_intersect(res):
    lib = _lib()
    rng = np.random.default_rng(11)
    n = 4000
    xs, ys, ths = _poses(rng, n, 8.0)
    # partners close enough to share cells about half of the time
    xq = (xs + rng.uniform(-0.9, 0.9, n)).astype(np.float32)
    yq = (ys + rng.uniform(-0.9, 0.9, n)).astype(np.float32)
    tq = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    p, q = np.zeros(4, np.int32), np.zeros(4, np.int32)
    met = 0
    for i in range(n):
        lib.emul_outline_bits(res, xs[i], ys[i], ths[i], p.ctypes.data)
        lib.emul_outline_bits(res, xq[i], yq[i], tq[i], q.ctypes.data)
        want = bool(_cells(lib, res, xs[i], ys[i], ths[i]) & _cells(lib, res, xq[i], yq[i], tq[i]))
        assert bool(lib.emul_outline_meet(p.ctypes.data, q.ctypes.data)) == want
        assert bool(lib.emul_outline_meet(q.ctypes.data, p.ctypes.data)) == want
        met += want
    assert 0.15 * n < met < 0.9 * n
    # far apart (anchors more than a window apart in either direction): never
    lib.emul_outline_bits(res, 0.0, 0.0, 0.3, p.ctypes.data)
    for dx, dy in ((40.0, 0.0), (-40.0, 0.1), (0.0, 40.0), (0.2, -40.0), (8 * res, 0.0), (-8 * res, 0.0)):
        lib.emul_outline_bits(res, dx, dy, 1.0, q.ctypes.data)
        assert lib.emul_outline_meet(p.ctypes.data, q.ctypes.data) == 0


def _rays(rng, n, res):
    """origins, directions and one neighbour outline per ray: neighbours 0.3 .. 7 m away, rays aimed at / past them, with
    the adversarial families a closed form can get wrong: axis-parallel rays, rays through raster corners, origins on
    raster lines, origins inside the neighbour's own cells"""
    ox = rng.uniform(-9, 9, n).astype(np.float32)
    oy = rng.uniform(-9, 9, n).astype(np.float32)
    dist = rng.uniform(0.0, 7.0, n)
    bearing = rng.uniform(-np.pi, np.pi, n)
    nx = (ox + dist * np.cos(bearing)).astype(np.float32)
    ny = (oy + dist * np.sin(bearing)).astype(np.float32)
    nth = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    ang = bearing + rng.normal(0, 0.08, n) * np.minimum(1.0, 1.0 / np.maximum(dist, 0.3))
    k = n // 8
    ang[:k] = rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2], k)                  # axis-parallel
    nx[:k] = (ox[:k] + np.cos(ang[:k]) * dist[:k] + rng.uniform(-0.3, 0.3, k)).astype(np.float32)
    ny[:k] = (oy[:k] + np.sin(ang[:k]) * dist[:k] + rng.uniform(-0.3, 0.3, k)).astype(np.float32)
    ox[k:2 * k] = (np.round(ox[k:2 * k] / res) * res).astype(np.float32)          # origins on raster lines / corners
    oy[2 * k:3 * k] = (np.round(oy[2 * k:3 * k] / res) * res).astype(np.float32)
    oy[k:k + k // 2] = (np.round(oy[k:k + k // 2] / res) * res).astype(np.float32)
    ang[3 * k:4 * k] = rng.choice([np.pi / 4, 3 * np.pi / 4, -np.pi / 4, -3 * np.pi / 4], k)   # through raster corners
    ox[3 * k:4 * k] = (np.round(ox[3 * k:4 * k] / res) * res).astype(np.float32)
    oy[3 * k:4 * k] = (np.round(oy[3 * k:4 * k] / res) * res).astype(np.float32)
    nx[3 * k:4 * k] = (ox[3 * k:4 * k] + np.cos(ang[3 * k:4 * k]) * dist[3 * k:4 * k]).astype(np.float32)
    ny[3 * k:4 * k] = (oy[3 * k:4 * k] + np.sin(ang[3 * k:4 * k]) * dist[3 * k:4 * k]).astype(np.float32)
    dx = np.cos(ang).astype(np.float32)
    dy = np.sin(ang).astype(np.float32)
    dx[:k][np.abs(dx[:k]) < 1e-6] = 0.0
    dy[:k][np.abs(dy[:k]) < 1e-6] = 0.0
    return ox, oy, dx, dy, nx, ny, nth


@pytest.mark.parametrize("res,kw", [(0.2, 4), (0.2, -4), (0.2, 8), (0.1, 8), (0.25, 4), (0.25, -4), (0.13, 8), (0.5, 4), (0.198, 4)])
def test_closed_form_return_equals_the_walk_through_marked_cells(res, kw):
    lib = _lib()
    rng = np.random.default_rng(int(res * 1000) + abs(kw))
    n = 60000
    assert lib.emul_outline_span(res) <= abs(kw)        # (kw = 4: the rearranged 4 x 4 form of the kernel, -4: the plain one)
    ox, oy, dx, dy, nx, ny, nth = _rays(rng, n, res)
    ob = np.zeros((n, 4), np.int32)
    for i in range(n):
        assert lib.emul_outline_bits(res, nx[i], ny[i], nth[i], ob[i].ctypes.data) == 1
    closed = np.zeros(n, np.float32)
    walk = np.zeros(n, np.float32)
    lib.emul_ray_outline(n, kw, res, ox.ctypes.data, oy.ctypes.data, dx.ctypes.data, dy.ctypes.data, ob.ctypes.data, 6.0,
                         closed.ctypes.data, walk.ctypes.data)
    # (the sign of a zero range is not part of the result: a crossing at time -0.0 -- an origin on a raster line -- makes the
    # walk return -0.0, and the scan ring keeps |range|)
    bad = np.nonzero(np.abs(closed).view(np.uint32) != np.abs(walk).view(np.uint32))[0]
    assert bad.size == 0, (bad[:5], closed[bad[:5]], walk[bad[:5]])
    hits = walk < 6.0
    assert 0.2 * n < hits.sum() < 0.95 * n            # the sample exercises both outcomes ...
    assert (walk == 0.0).sum() > 20                   # ... and origins inside a marked cell
