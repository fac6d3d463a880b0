"""Stage ``.world`` + bitmap ingest: the subset of the Stage DSL the reference's
training worlds use -- ``resolution``, ``floorplan( bitmap size pose )``, ``agent( pose [...] )``,
``obstacle( pose size block( points ... ) )`` -- turned into a bit-packed occupancy grid + agent poses.

Rules (libstage behaviour per SURVEY Appendix B):
  * a bitmap pixel is occupied iff gray < 128;
  * the bounding box of the occupied pixels is scaled onto the floorplan ``size``, image row 0 = +y;
  * a cell is occupied iff an occupied pixel's rectangle overlaps it; the outer ring is set (``boundary 1``);
  * polygon obstacles are rescaled so their bounding box equals the model ``size`` centred on the model
    pose; a cell is occupied iff one of 8x8 sample points in it is inside (even-odd) or an outline sample
    falls in it.
Reference files this was written against: worlds/stage1.world, worlds/stage2.world, worlds/circle.world.
"""
import os
import re

import numpy as np

from .scenario import GridData


def parse_world(path):
    """Tiny parser for the Stage DSL subset the three training worlds use."""
    txt = open(path).read()
    txt = re.sub(r"#.*", "", txt)
    out = {"agents": [], "obstacles": []}
    m = re.search(r"^resolution\s+([\d.]+)", txt, re.M)
    out["resolution"] = float(m.group(1))
    m = re.search(r"floorplan\s*\((.*?)\n\)", txt, re.S)
    body = m.group(1)
    out["bitmap"] = re.search(r'bitmap\s+"([^"]+)"', body).group(1)
    out["size"] = [float(v) for v in re.search(r"size\s*\[([^\]]+)\]", body).group(1).split()]
    out["pose"] = [float(v) for v in re.search(r"pose\s*\[([^\]]+)\]", body).group(1).split()]
    for m in re.finditer(r"agent\(\s*pose\s*\[([^\]]+)\]\s*\)", txt):
        out["agents"].append([float(v) for v in m.group(1).split()])
    for m in re.finditer(r"obstacle\(\s*pose\s*\[([^\]]+)\]\s*size\s*\[([^\]]+)\]\s*block\((.*?)z\s*\[",
                         txt, re.S):
        pose = [float(v) for v in m.group(1).split()]
        size = [float(v) for v in m.group(2).split()]
        npts = int(re.search(r"points\s+(\d+)", m.group(3)).group(1))
        pts = {}
        for pm in re.finditer(r"point\[(\d+)\]\s*\[\s*([-+\d.]+)\s+([-+\d.]+)\s*\]", m.group(3)):
            pts[int(pm.group(1))] = (float(pm.group(2)), float(pm.group(3)))  # later def wins
        poly = [pts[i] for i in sorted(pts) if i < npts]
        out["obstacles"].append({"pose": pose, "size": size, "points": poly})
    return out


def raster_bitmap(png, size_m, cell):
    from PIL import Image
    g = np.array(Image.open(png))
    if g.ndim == 3:
        g = g[..., 0]
    occ = g < 128
    rows = np.where(occ.any(1))[0]
    cols = np.where(occ.any(0))[0]
    r0, r1, c0, c1 = rows.min(), rows.max(), cols.min(), cols.max()
    occ = occ[r0:r1 + 1, c0:c1 + 1]
    ph, pw = occ.shape
    sx, sy = size_m
    n_x = int(round(sx / cell))
    n_y = int(round(sy / cell))
    assert abs(n_x * cell - sx) < 1e-9 and abs(n_y * cell - sy) < 1e-9, "size must be a cell multiple"
    grid = np.zeros((n_y, n_x), dtype=bool)
    px_w = sx / pw
    px_h = sy / ph
    rr, cc = np.nonzero(occ)
    # pixel rectangle in metres relative to the lower-left corner of the floorplan
    x_lo = cc * px_w
    x_hi = (cc + 1) * px_w
    y_hi = sy - rr * px_h
    y_lo = sy - (rr + 1) * px_h
    eps = 1e-9
    ix0 = np.clip(np.floor(x_lo / cell + eps).astype(int), 0, n_x - 1)
    ix1 = np.clip(np.ceil(x_hi / cell - eps).astype(int) - 1, 0, n_x - 1)
    iy0 = np.clip(np.floor(y_lo / cell + eps).astype(int), 0, n_y - 1)
    iy1 = np.clip(np.ceil(y_hi / cell - eps).astype(int) - 1, 0, n_y - 1)
    for a, b, c, d in zip(iy0, iy1, ix0, ix1):
        grid[a:b + 1, c:d + 1] = True
    # boundary 1
    grid[0, :] = grid[-1, :] = True
    grid[:, 0] = grid[:, -1] = True
    return grid


def point_in_poly(px, py, poly):
    inside = np.zeros(px.shape, dtype=bool)
    n = len(poly)
    for i in range(n):
        x0, y0 = poly[i]
        x1, y1 = poly[(i + 1) % n]
        cond = (y0 > py) != (y1 > py)
        with np.errstate(divide="ignore", invalid="ignore"):
            xint = (x1 - x0) * (py - y0) / (y1 - y0) + x0
        inside ^= cond & (px < xint)
    return inside


def raster_polygons(grid, obstacles, size_m, cell):
    sx, sy = size_m
    for ob in obstacles:
        pts = np.array(ob["points"], dtype=np.float64)
        lo = pts.min(0)
        hi = pts.max(0)
        scale = np.array(ob["size"][:2]) / (hi - lo)
        ctr = 0.5 * (lo + hi)
        th = np.deg2rad(ob["pose"][3])
        loc = (pts - ctr) * scale
        wx = ob["pose"][0] + loc[:, 0] * np.cos(th) - loc[:, 1] * np.sin(th)
        wy = ob["pose"][1] + loc[:, 0] * np.sin(th) + loc[:, 1] * np.cos(th)
        poly = list(zip(wx + sx / 2, wy + sy / 2))  # grid-relative metres
        bx0 = int(np.floor(min(p[0] for p in poly) / cell)) - 1
        bx1 = int(np.ceil(max(p[0] for p in poly) / cell)) + 1
        by0 = int(np.floor(min(p[1] for p in poly) / cell)) - 1
        by1 = int(np.ceil(max(p[1] for p in poly) / cell)) + 1
        sub = (np.arange(8) + 0.5) / 8.0
        for iy in range(by0, by1 + 1):
            for ix in range(bx0, bx1 + 1):
                px, py = np.meshgrid((ix + sub) * cell, (iy + sub) * cell)
                if point_in_poly(px, py, poly).any():
                    grid[iy, ix] = True
        # outline samples (thin slivers)
        n = len(poly)
        for i in range(n):
            x0, y0 = poly[i]
            x1, y1 = poly[(i + 1) % n]
            for s in np.linspace(0, 1, 65):
                grid[int(np.floor((y0 + s * (y1 - y0)) / cell)), int(np.floor((x0 + s * (x1 - x0)) / cell))] = True
    return grid


def pack_bits(grid):
    h, w = grid.shape
    wpr = (w + 31) // 32
    pad = np.zeros((h, wpr * 32), dtype=bool)
    pad[:, :w] = grid
    bits = pad.reshape(h, wpr, 32)
    words = (bits.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    return words




def load_world(path, cell):
    """Parse ``path`` (a Stage .world file; its bitmap is resolved relative to it) and rasterise at
    ``cell`` metres.  Returns (GridData, agents [[x, y, yaw]] with yaw in (-pi, pi], parsed dict)."""
    w = parse_world(path)
    size = w["size"][:2]
    grid = raster_bitmap(os.path.join(os.path.dirname(os.path.abspath(path)), w["bitmap"]), size, cell)
    if w["obstacles"]:
        grid = raster_polygons(grid, w["obstacles"], size, cell)
    agents = []
    for p in w["agents"]:
        th = np.deg2rad(p[3])
        agents.append([p[0], p[1], float(np.arctan2(np.sin(th), np.cos(th)))])
    cx, cy = (w["pose"] + [0.0, 0.0])[:2]
    return GridData.from_dense(grid, cell, cx - size[0] / 2.0, cy - size[1] / 2.0), agents, w
