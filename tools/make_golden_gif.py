#!/usr/bin/env python3
"""Golden measurements of the ONE libstage-rendered artefact the reference holds: ``doc/stage2.gif``
(README.md:5) -- 23 frames of Stage's own GUI drawing ``worlds/stage2.world`` while the reference's
policy drives the robots.  libstage itself is not in the checkout, so this picture is the only
output of the library the oracle's libstage half can be held against.

What is READ OFF the picture (nothing here comes from this repository's loader or oracle):

  px <-> metre   Stage's ``show_grid`` checkerboard has 1 m squares: the colour edges of the board, least-squares
                 fitted, give the scale (23.92 px/m) and the phase; the axis labels (drawn every 2 m, text origin at
                 the tick) name the lattice line that is 0: the label "0" nearest the picture's centre
                 (``window ( center [0 0] )``, stage2.world:56).  Both axes; y runs up.
  wall mask      pixels of the floorplan's ``color "gray30"`` (stage2.world:29) that sit in a large component.
  markers        every coloured component of robot / obstacle size: centroid, area, mean colour, the extents
                 of its minimum-area bounding rectangle; tracked through the 23 frames by colour
                 (``color "random"``: every model has its own).
  pose label     the GUI prints the selected model's pose -- ``position:2 [x y z a]`` -- in an 6 x 10 bitmap
                 font; the glyph table below was transcribed from this very picture (each digit occurs dozens of
                 times); frames where a wall hides part of the label yield NaN.

Output: tests/golden/stage_gui_stage2.npz (committed; ~20 kB).  tests/test_golden_gui.py holds
mrca/worldfile.py, the shipped maps, the footprint, the pose tables and the tick's kinematic constants against it.

    python tools/make_golden_gif.py            # needs $MRCA_REFERENCE (default /root/reference) and Pillow
"""
import hashlib
import json
import os
import re

import numpy as np
from scipy import ndimage as ndi

REF = os.environ.get("MRCA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "stage_gui_stage2.npz")

# the two colours of the checkerboard, read off the picture's palette
BG_A = np.array([252, 255, 251])
BG_B = np.array([247, 244, 249])

GLYPHS = {  # 5 x 10 cells (one column of spacing either side); '.' hangs one row under the digits' baseline
    "0": "..#.. .#.#. #...# #...# #...# #...# #...# .#.#. ..#.. .....",
    "1": "..#.. .##.. #.#.. ..#.. ..#.. ..#.. ..#.. ..#.. ##### .....",
    "2": ".###. #...# #...# ....# ...#. ..#.. .#... #.... ##### .....",
    "3": "##### ....# ...#. ..#.. .###. ....# ....# #...# .###. .....",
    "4": "...#. ...#. ..##. .#.#. .#.#. #..#. ##### ...#. ...#. .....",
    "5": "##### #.... #.... #.##. ##..# ....# ....# #...# .###. .....",
    "6": ".###. #...# #.... #.... ####. #...# #...# #...# .###. .....",
    "7": "##### ....# ...#. ...#. ..#.. ..#.. .#... .#... .#... .....",
    "8": ".###. #...# #...# #...# .###. #...# #...# #...# .###. .....",
    "9": ".###. #...# #...# #...# .#### ....# ....# #...# .###. .....",
    "-": "..... ..... ..... ..... ##### ..... ..... ..... ..... .....",
    ".": "..... ..... ..... ..... ..... ..... ..... ..#.. .###. ..#..",
}
_T = {k: np.array([[c == "#" for c in row] for row in v.split()]) for k, v in GLYPHS.items()}


def frames(path):
    from PIL import Image
    im = Image.open(path)
    out = []
    for f in range(im.n_frames):
        im.seek(f)
        out.append(np.asarray(im.convert("RGB")).astype(np.int32))
    return out, int(im.info.get("duration", 0))


def _bg_dist(a):
    return np.minimum(np.abs(a - BG_A).max(2), np.abs(a - BG_B).max(2))


def lattice(a):
    """Scale, origin of the metre lattice in pixel-EDGE coordinates (pixel k covers [k, k+1)); x to the right, y DOWN
    in pixels.  Returns (sx, x0, sy, y0) with  x_m = (px - x0) / sx,  y_m = (y0 - py) / sy."""
    dA = np.abs(a - BG_A).max(2)
    dB = np.abs(a - BG_B).max(2)
    bg = np.minimum(dA, dB) <= 6
    isA = (dA < dB) & bg
    isB = (dB <= dA) & bg
    res = []
    for axis in (1, 0):
        if axis == 1:
            e = ((isA[:, :-1] & isB[:, 1:]) | (isB[:, :-1] & isA[:, 1:])).sum(0)
        else:
            e = ((isA[:-1] & isB[1:]) | (isB[:-1] & isA[1:])).sum(1)
        pos = np.where(e > 200)[0] + 1.0            # the edge between pixel k and k+1 is at k+1
        k = np.round((pos - pos[0]) / 24.0)         # 24 is only the rounding guess; the fit decides
        assert len(np.unique(k)) == len(k) and len(k) >= 30
        s, p0 = np.polyfit(k, pos, 1)
        res.append((s, p0, np.abs(pos - (p0 + s * k)).max()))
    (sx, px0, ex), (sy, py0, ey) = res
    assert ex < 0.75 and ey < 0.75, (ex, ey)
    # which lattice line is 0: the axis label "0" nearest the centre; the text origin (lower-left) sits on the tick
    txt = (a.sum(2) > 120) & (a.sum(2) < 200) & ((a.max(2) - a.min(2)) < 20)       # the labels' dark gray (50, 52, 49)
    t0 = _T["0"][:9]
    H, W = a.shape[:2]
    best = None
    for y in range(H // 2 - 30, H // 2 + 30):
        for x in range(W // 2 - 30, W // 2 + 30):
            win = txt[y:y + 9, x - 1:x + 6]
            if win.shape == (9, 7) and (win[:, 1:6] == t0).all() and not win[:, 0].any() and not win[:, 6].any():
                d = abs(x - W / 2) + abs(y - H / 2)
                if best is None or d < best[0]:
                    best = (d, x, y + 9)
    assert best is not None, "no '0' label near the centre"
    _, lx, ly = best
    kx = np.round((lx - px0) / sx)
    ky = np.round((ly - py0) / sy)
    x0 = px0 + sx * kx
    y0 = py0 + sy * ky
    assert abs(x0 - lx) < 1.0 and abs(y0 - ly) < 1.0, (x0, lx, y0, ly)
    return float(sx), float(x0), float(sy), float(y0)


def wall_mask(a):
    gray = (np.abs(a[..., 0] - a[..., 1]) < 12) & (np.abs(a[..., 1] - a[..., 2]) < 12) & (a.max(2) < 140) & (_bg_dist(a) > 14)
    lab, n = ndi.label(gray, structure=np.ones((3, 3)))
    area = ndi.sum(gray, lab, np.arange(1, n + 1))
    keep = np.zeros(n + 1, bool)
    keep[1:] = area >= 150                           # labels, ticks and the thin frame line are smaller or thinner
    m = keep[lab]
    # the 1-px frame line around the floorplan joins the big component: drop pixels whose 3x3 neighbourhood is mostly bg
    dens = ndi.uniform_filter(m.astype(np.float32), 3)
    return m & (dens > 0.45)


def frame_line(a):
    """The thin rectangle Stage draws around the floorplan (its ``size`` box): darkness-weighted position of the line on
    each side (pixel-edge coordinates), median over the rows / columns where nothing else is near.
    Returns (left, right, top, bottom)."""
    dark = np.clip(700.0 - a.sum(2), 0, None)          # background is ~750, the line ~110
    H, W = dark.shape
    out = []
    for side in range(4):
        v = []
        for k in range(40, (H if side < 2 else W) - 40, 7):
            strip = dark[k, :14] if side == 0 else dark[k, W - 14:] if side == 1 else dark[:14, k] if side == 2 else dark[H - 14:, k]
            if strip.sum() == 0 or (strip > 0).sum() > 3:      # a wall / label here: not the bare line
                continue
            pos = (strip * (np.arange(14) + 0.5)).sum() / strip.sum()
            v.append(pos + (0 if side in (0, 2) else (W - 14 if side == 1 else H - 14)))
        assert len(v) >= 10, (side, len(v))
        out.append(float(np.median(v)))
    return out


def _min_area_rect(xs, ys):
    """Extents (long, short) of the minimum-area rectangle around pixel centres, +1 for the pixels' own size."""
    pts = np.stack([xs, ys], 1).astype(np.float64)
    best = None
    for ang in np.deg2rad(np.arange(0.0, 90.0, 1.0)):
        c, s = np.cos(ang), np.sin(ang)
        u = pts @ np.array([c, s])
        v = pts @ np.array([-s, c])
        w, h = u.max() - u.min() + 1.0, v.max() - v.min() + 1.0
        if best is None or w * h < best[0]:
            best = (w * h, max(w, h), min(w, h), ang)
    return best[1], best[2]


def markers(a):
    fg = _bg_dist(a) > 14
    lab, n = ndi.label(fg, structure=np.ones((3, 3)))
    out = []
    for i, s in enumerate(ndi.find_objects(lab)):
        m = lab[s] == i + 1
        area = int(m.sum())
        if not (60 <= area <= 260) or m.shape[0] > 22 or m.shape[1] > 22:
            continue
        px = a[s][m]
        if (px.max(1) - px.min(1)).mean() <= 20:      # gray: a wall fragment, not a "random" colour
            continue
        ys, xs = np.nonzero(m)
        lo, sh = _min_area_rect(xs, ys)
        out.append(dict(cx=xs.mean() + s[1].start + 0.5, cy=ys.mean() + s[0].start + 0.5, area=area,
                        col=px.mean(0), long=lo, short=sh,
                        bbox=(s[1].start, s[0].start, s[1].stop, s[0].stop)))
    return out


_NUM = re.compile(r"^-?\d+\.\d\d$")


def read_label(a):
    """The four numbers of ``position:N [x y z a]``; NaN for a number any part of which a wall hides.

    The line is cut into its 6-px cells.  A cell is a GLYPH (exact match of a 5 x 10 template with a clear column either
    side), BLANK (every pixel is checkerboard background), a BAR (a column of >= 9 text pixels: a bracket) or OTHER
    (a wall over the text, an unknown glyph).  Numbers are read outward from the brackets -- x, y, ... rightward from
    '[', a, z, ... leftward from ']' -- glyph runs separated by exactly one BLANK cell, and the walk stops at the first
    OTHER cell, so a partly hidden number can never be mistaken for a shorter one."""
    txt = a.sum(2) < 30                                # the label is pure black (1, 2, 2); axis labels are dark gray
    H, W = txt.shape
    hits = {}
    ys, xs = np.nonzero(txt)
    for ch, t in _T.items():
        ty, tx = np.argwhere(t)[0]
        for y, x in set(zip((ys - ty).tolist(), (xs - tx).tolist())):
            if y < 0 or x < 1 or y + 10 > H or x + 6 > W:
                continue
            win = txt[y:y + 10, x - 1:x + 6]
            if (win[:, 1:6] == t).all() and not win[:, 0].any() and not win[:, 6].any():
                hits[(y, x)] = ch
    rows = [y for (y, x), ch in hits.items() if ch.isdigit()]
    if not rows:
        return [np.nan] * 4
    y1 = max(set(rows), key=rows.count)                # the label's line: the most common digit row
    on = sorted(x for (y, x) in hits if y == y1)
    phase = on[0] % 6
    band = txt[y1 - 1:y1 + 10]
    cols = np.nonzero(band.any(0))[0]
    bgd = _bg_dist(a)[y1 - 1:y1 + 10]
    first = (cols.min() - phase) // 6
    last = (cols.max() - phase) // 6
    cells = []
    for k in range(first - 1, last + 2):
        x = phase + 6 * k
        if (y1, x) in hits:
            cells.append(hits[(y1, x)])
        elif not band[:, max(x - 1, 0):x + 5].any() and ((bgd[1:, max(x - 1, 0):x + 5] > 14).any(1).sum() <= 1):
            # no text, and nothing but checkerboard in the digits' ten rows -- bar one row (a wall's edge grazing the
            # line cannot hide a glyph: the glyph's other rows would still show as text)
            cells.append(" ")
        elif band[:, max(x - 1, 0):x + 5].sum(0).max() >= 9 and \
                ((bgd[1:, max(x - 1, 0):x + 5] > 14) & ~band[1:, max(x - 1, 0):x + 5]).any(1).sum() <= 1:
            cells.append("|")
        else:
            cells.append("?")
    line = "".join(cells)
    nums = [np.nan] * 4

    def walk(seq, slots, rev=False):
        for slot in slots:
            # a number ends at a blank cell, at a bracket -- or, walking leftward, at its own sign
            m = re.match(r"^([-.\d]+)([ |])", seq) or (rev and re.match(r"^([.\d]+-)(.)", seq))
            if not m:
                return
            yield slot, m.group(1)
            if m.group(2) != " ":
                return
            seq = seq[m.end():]

    i = line.find("|")
    j = line.rfind("|")
    if i >= 0 and (i != j or re.match(r"^-?\d", line[i + 1:])):
        for slot, tok in walk(line[i + 1:], range(4)):
            if _NUM.match(tok):
                nums[slot] = float(tok)
    if j > 0 and (j != i or re.match(r"^\d", line[j - 1])):
        for slot, tok in walk(line[:j][::-1], range(3, -1, -1), rev=True):
            if _NUM.match(tok[::-1]):
                nums[slot] = float(tok[::-1])
    return nums


def main():
    path = os.path.join(REF, "doc", "stage2.gif")
    fr, duration = frames(path)
    sx, x0, sy, y0 = lattice(fr[0])
    for a in fr[1:]:                                   # the camera does not move
        l = lattice(a)
        assert max(abs(l[0] - sx), abs(l[2] - sy)) < 0.01 and max(abs(l[1] - x0), abs(l[3] - y0)) < 0.25, l
    wm = wall_mask(fr[0])
    for a in fr[1:]:
        wm &= wall_mask(a)                             # static: a robot never covers a wall
    per = [markers(a) for a in fr]
    n = len(per[0])
    assert all(len(p) == n for p in per), [len(p) for p in per]
    # track by colour + nearness from frame 0 on
    order = [list(range(n))]
    for f in range(1, len(per)):
        prev = [per[f - 1][i] for i in order[-1]]
        cost = np.array([[np.abs(p["col"] - q["col"]).sum() + 2.0 * np.hypot(p["cx"] - q["cx"], p["cy"] - q["cy"])
                          for q in per[f]] for p in prev])
        from scipy.optimize import linear_sum_assignment
        r, c = linear_sum_assignment(cost)
        assert (r == np.arange(n)).all()
        order.append(list(c))
    F = len(per)
    xy = np.zeros((F, n, 2))
    ext = np.zeros((F, n, 2))
    area = np.zeros((F, n))
    bbox = np.zeros((F, n, 4))
    col = np.zeros((n, 3))
    for f in range(F):
        for j, i in enumerate(order[f]):
            m = per[f][i]
            xy[f, j] = ((m["cx"] - x0) / sx, (y0 - m["cy"]) / sy)
            ext[f, j] = (m["long"], m["short"])
            area[f, j] = m["area"]
            bbox[f, j] = m["bbox"]
            if f == 0:
                col[j] = m["col"]
    labels = np.array([read_label(a) for a in fr], np.float64)
    # the big markers (the polygon obstacles): their pixel masks in a 28 x 28 window anchored at bbox - 2
    big = [j for j in range(n) if area[0, j] >= 150 and np.ptp(xy[:, j], axis=0).max() < 0.05]
    omask = np.zeros((len(big), 28, 28), bool)
    oorg = np.zeros((len(big), 2))
    fg = _bg_dist(fr[0]) > 14
    for k, j in enumerate(big):
        bx0, by0 = int(bbox[0, j, 0]) - 2, int(bbox[0, j, 1]) - 2
        omask[k] = fg[by0:by0 + 28, bx0:bx0 + 28]
        oorg[k] = (bx0, by0)
    frame = frame_line(fr[0])
    meta = dict(source="doc/stage2.gif (README.md:5)", sha256=hashlib.sha256(open(path, "rb").read()).hexdigest(),
                frames=F, frame_duration_ms=duration, shape=list(fr[0].shape[:2]),
                note="pixel-edge coordinates: x_m = (px - origin_px[0]) / px_per_m[0], y_m = (origin_px[1] - py) / px_per_m[1]")
    np.savez_compressed(OUT, px_per_m=np.array([sx, sy]), origin_px=np.array([x0, y0]),
                        wall_mask=np.packbits(wm, axis=1), wall_shape=np.array(wm.shape),
                        marker_xy_m=xy, marker_extent_px=ext, marker_area_px=area, marker_bbox_px=bbox, marker_rgb=col,
                        label_pose=labels, obstacle_marker=np.array(big), obstacle_mask=omask, obstacle_mask_origin_px=oorg,
                        frame_line_px=np.array(frame), meta=json.dumps(meta))
    print("px/m %.4f %.4f  origin px (%.2f, %.2f)  wall px %d  markers %d  label frames read %d / %d -> %s (%d bytes)"
          % (sx, sy, x0, y0, wm.sum(), n, np.isfinite(labels).all(1).sum(), F, os.path.relpath(OUT), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
