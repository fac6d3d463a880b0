"""The libstage half held against the one libstage-rendered artefact the reference ships: ``doc/stage2.gif``
(README.md:5), Stage's own GUI drawing ``worlds/stage2.world`` for 23 frames while a trained policy drives 44 robots.
``tools/make_golden_gif.py`` measured the picture (px <-> metre from the GUI's 1 m checkerboard and axis labels, the wall
mask, every marker's centre / extents per frame, the numeric pose label of ``position:2``) into
``tests/golden/stage_gui_stage2.npz``; nothing in that file comes from this repository.  These tests hold the loader, the
shipped map, the footprint, the pose tables and the tick's constants against it.

What this PINS of SURVEY App. B / DESIGN 3 (until now "uncited recollection"):
  (v)   bitmap -> world: the occupied bounding box of the bitmap is scaled onto ``size``, image row 0 is +y, the floorplan
        is centred on its ``pose``                                   test_floorplan_box / test_wall_edges / test_wrong_maps
  App.B polygon obstacles are rescaled so their bounding box equals ``size [0.7 0.7]`` centred on ``pose``
                                                                     test_obstacles_*
  a3    the robot footprint is the 0.44 x 0.38 m ``size`` rectangle centred on the pose (``origin [0 0 0 0]``)
                                                                     test_robot_footprint
  a10   ``cmd_pose`` teleports: at episode start every robot stands ON its table pose (model/utils.py:41-52) and the GUI
        prints exactly -18.00 11.50 0.00 0.00 for robot 2          test_reset_poses_are_the_table
  a9    a robot counts as arrived inside 0.5 m of ITS table goal (model/utils.py:54-62, stage_world2.py:34) -- the picture's
        driver zeroes v from then on (circle_test.py:64-66), and every robot's final rest is 0.2 .. 0.5 m from its own goal
                                                                     test_robots_halt_inside_goal_size
  a2    one tick moves a robot v * 0.1 s along its heading, v <= 1: frames are 10 ticks apart, the fastest robots move
        exactly 1.00 m per frame, nobody more; headings are degrees CCW from +x and motion is along the heading
                                                                     test_tick_displacement_quantum / test_motion_follows_heading
What it does NOT pin (still "parity unpinned (libstage)"): the order of operations inside one tick, Stage's raster
collision test, and the quantisation of ranges to raster cells -- nothing in a GUI picture shows them.
"""
import json
import os

import numpy as np
import pytest

import util as U
from util import S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage_gui_stage2.npz")
REF = os.environ.get("MRCA_REFERENCE", "/root/reference")
HAVE_WORLD = os.path.exists(os.path.join(REF, "worlds", "stage2.world"))
needs_world = pytest.mark.skipif(not HAVE_WORLD, reason="the reference's worlds/ are not reachable")

# the picture is a down-scaled screen capture: edges are blurred over ~1 px, outlines add ~0.5 px per side
PX_TOL = 1.5


@pytest.fixture(scope="module")
def gui():
    d = np.load(GOLD)
    g = {k: d[k] for k in d.files}
    H, W = g["wall_shape"]
    g["walls"] = np.unpackbits(g["wall_mask"], axis=1)[:, :W].astype(bool)
    g["meta"] = json.loads(str(g["meta"]))
    g["sx"], g["sy"] = g["px_per_m"]
    g["x0"], g["y0"] = g["origin_px"]
    area = g["marker_area_px"][0]
    g["obst"] = list(g["obstacle_marker"])
    g["robots"] = [j for j in range(len(area)) if j not in g["obst"]]
    return g


def _render(gui, grid, warp=None):
    """The occupancy grid sampled at every picture pixel's centre (True = occupied)."""
    H, W = gui["walls"].shape
    xs = (np.arange(W) + 0.5 - gui["x0"]) / gui["sx"]
    ys = (gui["y0"] - (np.arange(H) + 0.5)) / gui["sy"]
    if warp is not None:
        xs, ys = warp(xs, ys)
    dn = grid.dense()
    ix = np.floor((xs - grid.x0) / grid.cell).astype(int)
    iy = np.floor((ys - grid.y0) / grid.cell).astype(int)
    okx = (ix >= 0) & (ix < grid.width)
    oky = (iy >= 0) & (iy < grid.height)
    out = np.zeros((H, W), bool)
    out[np.ix_(oky, okx)] = dn[np.ix_(iy[oky], ix[okx])]
    return out


def _edges(line):
    d = np.diff(line.astype(np.int8))
    return np.where(d == 1)[0] + 1.0, np.where(d == -1)[0] + 1.0


def _clean_zone(gui):
    """Picture pixels where the wall mask can be compared: away from the axis-label bands (their glyphs touch walls),
    from the three polygon obstacles' block (the picture's world holds three of stage2.world's nine) and from the
    1-px frame line."""
    H, W = gui["walls"].shape
    ok = np.ones((H, W), bool)
    x0, y0 = int(gui["x0"]), int(gui["y0"])
    ok[y0 - 14:y0 + 6, :] = False                       # the x axis' labels
    ok[:, x0 - 4:x0 + 22] = False                       # the y axis' labels
    xa = int(gui["x0"] + 8.5 * gui["sx"])
    ya, yb = int(gui["y0"] + 5.5 * gui["sy"]), int(gui["y0"] + 12.5 * gui["sy"])
    ok[ya:yb, xa:] = False                              # obstacles live in x 8.5..20, y -12.5..-5.5
    ok[:12, :] = ok[-10:, :] = False
    ok[:, :3] = ok[:, -6:] = False
    return ok


def _edge_offsets(gui, R):
    """Signed offsets (map - picture, px) of every stable wall edge on scan lines 16 px apart, both directions."""
    wm = gui["walls"]
    ok = _clean_zone(gui)
    H, W = wm.shape
    offs, lines = [], 0
    for axis in (0, 1):
        n = H if axis == 0 else W
        for k in range(20, n - 20, 16):
            a, b, z = (wm[k], R[k], ok[k]) if axis == 0 else (wm[:, k], R[:, k], ok[:, k])
            a3 = [(wm[k + s] if axis == 0 else wm[:, k + s]) for s in (-3, 3)]
            used = False
            for pol, (ea, eb) in enumerate(zip(_edges(a), _edges(b))):
                for e in ea:
                    e = int(e)
                    if not z[max(e - 3, 0):e + 3].all():
                        continue
                    # a stable edge: the lines 3 px either side show the same edge within a pixel (not a bar's end grazed)
                    if not all(np.abs(_edges(l)[pol] - e).min(initial=99) <= 1 for l in a3):
                        continue
                    offs.append((eb[np.argmin(np.abs(eb - e))] - e) if len(eb) else 99.0)
                    used = True
            lines += used
    return np.array(offs), lines


def test_golden_is_what_the_tool_measures(gui):
    m = gui["meta"]
    assert m["frames"] == 23 and m["shape"] == [972, 970] and m["source"].startswith("doc/stage2.gif")
    # one isotropic scale, fitted independently per axis from the checkerboard
    assert abs(gui["sx"] / gui["sy"] - 1.0) < 1e-3 and 23.5 < gui["sx"] < 24.5
    assert len(gui["robots"]) == 44 and len(gui["obst"]) == 3        # stage2.world:113-165: 44 agents
    if os.path.exists(os.path.join(REF, "doc", "stage2.gif")):
        import hashlib
        assert hashlib.sha256(open(os.path.join(REF, "doc", "stage2.gif"), "rb").read()).hexdigest() == m["sha256"]


def test_pose_label_agrees_with_the_marker(gui):
    """Two independent readings of one robot's position: the GUI's printed pose (metres, Stage's own numbers) and the
    marker's centroid through the checkerboard / axis-label mapping.  Pins the mapping itself: scale, origin, y up."""
    lab = gui["label_pose"]
    xy = gui["marker_xy_m"]
    sel = int(np.argmax(gui["marker_area_px"][0] * np.isin(np.arange(xy.shape[1]), gui["robots"])))   # the selected robot is drawn with a highlight box
    n = 0
    for f in range(len(lab)):
        if np.isfinite(lab[f, :2]).all():
            d = np.abs(xy[f, sel] - lab[f, :2]) * gui["sx"]
            assert d.max() <= PX_TOL, (f, xy[f, sel], lab[f, :2])
            n += 1
    assert n >= 12
    assert np.nanmax(np.abs(lab[:, 2])) == 0.0                       # z


def test_floorplan_box(gui):
    """``size [40 40]`` centred on ``pose [0 0]`` (stage2.world:42-48): the rectangle Stage draws around the floorplan
    sits at +-20 m, and the outermost wall pixels start ON it -- the occupied bounding box of the bitmap, not the whole
    1000 x 800 image, is what gets scaled onto ``size``."""
    l, r, t, b = gui["frame_line_px"]
    box_m = np.array([(l - gui["x0"]) / gui["sx"], (r - gui["x0"]) / gui["sx"], (gui["y0"] - t) / gui["sy"], (gui["y0"] - b) / gui["sy"]])
    assert np.abs(box_m - [-20, 20, 20, -20]).max() * gui["sx"] <= 1.0, box_m
    wm = gui["walls"].copy()
    x0, y0 = int(gui["x0"]), int(gui["y0"])
    wm[y0 - 14:y0 + 6, :] = False                      # the axis labels' bands (a "20" sits outside the box at the top)
    wm[:, x0 - 4:x0 + 22] = False
    ys, xs = np.nonzero(wm)
    ext = np.array([xs.min(), xs.max() + 1, ys.min(), ys.max() + 1], float)
    assert np.abs(ext - [l, r, t, b]).max() <= 2.0, (ext, gui["frame_line_px"])
    grid = S.load_map("stage2_testenv")
    assert (grid.x0, grid.y0, grid.width * grid.cell, grid.height * grid.cell) == (-20.0, -20.0, 40.0, 40.0)


@pytest.mark.parametrize("which", ["shipped 0.05 m map", pytest.param("loader at 0.0125 m", marks=needs_world)])
def test_wall_edges(gui, which):
    """Every stable wall edge of the picture, on >= 80 scan lines in x and y, against the loader's occupancy sampled at
    the same pixels.  Tolerance: 1.5 px of picture blur + one raster cell (a cell is occupied if a wall pixel touches it,
    so the raster may be fatter by up to a cell per side, never thinner)."""
    if which.startswith("shipped"):
        grid = S.load_map("stage2_testenv")
    else:
        from mrca import worldfile
        grid = worldfile.load_world(os.path.join(REF, "worlds", "stage2.world"), 0.0125)[0]
    offs, lines = _edge_offsets(gui, _render(gui, grid))
    tol = PX_TOL + grid.cell * gui["sx"]
    assert lines >= 80 and len(offs) >= 300
    assert np.abs(offs).max() <= tol, (np.abs(offs).max(), tol)
    assert np.mean(np.abs(offs) <= 1.0) >= 0.95
    assert abs(np.mean(offs)) <= 0.6                                  # no systematic shift: the origin and the flip are right


@needs_world
def test_wrong_maps_fail(gui):
    """The test above has teeth: the same comparison rejects the three mistakes App. B's recollection could have made --
    image row 0 = -y, no flip of x / y mix-up, and scaling the whole image (not its occupied bounding box) onto ``size``."""
    from mrca import worldfile
    grid = worldfile.load_world(os.path.join(REF, "worlds", "stage2.world"), 0.025)[0]
    good, _ = _edge_offsets(gui, _render(gui, grid))
    assert np.abs(good).max() <= PX_TOL + 1.0
    from PIL import Image
    img = np.array(Image.open(os.path.join(REF, "worlds", "testenv.png")))
    img = img[..., 0] if img.ndim == 3 else img
    rows = np.where((img < 128).any(1))[0]
    cols = np.where((img < 128).any(0))[0]
    ih, iw = img.shape

    def whole_image(xs, ys):       # metre coordinate under the hypothesis "whole image -> size"  ->  coordinate in the bbox map
        c = (xs + 20.0) / 40.0 * iw
        r = (20.0 - ys) / 40.0 * ih
        return (-20.0 + 40.0 * (c - cols.min()) / (cols.max() + 1 - cols.min()),
                20.0 - 40.0 * (r - rows.min()) / (rows.max() + 1 - rows.min()))

    for name, warp in [("row 0 = -y", lambda xs, ys: (xs, -ys)), ("x mirrored", lambda xs, ys: (-xs, ys)),
                       ("whole image scaled", whole_image)]:
        offs, _ = _edge_offsets(gui, _render(gui, grid, warp))
        assert np.mean(np.abs(offs) <= PX_TOL + 1.0) < 0.7, name


def _obstacle_boxes_m(gui):
    out = []
    for k, j in enumerate(gui["obst"]):
        m = gui["obstacle_mask"][k]
        ox, oy = gui["obstacle_mask_origin_px"][k]
        ys, xs = np.nonzero(m)
        out.append(((xs.min() + ox - gui["x0"]) / gui["sx"], (xs.max() + 1 + ox - gui["x0"]) / gui["sx"],
                    (gui["y0"] - (ys.max() + 1 + oy)) / gui["sy"], (gui["y0"] - (ys.min() + oy)) / gui["sy"]))
    return np.array(out)


def test_obstacles_are_size_boxes_on_their_poses_in_the_shipped_map(gui):
    """Three of stage2.world's nine polygon obstacles are in the picture.  Each one's drawn bounding box is 0.7 x 0.7 m
    (``size [0.7 0.7 0.8]``, not the 1 x 1 or 0.45 x 0.45 of its ``block`` points) centred on an integer ``pose`` -- and
    the shipped map has its occupied cells exactly there."""
    boxes = _obstacle_boxes_m(gui)
    grid = S.load_map("stage2_testenv")
    dn = grid.dense()
    for (xa, xb, ya, yb) in boxes:
        w, h = xb - xa, yb - ya
        # the outline is drawn centred on the polygon's edge: + <= 1 px per side
        assert 0.7 - 0.5 / gui["sx"] <= w <= 0.7 + 2.5 / gui["sx"] and 0.7 - 0.5 / gui["sx"] <= h <= 0.7 + 2.5 / gui["sx"], (w, h)
        cx, cy = (xa + xb) / 2, (ya + yb) / 2
        assert abs(cx - round(cx)) * gui["sx"] <= PX_TOL and abs(cy - round(cy)) * gui["sx"] <= PX_TOL, (cx, cy)
        px, py = round(cx), round(cy)
        # occupied cells of the map within 1 m of the pose: their bounding box is the 0.7 m box (+ <= 1 cell per side)
        i0, i1 = int((px - 1 - grid.x0) / grid.cell), int((px + 1 - grid.x0) / grid.cell)
        j0, j1 = int((py - 1 - grid.y0) / grid.cell), int((py + 1 - grid.y0) / grid.cell)
        sub = dn[j0:j1, i0:i1]
        jj, ii = np.nonzero(sub)
        mx = ((ii.min() + i0) * grid.cell + grid.x0, (ii.max() + 1 + i0) * grid.cell + grid.x0)
        my = ((jj.min() + j0) * grid.cell + grid.y0, (jj.max() + 1 + j0) * grid.cell + grid.y0)
        assert abs(mx[0] - (px - 0.35)) <= grid.cell + 1e-9 and abs(mx[1] - (px + 0.35)) <= grid.cell + 1e-9, mx
        assert abs(my[0] - (py - 0.35)) <= grid.cell + 1e-9 and abs(my[1] - (py + 0.35)) <= grid.cell + 1e-9, my


@needs_world
def test_obstacle_shapes_follow_the_rescale_rule(gui):
    """Shape, not only box: each drawn obstacle against its ``block`` polygon after the loader's rule (bounding box of the
    points -> ``size``, centred on ``pose``), filled at the picture's pixels.  IoU >= 0.75 with the right polygon at the right
    pose (outline and blur cost the rest); using the raw ``points`` as metres (no rescale) falls under 0.6."""
    from mrca import worldfile
    w = worldfile.parse_world(os.path.join(REF, "worlds", "stage2.world"))
    boxes = _obstacle_boxes_m(gui)
    seen = set()
    for k, (xa, xb, ya, yb) in enumerate(boxes):
        pose = (round((xa + xb) / 2), round((ya + yb) / 2))
        ob = [o for o in w["obstacles"] if (o["pose"][0], o["pose"][1]) == pose]
        assert len(ob) == 1, pose
        seen.add(pose)
        pts = np.array(ob[0]["points"], float)
        lo, hi = pts.min(0), pts.max(0)
        m = gui["obstacle_mask"][k]
        ox, oy = gui["obstacle_mask_origin_px"][k]
        sub = (np.arange(4) + 0.5) / 4
        yy, xx = np.meshgrid(np.arange(28), np.arange(28), indexing="ij")
        ious = []
        for scaled in (True, False):
            loc = (pts - 0.5 * (lo + hi)) * (np.array(ob[0]["size"][:2]) / (hi - lo)) if scaled else pts - 0.5 * (lo + hi)
            poly = [(pose[0] + p[0], pose[1] + p[1]) for p in loc]
            fill = np.zeros((28, 28), bool)
            for sy_ in sub:
                for sx_ in sub:
                    X = (xx + sx_ + ox - gui["x0"]) / gui["sx"]
                    Y = (gui["y0"] - (yy + sy_ + oy)) / gui["sy"]
                    fill |= worldfile.point_in_poly(X, Y, poly)
            ious.append((fill & m).sum() / (fill | m).sum())
        assert ious[0] >= 0.75, (pose, ious)
        if abs((hi - lo).max() - 0.7) > 0.15:                        # (a block drawn at 0.7 already would not tell)
            assert ious[1] < 0.6, (pose, ious)
    assert len(seen) == 3


def test_robot_footprint(gui):
    """``size [0.44 0.38 0.22]``, ``origin [0 0 0 0]`` (stage2.world:83-88): the markers' minimum-area rectangles and their
    pixel areas.  An outline of ~1 px is part of a marker, so extents may exceed the footprint by up to 1.5 px and never
    fall short by more than 0.5 px; area within 8 %."""
    rob = [j for j in gui["robots"] if gui["marker_area_px"][0, j] < 150]          # (not the highlighted one)
    ext = gui["marker_extent_px"][:, rob]
    lo, sh = np.median(ext[..., 0]), np.median(ext[..., 1])
    assert -0.5 <= lo - 0.44 * gui["sx"] <= 1.5, lo
    assert -0.5 <= sh - 0.38 * gui["sx"] <= 1.5, sh
    area_m2 = np.median(gui["marker_area_px"][:, rob]) / (gui["sx"] * gui["sy"])
    assert abs(area_m2 / (0.44 * 0.38) - 1.0) <= 0.08, area_m2
    # what the oracle collides: the same rectangle
    assert (2 * U.O.HALF_LEN, 2 * U.O.HALF_WID) == (0.44, 0.38)


def _identities(gui):
    """marker -> table index: where each robot stands in the picture's last frame (the run's FIRST instant, see below)."""
    tb = S.load_tables()["stage2"]
    init = np.array(tb["init_pose"])[:, :2]
    P = gui["marker_xy_m"][-1][gui["robots"]]
    D = np.linalg.norm(P[:, None] - init[None], axis=2)
    return D.argmin(1), D.min(1), tb


def test_reset_poses_are_the_table(gui):
    """The GIF runs BACKWARDS in simulation time (robot 2 drives from x = -18 to x = -7 as the frames go 22 -> 0, its
    printed heading ~0 deg = facing +x): the last two frames are the instant after ``reset_pose``.  Every one of the 44
    robots stands within half a pixel of its own ``get_init_pose`` entry (model/utils.py:41-52 == stage2.world:113-165)
    and Stage prints robot 2's pose as exactly the table's numbers: ``cmd_pose`` is a teleport (stageros.cpp:282-296)."""
    ident, dist, tb = _identities(gui)
    assert sorted(ident) == list(range(44))
    assert dist.max() * gui["sx"] <= 0.75, dist.max()
    assert list(gui["label_pose"][-1]) == [-18.0, 11.5, 0.0, 0.0] == list(gui["label_pose"][-2])
    assert tb["init_pose"][2][:2] == [-18.0, 11.5] and abs(tb["init_pose"][2][2]) < 1e-12


def test_robots_halt_inside_goal_size(gui):
    """The picture's driver zeroes a robot's linear speed once ``get_reward_and_terminate`` reports it terminal
    (circle_test.py:64-66) and nobody crashes in this run, so where a robot comes to rest is where 'Reach Goal' fired:
    ``distance < goal_size`` = 0.5 m from ITS OWN ``get_goal_point`` entry (stage_world2.py:34,183-186).  All 34 table-goal
    robots rest 0.2 .. 0.5 m from their goals at the end (frame 0) -- never at the goal itself, never outside the disc."""
    ident, _, tb = _identities(gui)
    goal = np.array(tb["goal_point"])
    assert U.O.GOAL_RADIUS == 0.5
    xy = gui["marker_xy_m"][:, gui["robots"]]
    n = 0
    for r, i in enumerate(ident):
        if i < len(goal):
            d = np.linalg.norm(xy[0, r] - goal[i])
            assert 0.15 < d < 0.5 + PX_TOL / gui["sx"], (i, d)
            # in simulation order (frames reversed): from the first frame that shows it inside the disc it moves at most
            # one more tick's worth (the reference's loop reads the pose a callback late) and then stays put
            dist = np.linalg.norm(xy[::-1, r] - goal[i], axis=1)
            k = int(np.argmax(dist < 0.5))
            assert k >= 1 and dist[k - 1] > 0.5
            after = np.linalg.norm(xy[::-1, r][k:] - xy[::-1, r][k], axis=1)
            assert after.max() <= 0.1 + 0.035, (i, after)
            assert np.linalg.norm(np.diff(xy[::-1, r][k + 1:], axis=0), axis=1).max(initial=0.0) * gui["sx"] <= 0.75
            n += 1
    assert n == 34


def test_tick_displacement_quantum(gui):
    """``interval_sim`` 100 ms x v <= 1 m/s, no acceleration limit (the bounds are commented out, stage2.world:102-104):
    between two frames nobody moves farther than K x 0.1 m for ONE integer K per frame pair, and with the trained
    policy saturating v at 1.0 the fastest robots move exactly that: 1.00 m (K = 10) for 15 of the 22 pairs, 0.90, 0.80 and
    0.10 m for others.  (A weak bound: the frame spacing in ticks is inferred from the same displacements; what it pins
    is that the largest displacement is a multiple of 0.1 m to within 2 cm, frame after frame, and that dozens of
    robots share it exactly -- v_max * dt is a hard ceiling, not a mean.)"""
    assert U.O.DT == 0.1                                              # what the oracle (and the kernels' kDt) integrate with
    xy = gui["marker_xy_m"][:, gui["robots"]]
    d = np.linalg.norm(np.diff(xy, axis=0), axis=2)                   # [22, 44] metres per frame
    top = d.max(1)
    moving = top > 0.05
    assert moving.sum() >= 18
    k = np.round(top[moving] / 0.1)
    assert np.abs(top[moving] - 0.1 * k).max() <= 0.035, top            # centroid noise is ~0.5 px = 0.02 m
    assert (k <= 10).all() and (k == 10).sum() >= 12
    # the ceiling is shared: over the 10-tick pairs >= 150 robot-frames sit within 2 cm of it, none beyond
    ten = [f for f in np.where(moving)[0] if round(top[f] / 0.1) == 10]
    assert sum(int((np.abs(d[f] - 1.0) <= 0.02).sum()) for f in ten) >= 150
    assert all((d[f] <= 1.035).all() for f in ten)


def test_motion_follows_heading(gui):
    """Robot 2's printed headings (degrees, CCW from +x: it faces +x at 0.00 and drives towards +x) against the chord it
    travels between frames, in simulation order: a differential drive's chord direction lies between the headings at the
    chord's two ends (here within 5 deg of that interval, ten ticks of steering lie between two frames)."""
    lab = gui["label_pose"][::-1]                                      # simulation order
    n = 0
    for a, b in zip(lab[:-1], lab[1:]):
        if np.isfinite(a[[0, 1, 3]]).all() and np.isfinite(b[[0, 1, 3]]).all():
            dx, dy = b[0] - a[0], b[1] - a[1]
            if np.hypot(dx, dy) < 0.3:
                continue
            chord = np.degrees(np.arctan2(dy, dx))
            lo, hi = min(a[3], b[3]), max(a[3], b[3])
            assert lo - 5.0 <= chord <= hi + 5.0, (a, b, chord)
            assert dx > 0                                              # forward = +x of the body frame, v >= 0
            n += 1
    assert n >= 3
    # ... and the heading never turns faster than the action bound allows: |w| <= 1 rad/s (action_bound, ppo_stage2.py:179, circle_test.py:96) x 10 ticks x 0.1 s
    # = 57.3 deg between two frames (a weak bound like the displacement's: 10 ticks per frame is read off the same picture)
    turns = [abs(((b[3] - a[3] + 180.0) % 360.0) - 180.0) for a, b in zip(lab[:-1], lab[1:])
             if np.isfinite(a[3]) and np.isfinite(b[3])]
    assert len(turns) >= 8 and max(turns) <= 57.3 + 1.0, turns
