#!/usr/bin/env python3
"""Ingest the reference's Stage world descriptions into device-ready data files.

Run HERE (the authoring container, where /root/reference is mounted); the outputs are
committed under ``rl-collision-avoidance_amd/mrca/data/`` so that nothing on the GPU box ever
needs /root/reference.

What is read (reference file:line):
  * worlds/stage1.world:3,43-49   resolution, floorplan size 20x20, bitmap rink.png
  * worlds/stage2.world:3,42-48   floorplan size 40x40, bitmap testenv.png
  * worlds/stage2.world:169-297   nine polygon ``obstacle`` models, size [0.7 0.7 0.8]
  * worlds/circle.world:3,42-49   floorplan size 60x60, bitmap rink.png
  * worlds/*.world ``agent( pose [...])`` lines (initial poses)
  * model/utils.py:6-63           init-pose / goal tables (imported, not copied)

Rasterisation rules (DESIGN.md "Map ingest"):
  * a bitmap pixel is occupied iff gray < 128 (black walls and the 127-gray wall fringe);
  * the bounding box of the occupied pixels is scaled onto the floorplan ``size``
    (libstage behaviour), image row 0 is +y;
  * a grid cell is occupied iff any occupied source pixel's rectangle overlaps it
    (conservative), the outermost ring of cells is set (``boundary 1``);
  * polygon obstacles are rescaled so their bounding box is 0.7 x 0.7 m centred on the model
    pose, then a cell is occupied iff any of 8x8 sample points in it is inside the polygon
    (even-odd rule) or a polygon vertex/edge sample falls in it.
Output grid: row-major, row 0 = lowest y, bit-packed little-endian into uint32 words
(bit b of word w on row r  <->  cell column 32*w + b).
"""
import json
import os
import re
import sys

import numpy as np
from PIL import Image

REF = os.environ.get("MRCA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "rl-collision-avoidance_amd", "mrca", "data")


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rl-collision-avoidance_amd'))
from mrca.worldfile import pack_bits, parse_world, raster_bitmap, raster_polygons  # noqa: E402


def save_map(name, grid, size_m, cell, extra=None):
    words = pack_bits(grid)
    meta = dict(cell=float(cell), x0=-size_m[0] / 2.0, y0=-size_m[1] / 2.0,
                width=int(grid.shape[1]), height=int(grid.shape[0]), words_per_row=int(words.shape[1]))
    if extra:
        meta.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), bits=words, meta=json.dumps(meta))
    print(name, meta, "occupied cells:", int(grid.sum()))


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")  # importing anything under the reference's model/ may create ./log
    from model import utils as ref_utils  # noqa: E402
    os.chdir(cwd)

    cell = 0.05
    w1 = parse_world(os.path.join(REF, "worlds/stage1.world"))
    g1 = raster_bitmap(os.path.join(REF, "worlds", w1["bitmap"]), w1["size"][:2], cell)
    save_map("stage1_rink", g1, w1["size"][:2], cell, {"stage_resolution": w1["resolution"]})

    w2 = parse_world(os.path.join(REF, "worlds/stage2.world"))
    g2 = raster_bitmap(os.path.join(REF, "worlds", w2["bitmap"]), w2["size"][:2], cell)
    assert len(w2["obstacles"]) == 9
    g2 = raster_polygons(g2, w2["obstacles"], w2["size"][:2], cell)
    save_map("stage2_testenv", g2, w2["size"][:2], cell, {"stage_resolution": w2["resolution"]})

    wc = parse_world(os.path.join(REF, "worlds/circle.world"))
    cell_c = 0.1
    gc = raster_bitmap(os.path.join(REF, "worlds", wc["bitmap"]), wc["size"][:2], cell_c)
    save_map("circle_rink", gc, wc["size"][:2], cell_c, {"stage_resolution": wc["resolution"]})

    # the same maps at the reference's OWN Stage resolutions (worlds/stage1.world:3, stage2.world:3: 0.2 m;
    # circle.world:3: 0.01 m) for the fidelity scenarios (scenario.*(..., stage_resolution=True)): wall ranges and
    # robot-vs-wall clearances are then quantised like Stage's raster
    for name, w, cell_f, polys in (("stage1_rink_r0200", w1, 0.2, None), ("stage2_testenv_r0200", w2, 0.2, w2["obstacles"]),
                                   ("circle_rink_r0010", wc, 0.01, None)):
        g = raster_bitmap(os.path.join(REF, "worlds", w["bitmap"]), w["size"][:2], cell_f)
        if polys:
            g = raster_polygons(g, polys, w["size"][:2], cell_f)
        save_map(name, g, w["size"][:2], cell_f, {"stage_resolution": w["resolution"]})

    def agents_xyth(w):
        # .world poses are [x y z yaw_deg]; wrap yaw to (-pi, pi] like the GT quaternion round trip
        out = []
        for p in w["agents"]:
            th = np.deg2rad(p[3])
            th = float(np.arctan2(np.sin(th), np.cos(th)))
            out.append([p[0], p[1], th])
        return out

    scen = {
        "stage1": {"world_agents": agents_xyth(w1), "num_agents": len(w1["agents"])},
        "stage2": {
            "world_agents": agents_xyth(w2), "num_agents": len(w2["agents"]),
            # model/utils.py:41-63; the goal table has 34 rows, robots 34..43 draw random goals
            "init_pose": [list(map(float, ref_utils.get_init_pose(i))) for i in range(44)],
            "goal_point": [list(map(float, ref_utils.get_goal_point(i))) for i in range(34)],
            "groups": [0, 6, 10, 15, 19, 24, 34, 44],  # model/utils.py:83
            "random_index_range": [34, 44],             # stage_world2.py:165,211
        },
        "circle": {
            "world_agents": agents_xyth(wc), "num_agents": len(wc["agents"]),
            "init_pose": [list(map(float, ref_utils.test_init_pose(i))) for i in range(50)],   # utils.py:6-23
            "goal_point": [list(map(float, ref_utils.test_goal_point(i))) for i in range(50)],  # utils.py:25-38
        },
    }
    with open(os.path.join(OUT, "scenarios.json"), "w") as f:
        json.dump(scen, f, indent=1)
    print("scenarios.json written:", {k: v["num_agents"] for k, v in scen.items()})


if __name__ == "__main__":
    main()
