"""`.world` + bitmap ingest (mrca/worldfile.py): a synthetic Stage world written by the test (runs
anywhere), and -- where the reference checkout is present -- the three reference worlds must
re-rasterise to exactly the committed data files."""
import os

import numpy as np
import pytest

import util as U
from util import S

REF = "/root/reference"

WORLD = """
resolution 0.1
define floorplan model ( boundary 1 ranger_return 1 obstacle_return 1 )
floorplan
(
  name "test"
  size [10.000 8.000 0.800]
  pose [0.000 0.000 0.000 0.000]
  bitmap "map.png"
)
define agent position ( size [0.44 0.38 0.22] drive "diff" )
define obstacle position ( ranger_return 1 )
agent( pose [1.00 2.00 0.00 90.00])   # a comment
agent( pose [-3.00 0.50 0.00 270.00])
obstacle( pose [2 -1 0.00 0]
  size [1.0 1.0 0.8]
  block( points 4
    point[0] [0 0]
    point[1] [0 2]
    point[2] [2 2]
    point[3] [2 0]
    z [0 1]
  )
)
"""


def test_synthetic_world(tmp_path):
    from PIL import Image
    from mrca import worldfile
    img = np.full((80, 100), 255, np.uint8)
    img[0, :] = img[-1, :] = 0
    img[:, 0] = img[:, -1] = 0
    img[40:44, 20:60] = 0                       # an inner wall
    Image.fromarray(img).save(tmp_path / "map.png")
    (tmp_path / "t.world").write_text(WORLD)
    grid, agents, w = worldfile.load_world(str(tmp_path / "t.world"), 0.1)
    assert (grid.width, grid.height, grid.cell, grid.x0, grid.y0) == (100, 80, 0.1, -5.0, -4.0)
    assert w["resolution"] == 0.1 and len(agents) == 2
    assert np.allclose(agents[0], [1.0, 2.0, np.pi / 2]) and np.allclose(agents[1], [-3.0, 0.5, -np.pi / 2])
    d = grid.dense()
    assert d[0].all() and d[-1].all() and d[:, 0].all() and d[:, -1].all()
    # image rows 40..43 from the top -> y in [-0.4, 0.0): grid rows 36..39; columns 20..59
    assert d[36:40, 20:60].all() and not d[30, 40] and not d[45, 40]
    # the 1 x 1 m obstacle centred on (2, -1): cells x 65..74, y 25..34
    assert d[26:34, 66:74].all() and not d[20, 70] and not d[30, 80]
    # and it drives an env: the oracle can ray-cast it
    sc = S.stage1(num_worlds=1, robots_per_world=2, grid=grid)
    o = U.oracle_env(sc)
    o.reset(None, np.array([[1.0, 2.0, np.pi / 2], [-3.0, 0.5, -np.pi / 2]], np.float32), np.zeros((2, 2), np.float32))
    assert abs(o.scan[0, 255] - 1.9) < 0.06       # wall at y = 3.9 seen from y = 2 looking +y


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "worlds", "stage2.world")), reason="reference checkout absent")
@pytest.mark.parametrize("world,data,cell", [("stage1.world", "stage1_rink", 0.05), ("stage2.world", "stage2_testenv", 0.05),
                                             ("circle.world", "circle_rink", 0.1)])
def test_reference_worlds_reproduce_committed_data(world, data, cell):
    from mrca import worldfile
    grid, agents, w = worldfile.load_world(os.path.join(REF, "worlds", world), cell)
    ref = S.load_map(data)
    assert (grid.width, grid.height) == (ref.width, ref.height) and np.array_equal(grid.bits, ref.bits)
    n = {"stage1.world": 24, "stage2.world": 44, "circle.world": 50}[world]
    assert len(agents) == n
    if world == "stage2.world":
        tb = S.load_tables()["stage2"]["init_pose"]
        for a, t in zip(agents, tb):        # the .world poses equal model/utils.py's table
            assert abs(a[0] - t[0]) < 1e-9 and abs(a[1] - t[1]) < 1e-9
            assert abs(np.angle(np.exp(1j * (a[2] - t[2])))) < 1e-6
