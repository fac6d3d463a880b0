#!/usr/bin/env python3
"""Upper bounds of two learner items before anything is built for them, through bench.py --mode train:

  no-gather : every minibatch takes the SAME pre-gathered observation stacks (what reading the frames through `fidx` inside
              the front end could save at most: the gather runs on a side stream next to the minibatch before it)
  no-bias   : fc1 / fc2 biases do not require a gradient (what forming their gradients in the epilogues of the kernels that
              produce dh1 / dz could save at most: four `sum` launches per minibatch)

    python tools/train_bounds_probe.py base|no-gather|no-bias [bench.py arguments]
Results of these runs are NOT training (wrong data / frozen biases): timing only.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-collision-avoidance_amd"))
sys.path.insert(0, ROOT)
what = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py"), "--mode", "train"] + sys.argv[2:]
import bench  # noqa: E402
from mrca import ppo  # noqa: E402

if what == "no-gather":
    take0 = ppo._ObsPrefetch.take
    issue0 = ppo._ObsPrefetch._issue
    cache = {}

    def _issue(self, k):
        if "x" not in cache:
            issue0(self, k)

    def take(self, k):
        if "x" not in cache:
            cache["x"] = take0(self, 0 if k == 0 else k)
        return cache["x"]

    ppo._ObsPrefetch._issue = _issue
    ppo._ObsPrefetch.take = take
elif what == "no-bias":
    from mrca import net

    init0 = net.CNNPolicy.__init__

    def init(self, *a, **kw):
        init0(self, *a, **kw)
        for name in ("act_fc1", "act_fc2", "crt_fc1", "crt_fc2"):
            getattr(self, name).bias.requires_grad_(False)

    net.CNNPolicy.__init__ = init
bench.main()
