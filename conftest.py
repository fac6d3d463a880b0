"""Repo-root pytest bootstrap: make the hyphenated package directory importable as ``mrca``."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "rl-collision-avoidance_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
