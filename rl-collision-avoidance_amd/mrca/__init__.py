"""mrca -- MI355X-native multi-robot collision-avoidance environment + PPO rollout.

Host-side mirror of the reference's ``StageWorld`` / ``model.ppo`` surface over the HIP library
``libmrca_env.so`` (C ABI: include/mrca_env.h).  Importing this package never touches the GPU;
constructing ``VecStageWorld`` requires one and there is no CPU fallback.
"""
from . import scenario  # noqa: F401
from .scenario import Scenario, GridData, stage1, stage2, circle, load_map  # noqa: F401

__all__ = ["scenario", "Scenario", "GridData", "stage1", "stage2", "circle", "load_map"]
