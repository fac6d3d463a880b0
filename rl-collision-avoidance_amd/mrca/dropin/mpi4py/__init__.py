"""Drop-in for ``from mpi4py import MPI`` (ppo_stage1.py:9): rank threads of mrca.spmd."""
from . import MPI  # noqa: F401
