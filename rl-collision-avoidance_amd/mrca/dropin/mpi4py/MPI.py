"""COMM_WORLD over the in-process SPMD runtime (gather / scatter / bcast / Get_rank / Get_size)."""
from mrca.spmd import Comm

COMM_WORLD = Comm()
