"""Drop-in for the reference's ``stage_world2.py``: same class name and constructor, backed by the
batched MI355X world (mrca/stage_world.py)."""
from mrca import spmd
from mrca.stage_world import Stage2World


class StageWorld(Stage2World):
    def __init__(self, beam_num, index, num_env):
        super().__init__(beam_num, index, num_env)
        spmd.runtime().world = self.world
