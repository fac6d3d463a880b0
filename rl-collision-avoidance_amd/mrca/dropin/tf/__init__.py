"""Drop-in for ``tf`` (only tf.transformations is used, stage_world1.py:90,244)."""
from . import transformations  # noqa: F401
