from .raymarching import *  # noqa: F401,F403  (`from raymarching import raymarching` also works, editing/editgrid.py:3)
from . import raymarching  # noqa: F401
