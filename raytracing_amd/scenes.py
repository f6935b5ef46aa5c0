"""Access to the reference's scenes as flat scene descriptions.

`load(name)` returns the scene serialised from the reference's own scene function (tests/golden/scene*.npz, produced by
oracle/refgen.cpp); the C++ host API (raytracing_amd/host) builds the same descriptions from scene code written against
the reference's scene API."""
from __future__ import annotations

import os

from . import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE_IDS = {"first_scene": 1, "cornell_box": 2, "final_scene": 3, "mesh_scene": 4}


def load(name: str) -> abi.Scene:
    n = SCENE_IDS[name]
    return abi.Scene.load(os.path.join(_ROOT, "tests", "golden", f"scene{n}.npz"))
