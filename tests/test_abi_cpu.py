"""CPU: the C-ABI library loads without a GPU and exports every symbol include/rt_amd.h declares; struct layouts of
the Python mirror match the compiled ones; host-side argument checks that need no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracing_amd import abi, camera, distributed, lib
from tests import fixtures as fx
from tests import stream_check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="rt_amd.h", prefix="rt_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_host_header_symbols_are_exported():
    from raytracing_amd import host
    L = host.load()
    names = declared_functions("rtamd_host.h", "rtamd_host_")
    assert set(names) == set(host.EXPORTS)
    for n in names:
        assert hasattr(L, n)


def test_header_symbols_are_exported():
    L = lib.load()
    names = declared_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rt_amd.h but not exported"
    assert set(lib.EXPORTS) == set(names)


def test_struct_layouts_match_compiled_library():
    abi.check_layout()
    L = lib.load()
    for name, (_, size) in abi.EXPECTED_SIZES.items():
        assert L.rt_abi_sizeof(name.encode()) == size, name
    assert L.rt_abi_sizeof(b"rt_scene_desc") == C.sizeof(abi.SceneDesc)
    assert L.rt_abi_sizeof(b"rt_render_params") == C.sizeof(abi.RenderParams)
    assert L.rt_abi_sizeof(b"rt_stats") == C.sizeof(abi.Stats)
    assert L.rt_abi_sizeof(b"nope") == -1


def test_no_cpu_fallback():
    """Without a GPU rt_init must fail loudly (the product never routes through the oracle or any CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.RtError):
        lib.Context(0)


def test_product_does_not_reference_oracle():
    for base in ("raytracing_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                    text = open(os.path.join(dp, fn), errors="ignore").read()
                    assert "liboracle" not in text and "oracle_lib" not in text, (dp, fn)
                    assert not re.search(r"#include[^\n]*oracle|import[^\n]*oracle|from[^\n]*oracle[^\n]*import", text), (dp, fn)


def test_scene_npz_roundtrip(tmp_path):
    sc = fx.scene(2)
    p = str(tmp_path / "s.npz")
    sc.save_npz(p)
    assert abi.Scene.load(p).equal(sc)
    d = sc.desc()
    assert d.n_nodes == len(sc.nodes) and d.n_quads == 13 and d.n_spheres == 2


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_python_camera_matches_reference_frame(n):
    assert camera.scene_camera(n, 64, 36, 9, 10).tobytes() == fx.scene(n).camera.tobytes()
    cam32, _, _, _ = fx.img(n)
    assert camera.scene_camera(n, 32, 18, 9, 12).tobytes() == cam32.tobytes()


def test_split_range_properties():
    for n in (1, 9, 961, 484):
        for world in (1, 2, 3, 4, 8):
            parts = [distributed.split_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert sorted(sum((distributed.frame_shards(124, 8, r) for r in range(8)), [])) == list(range(124))


def test_flattener_stream_sizes_and_box_fusion():
    """Host-only run of the flattener (rt_scene_flatten_info): the f32 stream fuses every box() list of six axis-aligned
    quads (quad.h:103) into one AABOX record, the f64 parity stream keeps the reference's quads; malformed graphs are
    refused with a status instead of a crash."""
    info = {n: lib.flatten_info(fx.scene(n)) for n in (1, 2, 3, 4)}
    assert info[3]["fused_boxes"] == 400  # final_scene's ground: 400 boxes = 2400 quads (cpuRenderer.cpp:298-312)
    assert info[1]["fused_boxes"] == 0 and info[4]["fused_boxes"] == 0
    assert info[2]["fused_boxes"] == 1  # cornell's rotated box is axis-aligned in its own (object) space
    for n in (1, 2, 3, 4):
        assert info[n]["f32_units"] <= info[n]["f64_units"]
    assert [info[n]["baked_instances"] for n in (1, 2, 3, 4)] == [0, 0, 1, 1]  # cornell's instance is a box of quads (kept: one AABOX in its own frame); mesh: the bunny
    # with the reference's own nodes and order (reference_order) the two streams differ by exactly the fused and baked
    # records: six 5-unit quads -> one 2-unit record; the translate(rotate_y(1000 spheres)) instance (cpuRenderer.cpp:369-375)
    # is emitted in world space in the f32 stream: no XF_ENTER / XF_EXIT pair for either transform
    ref = {n: lib.flatten_info(fx.scene(n), reference_order=True) for n in (1, 3)}
    # (+ 3: the ceiling light, an axis-aligned quad, is a flat 2-unit AABOX record in the f32 stream)
    assert ref[3]["f64_units"] - ref[3]["f32_units"] == 400 * (6 * 5 - 2) + 4 * 2 + 3
    assert ref[1]["f32_units"] == ref[1]["f64_units"]
    assert [info[n]["lights"] for n in (1, 2, 3, 4)] == [1, 2, 1, 2]
    # a box with one face moved is not fused
    sc = fx.scene(2)
    nodes = sc.nodes
    lists = np.nonzero((nodes["kind"] == abi.RT_NODE_LIST) & (nodes["b"] == 6))[0]
    assert len(lists) == 1
    q = nodes["a"][sc.arrays["child_index"][nodes["a"][lists[0]]]]
    sc.arrays["quads"]["Q"][q][0] += 1.0
    sc.quads["Q"][q, 0] += 1.0
    assert lib.flatten_info(sc)["fused_boxes"] == 0
    bad = fx.scene(2)
    bad.world_root = 10 ** 6
    with pytest.raises(lib.RtError):
        lib.flatten_info(bad)


def test_flattener_refuses_cyclic_graphs():
    """A node that reaches itself (a bvh_node whose child is the node, a list that lists an ancestor) is refused with a
    status; shared subtrees (a DAG, as the reference's final_scene has) are fine."""
    sc = fx.scene(2)
    bvh = np.nonzero(sc.nodes["kind"] == abi.RT_NODE_BVH)[0]
    assert len(bvh) > 2
    bad = fx.scene(2)
    bad.nodes["a"][bvh[1]] = bvh[1]  # self reference
    with pytest.raises(lib.RtError):
        lib.flatten_info(bad)
    bad = fx.scene(2)
    bad.nodes["b"][bvh[-1]] = sc.world_root  # back edge to the root
    with pytest.raises(lib.RtError):
        lib.flatten_info(bad)
    dag = fx.scene(2)
    dag.nodes["b"][bvh[-1]] = dag.nodes["a"][bvh[-1]]  # the same child twice: legal (bvh.h:30 does it for span 1)
    lib.flatten_info(dag)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6, "random 1", "random 2", "random 3"])
def test_own_hierarchy_holds_the_reference_leaves(n):
    """The f32 stream's BOX records are a hierarchy of the flattener's own (surface-area heuristic over all primitives of a
    frame), not the reference's bvh_nodes.  Host-only structure check: the same leaf records as the stream in the
    reference's order (every primitive once; a medium below a span-1 bvh_node twice), every skip link forward, on a record
    boundary and nested within the enclosing BOX's, and every BOX inside the BOX that encloses it."""
    if isinstance(n, str):  # tests/random_scene.py: duplicates, concentric shells, a huge ground, media, instances
        from tests import random_scene
        sc = random_scene.build(int(n.split()[1]))
    else:
        sc = fx.scene(n)
    units, ids = lib.flatten_stream(sc)
    r_units, r_ids = lib.flatten_stream(sc, reference_order=True)
    n_box = stream_check.check(units, ids, r_units, r_ids)
    if not isinstance(n, str):  # (the synthetic scenes are flat lists: no bvh_node at all in the reference's order)
        r_box = sum(k == 0 for _, k in stream_check.records(r_units))
        assert n_box <= r_box + 2, "the reference puts a node above every one or two leaves; ours has no reason to need more"
    else:
        assert n_box > 50
