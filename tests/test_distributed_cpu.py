"""CPU, world_size 2 over gloo: the spp-range split + reduce used by bench.py gives the single-process frame.
(The renderer stand-in on CPU ranks is the oracle, which shares the HIP kernels' RNG stream protocol.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd.distributed import reduce_accum, split_range
from tests import fixtures as fx
from tests import oracle_lib as O


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = fx.scene(2)
    cam, _, _, _ = fx.img(2)
    n = int(cam["sqrt_spp"][0])
    s0, s1 = split_range(n * n, world, rank)
    acc, _ = O.render(sc, cam, seed=11, s_begin=s0, s_end=s1, rng_mode=1)
    t = torch.from_numpy(acc.astype(np.float32).reshape(-1))
    reduce_accum(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_spp_split_matches_single(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    sc = fx.scene(2)
    cam, _, _, _ = fx.img(2)
    whole, _ = O.render(sc, cam, seed=11, rng_mode=1)
    np.testing.assert_allclose(got, whole.reshape(-1), rtol=1e-5, atol=1e-6)


def test_split_range_partitions_for_every_world_size():
    """1, 2, 4, 8 ... ranks (and more ranks than strata): the ranges are contiguous, disjoint, exhaustive, sizes within one."""
    for n in (961, 484, 9, 1, 0):
        for world in (1, 2, 3, 4, 8, 16, 1024):
            parts = [split_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1


def test_bench_reference_leg_times_the_compiled_reference():
    """bench.py's cpu_baseline carries, beside the port, the compiled reference itself (oracle/_ref/refgen) when the binary is
    there: a tiny cornell_box sample here; absent binary -> None, never an error."""
    import os

    import bench
    exe = os.path.join(bench.ROOT, "oracle", "_ref", "refgen")
    leg = bench.reference_leg(2, 32, 18, 16)
    if not os.access(exe, os.X_OK):
        assert leg is None
        return
    assert leg is not None and leg["kind"] == "reference" and leg["threads"] == 1 and leg["msamples_per_s"] > 0
    assert bench.reference_leg(4) is None  # (mesh_scene needs the reference's own shapes directory)


def test_bench_reads_the_latest_committed_hbm_counters():
    """roofline.traffic comes from the newest round's committed PMC summary (2 x FETCH_SIZE + WRITE_SIZE of the render kernel's
    line, KB per launch); a comment that names the kernel must not derail the parser."""
    import glob
    import os

    import bench
    got = bench.profiled_hbm_traffic()
    newest = sorted(os.path.basename(p)[:3] for p in glob.glob(os.path.join(bench.ROOT, "profiles", "r*_hbm_counters.txt")))[-1]
    assert got is not None and os.path.basename(got[1]).startswith(newest), got
    assert 50e6 < got[0] < 5e9
