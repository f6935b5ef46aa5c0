#!/usr/bin/env python3
"""Headline benchmark: final_scene 1920x1080 @ 1000 spp (-> 31^2 = 961 stratified samples), depth 50, on N GPUs.

One process per GPU (torch.distributed / RCCL).  A "step" is one whole frame: every rank path-traces a disjoint range of
the 961 strata for all pixels (counter-based RNG keyed by (seed, pixel, stratum), so the image does not depend on N),
the float accumulation buffers are summed onto rank 0 with one RCCL reduce over xGMI, and rank 0 applies write_color.
Total work is fixed as N grows -> "scaling": "strong".  value = samples of the frame / wall time per frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracing_amd import abi, camera, lib, scenes  # noqa: E402
from raytracing_amd.distributed import reduce_accum, split_range  # noqa: E402

# Work of the reference algorithm per camera sample on final_scene (BASELINE.md / SURVEY.md 8d; re-measured by the
# oracle's counters in tests/test_work_counts.py): flops = 27*A + 30*S + 55*Q + 52*T + 120*R
FLOPS_PER_SAMPLE = {"final_scene": 27 * 102.5 + 30 * 22.1 + 55 * 29.2 + 52 * 0 + 120 * 4.14}
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (vector), packed FMA


def cpu_baseline(scene, cores):
    """The oracle (CPU restatement of the reference algorithm, same Philox streams) on a bounded sample of the workload."""
    from tests import oracle_lib as O
    W, H, spp = 960, 540, 144  # ~75 M samples: 10-20 s on 16 host cores
    cam = camera.scene_camera(3, W, H, spp, 50)
    t0 = time.time()
    _, cnt = O.render(scene, cam, seed=1, rng_mode=1, threads=cores)
    dt = time.time() - t0
    return {"value": round(W * H * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"final_scene {W}x{H}, {spp} spp, depth 50 ({W * H * spp} samples, {dt:.1f} s wall)"}


def profiled_hbm_traffic():
    """HBM bytes per render_kernel launch of this same command (N=1), from the committed rocprofv3 PMC passes
    (tools/prof_round.sh: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; both count KB, and FETCH_SIZE is doubled
    on gfx950 as MI355X_MICROARCH.md prescribes).  None when the summary is missing."""
    import ast
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_dealing_hbm_counters.txt")
    try:
        for line in open(path):
            if "render_kernel" in line:
                c = ast.literal_eval(line[line.index("{"):].strip())
                return int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))
    except (OSError, ValueError, KeyError, SyntaxError):
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    scene = scenes.build("final_scene", a.width, a.height, a.spp, a.depth)  # C++ scene API: BVH built on the host
    cam = scene.camera
    W, H, n = int(cam["image_width"][0]), int(cam["image_height"][0]), int(cam["sqrt_spp"][0])
    ctx = lib.Context(local_rank)
    ds = ctx.upload(scene)
    s0, s1 = split_range(n * n, world, rank)
    accum = torch.empty(H * W * 3, dtype=torch.float32, device=dev)
    rgb = torch.empty(H * W * 3, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def frame(k=None):
        if k is not None:
            ev[k][0].record()
        ds.render_device(cam, accum.data_ptr(), seed=1984, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, stream=stream)
        if k is not None:
            ev[k][1].record()
        reduce_accum(accum, dst=0)
        if rank == 0:
            ds.tonemap_device(accum.data_ptr(), 1.0 / (n * n), H * W, rgb.data_ptr(), stream=stream)

    for _ in range(a.warmup):
        frame()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        frame(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) if a.steps else 0.0
    t = torch.tensor([dt, kern_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, kern_ms = float(t[0]), float(t[1])
    # rays traced per frame (SURVEY 8d asks for rays/s too): one more, untimed, launch that reads the counter back
    st = ds.render_device(cam, accum.data_ptr(), seed=1984, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, stream=stream,
                          want_stats=True)
    rays = torch.tensor([float(st["rays"])], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    rays = float(rays[0])

    if rank == 0:
        samples = W * H * n * n
        sec_per_frame = dt / max(a.steps, 1)
        fps_flops = FLOPS_PER_SAMPLE["final_scene"]
        # dominant kernel = render_kernel<float>; per launch each rank traces (s1-s0)/n^2 of the frame
        rank_samples = W * H * (s1 - s0)
        achieved = rank_samples * fps_flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        traffic, traffic_src = profiled_hbm_traffic() if world == 1 and (a.width, a.height, a.spp) == (1920, 1080, 1000) else (None, None)
        out = {
            "metric": "Msamples/s (node), final_scene 1080p @1000 spp", "value": round(samples / sec_per_frame / 1e6, 3),
            "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(sec_per_frame * 1e3, 3), "sec_per_frame": round(sec_per_frame, 4),
            "grays_per_s": round(rays / sec_per_frame / 1e9, 3), "rays_per_sample": round(rays / samples, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"final_scene {W}x{H}, {a.spp} spp (-> {n * n} stratified samples), depth {a.depth}, "
                                   f"spp range split over {world} GPU(s) + RCCL reduce", "samples_per_frame": samples},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 5), "traffic": traffic,
                         "traffic_note": (f"HBM bytes per launch from the committed PMC passes ({traffic_src}); not collected live"
                                          if traffic else "HBM counters need their own profiler passes (profiles/)"),
                         "kernel": "render_kernel<float>", "kernel_ms": round(kern_ms, 3),
                         "flops_per_sample": round(fps_flops, 1),
                         "note": "non-MFMA FP32 vector roof (SURVEY 8d): the scene is cache-resident, HBM is not the bound"},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
