#!/usr/bin/env python3
"""Headline benchmark: final_scene 1920x1080 @ 1000 spp (-> 31^2 = 961 stratified samples), depth 50, on N GPUs.

One process per GPU (torch.distributed / RCCL).  A "step" is one whole frame: every rank path-traces a disjoint range of
the 961 strata for all pixels (counter-based RNG keyed by (seed, pixel, stratum), so the image does not depend on N),
the float accumulation buffers are summed onto rank 0 with one RCCL reduce over xGMI, and rank 0 applies write_color.
Total work is fixed as N grows -> "scaling": "strong".  value = samples of the frame / wall time per frame.

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (torch.distributed.run,
fresh child processes, before this process touches a GPU) and relays rank 0's JSON line; under torchrun it is a rank.
`--config` picks one of BASELINE.json's configurations (default: the headline one).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs.  flops = work of the REFERENCE algorithm per camera sample (measured on the CPU
# oracle): 27*A + 30*S + 55*Q + 52*T + 120*R with A/S/Q/T = aabb / sphere / quad / triangle tests, R = ray_color calls.
CONFIGS = {
    "final": dict(scene="final_scene", scene_id=3, width=1920, height=1080, spp=1000, depth=50,
                  flops=27 * 102.5 + 30 * 22.1 + 55 * 29.2 + 52 * 0 + 120 * 4.14),
    "cornell": dict(scene="cornell_box", scene_id=2, width=600, height=600, spp=1000, depth=50,
                    flops=27 * 35.1 + 30 * 7.3 + 55 * 43.8 + 52 * 0 + 120 * 5.09),
    "mesh": dict(scene="mesh_scene", scene_id=4, width=1920, height=1080, spp=500, depth=50,
                 flops=27 * 100.0 + 30 * 1.2 + 55 * 12.7 + 52 * 11.0 + 120 * 2.59),
    "first": dict(scene="first_scene", scene_id=1, width=400, height=225, spp=10, depth=50,
                  flops=27 * 393.4 + 30 * 65.9 + 55 * 0 + 52 * 0 + 120 * 20.05),
    # C5: the rotate + zoom sequence of camera::render_animation (62 frames), frames sharded over the ranks, no collective
    "anim": dict(scene="first_scene", scene_id=1, width=1920, height=1080, spp=100, depth=50,
                 flops=27 * 393.4 + 30 * 65.9 + 55 * 0 + 52 * 0 + 120 * 20.05),
}
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (vector), packed FMA


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="final")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end run of the C++ drop-in (rtamd_render)")
    return ap.parse_args(argv)


def launch_ranks(a, argv):
    """--gpus N > 1 outside torchrun: start N fresh rank processes (this process has made no GPU call and makes none),
    wait, relay rank 0's JSON line.  Never re-exec: the children are ordinary subprocesses."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {a.gpus}-rank launch failed (exit {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def cpu_baseline(scene_id, cores_all):
    """The oracle (CPU restatement of the reference algorithm, same Philox streams; OpenMP over rows) on bounded samples of
    the workload: one thread, then the cores this job may use; the better Msamples/s is `value` (the
    reference's own OpenMP path ANTI-scales, so both are stated)."""
    from tests import oracle_lib as O  # test infrastructure: allowed in this leg only, never in the timed GPU path
    from raytracing_amd import scenes
    name = {1: "first_scene", 2: "cornell_box", 3: "final_scene", 4: "mesh_scene"}[scene_id]
    legs = []
    for threads, (W, H, spp) in ((1, (160, 90, 144)), (cores_all, (960, 540, 144))):
        sc = scenes.build(name, W, H, spp, 50)
        cam = sc.camera
        t0 = time.time()
        O.render(sc, cam, seed=1, rng_mode=1, threads=threads)
        dt = time.time() - t0
        legs.append({"threads": threads, "msamples_per_s": round(W * H * spp / dt / 1e6, 4),
                     "sample": f"{name} {W}x{H}, {spp} spp, depth 50 ({W * H * spp} samples, {dt:.1f} s wall)"})
    best = max(legs, key=lambda l: l["msamples_per_s"])
    return {"value": best["msamples_per_s"], "unit": "Msamples/s", "cores": best["threads"], "kind": "port",
            "sample": best["sample"], "legs": legs, "host_cpus_visible": os.cpu_count(),
            "note": "oracle/liboracle.so (the pinned CPU restatement) on this box's host cores; best of {1, cores} threads"}


def profiled_hbm_traffic():
    """HBM bytes per render_kernel launch of this same command (N=1), from the committed rocprofv3 PMC passes
    (tools/prof_round.sh: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; both count KB, and FETCH_SIZE is doubled
    on gfx950 as MI355X_MICROARCH.md prescribes).  None when no summary is committed."""
    import ast
    import glob
    paths = glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_counters.txt"))
    # the latest round's "final" set first, then the other sets, latest round first
    for path in sorted(paths, key=lambda p: (os.path.basename(p)[:3], "_final_" in os.path.basename(p), os.path.basename(p)), reverse=True):
        try:
            for line in open(path):
                if "render_kernel" in line:
                    c = ast.literal_eval(line[line.index("{"):].strip())
                    return int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError, SyntaxError):
            continue
    return None, None


def e2e_dropin(cfg, W, H, spp, depth, n_gpus, devices=None):
    """The C++ drop-in timed end to end, as the reference times itself (cpuRenderer.cpp:446-467: scene build + render +
    PPM write): `rtamd_render <scene> <spp> <depth> false 0 true true 8 <W> <H> [--gpus N]` as a child process."""
    exe = os.path.join(ROOT, "raytracing_amd", "rtamd_render")
    if not os.path.exists(exe):
        return {"error": "raytracing_amd/rtamd_render is not built"}
    cmd = [exe, str(cfg["scene_id"]), str(spp), str(depth), "false", "0", "true", "true", "8", str(W), str(H)]
    env = dict(os.environ)
    env["RTAMD_ASSETS"] = os.path.join(ROOT, "raytracing_amd", "assets")
    env["RTW_IMAGES"] = env["RTAMD_ASSETS"]
    for k in ("RTAMD_DEVICES", "RTAMD_DEVICE"):
        env.pop(k, None)
    if devices:  # rehearsal: an explicit device list (indices may repeat)
        env["RTAMD_DEVICES"] = ",".join(str(d) for d in devices)
    elif n_gpus > 1:
        cmd += ["--gpus", str(n_gpus)]
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "images"))
        t0 = time.perf_counter()
        try:
            proc = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            return {"error": "rtamd_render timed out", "cmd": " ".join(cmd[1:])}
        wall = (time.perf_counter() - t0) * 1e3
        ppm = os.path.join(tmp, "images", "test.ppm")
        ppm_bytes = os.path.getsize(ppm) if os.path.exists(ppm) else 0
    if proc.returncode != 0:
        return {"error": f"rtamd_render exit {proc.returncode}: {proc.stderr.strip()[-300:]}", "cmd": " ".join(cmd[1:])}
    import re
    out = {"cmd": "rtamd_render " + " ".join(cmd[1:]), "process_wall_ms": round(wall, 1), "ppm_bytes": ppm_bytes}
    for key, pat in (("gpus", r"gpus: (\d+)"), ("gpu_init_ms", r"gpu init ms: ([\d.e+-]+)"),
                     ("scene_build_ms", r"scene build ms: ([\d.e+-]+)"), ("flatten_ms", r"flatten ms: ([\d.e+-]+)"),
                     ("upload_ms", r"upload ms: ([\d.e+-]+)"), ("render_readback_ms", r"render\+readback ms: ([\d.e+-]+)"),
                     ("ppm_write_ms", r"ppm write ms: ([\d.e+-]+)"), ("kernel_ms", r"kernel ms: ([\d.e+-]+)"),
                     ("total_render_time_ms", r"Total render time \(ms\): ([\d.e+-]+)")):
        m = re.search(pat, proc.stderr)
        if m:
            out[key] = float(m.group(1)) if key != "gpus" else int(m.group(1))
    if "total_render_time_ms" in out and "kernel_ms" in out:
        # the reference's own contract ("Total render time") minus the one-off creation of the GPU context(s)
        out["e2e_ms"] = round(out["total_render_time_ms"] - out.get("gpu_init_ms", 0.0), 1)
        out["e2e_minus_kernel_ms"] = round(out["e2e_ms"] - out["kernel_ms"], 1)
    return out


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a, argv))

    import numpy as np
    import torch
    import torch.distributed as dist

    from raytracing_amd import abi, animation, lib, scenes
    from raytracing_amd.distributed import reduce_accum, split_range

    cfg = CONFIGS[a.config]
    W, H = a.width or cfg["width"], a.height or cfg["height"]
    spp, depth = a.spp or cfg["spp"], a.depth or cfg["depth"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Rehearsal on a box with fewer GPUs than ranks: RT_AMD_DEVICE_MAP="0,0" puts local ranks 0 and 1 on device 0.  RCCL
    # refuses two ranks on one device, so the exchange then goes through host memory with gloo -- a test mode, labelled in
    # the output; every real run (one rank per device) reduces with RCCL on the device buffers.
    dev_map = [int(x) for x in os.environ.get("RT_AMD_DEVICE_MAP", "").split(",") if x.strip() != ""]
    device = dev_map[local_rank] if dev_map else local_rank
    shared_devices = bool(dev_map) and len(set(dev_map[:world])) < min(world, len(dev_map))
    backend = "gloo" if shared_devices else "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
        world = dist.get_world_size()
    if a.gpus != world and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but the process group has {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)

    t_b0 = time.perf_counter()
    scene = scenes.build(cfg["scene"], W, H, spp, depth, animation=0 if a.config == "anim" else None)  # C++ scene API: BVH built on the host
    host_build_ms = (time.perf_counter() - t_b0) * 1e3
    cam = scene.camera
    n = int(cam["sqrt_spp"][0])
    ctx = lib.Context(device)
    t_u0 = time.perf_counter()
    ds = ctx.upload(scene)
    flatten_upload_ms = (time.perf_counter() - t_u0) * 1e3
    # the same upload with the hierarchy built on the device (rt_scene_upload_device_build): a reported figure, the timed
    # frames below use the host-built stream.  None: the scene has leaves of more than one record (the device build says so)
    device_build_ms = None
    if rank == 0:
        try:
            for _ in range(2):  # (the first call also loads the build kernels' code object)
                t_d0 = time.perf_counter()
                ds_dev = ctx.upload(scene, device_build=True)
                dt_d = (time.perf_counter() - t_d0) * 1e3
                device_build_ms = dt_d if device_build_ms is None else min(device_build_ms, dt_d)
                ds_dev.close()
        except lib.RtError:
            pass
    accum = torch.empty(H * W * 3, dtype=torch.float32, device=dev)
    rgb = torch.empty(H * W * 3, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def exchange():
        if world == 1:
            return
        if backend == "nccl":
            reduce_accum(accum, dst=0)
        else:  # rehearsal: ranks share a device
            host = accum.cpu()
            reduce_accum(host, dst=0)
            if rank == 0:
                accum.copy_(host)

    if a.config == "anim":
        # frames are independent: frame f -> rank f mod world, nothing is exchanged
        cams = [scene.animation[f:f + 1] for f in animation.frame_shards(len(scene.animation), world, rank)]
        n_frames_total = len(scene.animation)
        s0, s1 = 0, n * n
        units_per_step = n_frames_total

        def frame(k=None):
            if k is not None:
                ev[k][0].record()
            for c in cams:
                ds.render_device(c, accum.data_ptr(), seed=1984, precision=abi.RT_PREC_F32, stream=stream)
                ds.tonemap_device(accum.data_ptr(), 1.0 / (n * n), H * W, rgb.data_ptr(), stream=stream)
            if k is not None:
                ev[k][1].record()
        launches_per_step = max(len(cams), 1)
    else:
        cams = [cam]
        s0, s1 = split_range(n * n, world, rank)
        units_per_step = 1

        def frame(k=None):
            if k is not None:
                ev[k][0].record()
            ds.render_device(cam, accum.data_ptr(), seed=1984, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, stream=stream)
            if k is not None:
                ev[k][1].record()
            exchange()
            if rank == 0:
                ds.tonemap_device(accum.data_ptr(), 1.0 / (n * n), H * W, rgb.data_ptr(), stream=stream)
        launches_per_step = 1

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
    # average duration of one render_kernel launch on this rank (HIP events on the stream the kernel runs on)
    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) / launches_per_step if a.steps else 0.0
    t = torch.tensor([dt, kern_ms], dtype=torch.float64, device=dev)
    if world > 1 and backend == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elif world > 1:
        th = t.cpu()
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        t = th
    dt, kern_ms = float(t[0]), float(t[1])
    # rays traced per step (SURVEY 8d asks for rays/s too): one more, untimed, pass that reads the counters back
    rays_local = 0
    for c in cams:
        st = ds.render_device(c, accum.data_ptr(), seed=1984, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, stream=stream,
                              want_stats=True)
        rays_local += int(st["rays"])
    rays = torch.tensor([float(rays_local)], dtype=torch.float64)
    if world > 1:
        if backend == "nccl":
            rays = rays.to(dev)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    rays = float(rays[0])

    if rank == 0:
        samples = W * H * n * n * units_per_step
        sec_per_step = dt / max(a.steps, 1)
        # dominant kernel = render_kernel<float>; per launch this rank traces (s1 - s0) strata of every pixel
        launch_samples = W * H * (s1 - s0)
        achieved = launch_samples * cfg["flops"] / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        headline = a.config == "final" and (W, H, spp) == (1920, 1080, 1000)
        traffic, traffic_src = profiled_hbm_traffic() if world == 1 and headline else (None, None)
        if a.config == "anim":
            workload = (f"first_scene rotate+zoom animation, {units_per_step} frames {W}x{H}, {spp} spp (-> {n * n} stratified samples), "
                        f"depth {depth}, frames sharded over {world} GPU(s), no collective")
            metric = "Msamples/s (node), first_scene animation 1080p"
        else:
            workload = (f"{cfg['scene']} {W}x{H}, {spp} spp (-> {n * n} stratified samples), depth {depth}, "
                        f"spp range split over {world} GPU(s) + " + ("RCCL reduce" if backend == "nccl" else "gloo reduce through host memory (rehearsal: ranks share a device)"))
            metric = f"Msamples/s (node), {cfg['scene']} {'1080p' if (W, H) == (1920, 1080) else f'{W}x{H}'} @{spp} spp"
        out = {
            "metric": metric, "value": round(samples / sec_per_step / 1e6, 3),
            "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 3), "sec_per_frame": round(sec_per_step / units_per_step, 4),
            "grays_per_s": round(rays / sec_per_step / 1e9, 3), "rays_per_sample": round(rays / samples, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "samples_per_step": samples, "collective": None if a.config == "anim" or world == 1 else backend},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 5), "traffic": traffic,
                         "traffic_note": (f"HBM bytes per launch from the committed PMC passes ({traffic_src}); not collected live"
                                          if traffic else "HBM counters need their own profiler passes (profiles/)"),
                         "kernel": "render_kernel<float>", "kernel_ms": round(kern_ms, 3),
                         "flops_per_sample": round(cfg["flops"], 1),
                         "note": "non-MFMA FP32 vector roof (SURVEY 8d): the scene is cache-resident, HBM is not the bound"},
            "host": {"scene_build_ms": round(host_build_ms, 2), "flatten_upload_ms": round(flatten_upload_ms, 2),
                     "device_build_upload_ms": None if device_build_ms is None else round(device_build_ms, 2),
                     "note": "scene function + host BVH build (C++ scene API); flattening + H2D upload of the f32 stream; "
                             "the same upload with the hierarchy built on the device"},
        }
        if not a.no_e2e and a.config != "anim":
            out["e2e"] = e2e_dropin(cfg, W, H, spp, depth, world, devices=dev_map[:world] if shared_devices else None)
        if world == 1 and not a.no_cpu_baseline and a.config != "anim":
            out["cpu_baseline"] = cpu_baseline(cfg["scene_id"], min(16, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
