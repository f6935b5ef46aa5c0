#!/usr/bin/env python3
"""Headline benchmark: final_scene 1920x1080 @ 1000 spp (-> 31^2 = 961 stratified samples), depth 50, on N GPUs.

One process per GPU (torch.distributed / RCCL).  A "step" is one whole frame: every rank path-traces a disjoint range of
the 961 strata for all pixels (counter-based RNG keyed by (seed, pixel, stratum), so the image does not depend on N),
the float accumulation buffers are summed onto rank 0 with one RCCL reduce over xGMI, and rank 0 applies write_color.
Total work is fixed as N grows -> "scaling": "strong".  value = samples of the frame / wall time per frame.

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (torch.distributed.run,
fresh child processes, before this process touches a GPU) and relays rank 0's JSON line; under torchrun it is a rank.
`--config` picks one of BASELINE.json's configurations (default: the headline one).  `--config anim` is C5 as BASELINE
states it: all three sequences of first_scene at 1080p -- 62 rotate + zoom frames, 42 translate frames, 20 frames of the
bouncing spheres (scene rebuilt and re-uploaded per frame) -- frames sharded f mod N, no collective.
`--force-collective` makes a ONE-rank run initialise the process group and go through the RCCL reduce / all_reduce / barrier
calls of the N-rank path (tests/test_gpu_multi.py drives it on the one-GPU box).
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
    # C5: the three animation sequences (camera.h:46-154, cpuRenderer.cpp:19-84, 447-451), frames sharded over the ranks, no collective
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
    ap.add_argument("--no-extras", action="store_true",
                    help="profiling runs: only the warm-up and timed launches and the ray-count pass (no tuning probes, no 1/8-range "
                         "and device-build launches, no end-to-end child, no CPU baseline), so that per-kernel counter sums are per launch")
    ap.add_argument("--force-collective", action="store_true",
                    help="one rank: still create the process group and run the N-rank path's collectives (rehearsal of the RCCL branch)")
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
    out = {"value": best["msamples_per_s"], "unit": "Msamples/s", "cores": best["threads"], "kind": "port",
           "sample": best["sample"], "legs": legs, "host_cpus_visible": os.cpu_count(),
           "note": "oracle/liboracle.so (the pinned CPU restatement) on this box's host cores; best of {1, cores} threads"}
    ref = reference_leg(scene_id)
    if ref:
        out["reference"] = ref
    return out


def reference_leg(scene_id, W=160, H=90, spp=144):
    """The COMPILED REFERENCE itself beside the port, when its binary travelled with the tree (oracle/_ref/refgen: the
    reference's own get_ray / ray_color from its headers, built in the build container by oracle/Makefile): one thread -- its
    OpenMP build anti-scales -- on a bounded sample.  final_scene and cornell_box only (mesh_scene reads the
    reference's own shapes directory, which does not travel).  None when the binary is absent or fails."""
    exe = os.path.join(ROOT, "oracle", "_ref", "refgen")
    if scene_id not in (1, 2, 3) or not os.access(exe, os.X_OK):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, "images"), exist_ok=True)
            env = dict(os.environ, RTW_IMAGES=os.path.join(ROOT, "raytracing_amd", "assets"), OMP_NUM_THREADS="1")
            t0 = time.time()
            r = subprocess.run([exe, "render_nocheck", str(scene_id), str(W), str(H), str(spp), "50", "1", os.path.join(tmp, "frame.bin"), "1"],
                               cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            dt = time.time() - t0
        if r.returncode != 0:
            return None
        return {"kind": "reference", "threads": 1, "msamples_per_s": round(W * H * spp / dt / 1e6, 4),
                "sample": f"scene {scene_id} {W}x{H}, {spp} spp, depth 50 ({W * H * spp} samples, {dt:.1f} s wall, scene build included)",
                "note": "oracle/_ref/refgen render_nocheck: the reference's own headers compiled with g++ -O2, glibc rand"}
    except Exception:
        return None


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
                if "render_kernel" in line and not line.lstrip().startswith("#"):
                    c = ast.literal_eval(line[line.index("{"):].strip())
                    return int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError, SyntaxError):
            continue
    return None, None


def partial_frame_bytes(W, H, n_chunks):
    """Algorithmic HBM bytes of one render_kernel launch: the library cuts the sample range into chunks (rt_abi.cpp plan_render;
    `sample_chunks` of rt_stats says how many) and every chunk writes a partial frame of 64-bit cells, 24 B / pixel; a one-chunk
    launch writes the float frame itself, 12 B / pixel.  (The summing pass then reads the partial frames once and writes the
    floats: another kernel.)"""
    cells = W * H * 3
    return 0 if n_chunks <= 0 else (cells * 4 if n_chunks == 1 else n_chunks * cells * 8)


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

    import torch
    import torch.distributed as dist

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
    grouped = world > 1 or a.force_collective  # the process group exists and every collective of the N-rank path runs
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (--force-collective outside torchrun)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
        world = dist.get_world_size()
        rank = dist.get_rank()
    if a.gpus != world and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but the process group has {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    out = run_rank(a, cfg, W, H, spp, depth, rank, world, dev, device, backend, grouped, dist if grouped else None,
                   dev_map[:world] if shared_devices else None)
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(a, cfg, W, H, spp, depth, rank, world, dev, device, backend, grouped, dist, rehearsal_devices):
    """One rank of the benchmark (the whole of it at N = 1).  `dist` is torch.distributed when the process group exists
    (world > 1, or --force-collective), else None; rank 0 returns the JSON line as a dict."""
    import numpy as np
    import torch

    from raytracing_amd import abi, animation, lib, scenes
    from raytracing_amd.distributed import frame_shards, reduce_accum, split_range

    anim = a.config == "anim"
    t_b0 = time.perf_counter()
    scene = scenes.build(cfg["scene"], W, H, spp, depth)  # C++ scene API: BVH built on the host
    host_build_ms = (time.perf_counter() - t_b0) * 1e3
    cam = scene.camera
    n = int(cam["sqrt_spp"][0])
    ctx = lib.Context(device)
    t_u0 = time.perf_counter()
    ds = ctx.upload(scene)
    flatten_upload_ms = (time.perf_counter() - t_u0) * 1e3
    # upload-time probe (rt_scene_tune): the hierarchy build's candidate settings timed on this camera, the fastest stream kept
    # -- outside the timed region, like the upload; reported below
    t_t0 = time.perf_counter()
    tune = None if anim or a.no_extras else ds.tune(cam)
    tune_ms = (time.perf_counter() - t_t0) * 1e3
    accum = torch.empty(H * W * 3, dtype=torch.float32, device=dev)
    rgb = torch.empty(H * W * 3, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def max_over_ranks(vals):
        """MAX over the ranks of a list of floats (RCCL on device tensors; the gloo rehearsal on host tensors)."""
        t = torch.tensor(vals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if grouped:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu()]

    def sum_over_ranks(val):
        t = torch.tensor([float(val)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if grouped:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.cpu()[0])

    def exchange():
        """The one exchange step of a frame: the ranks' float sums onto rank 0.  RCCL runs it on torch's
        current stream -- the stream the render kernel was enqueued on -- so it is ordered behind the kernel without a
        host synchronisation."""
        if not grouped:
            return
        if backend == "nccl":
            reduce_accum(accum, dst=0)
        else:  # rehearsal: ranks share a device
            host = accum.cpu()
            reduce_accum(host, dst=0)
            if rank == 0:
                accum.copy_(host)

    extra = {}
    if anim:
        # C5: frame f of each sequence -> rank f mod world, nothing is exchanged.  The camera sequences (rotate
        # + zoom, translate) move through ONE uploaded scene; every bounce frame builds, uploads and frees its own scene
        # inside the timed region (cpuRenderer.cpp:447-451 rebuilds it per frame too).
        seqs = {}
        for method, label in ((0, "rotate_zoom"), (1, "translate")):
            sc_m = scenes.build(cfg["scene"], W, H, spp, depth, animation=method)
            seqs[label] = (ctx.upload(sc_m), [sc_m.animation[f:f + 1] for f in frame_shards(len(sc_m.animation), world, rank)],
                           len(sc_m.animation))
        bounce_frames = frame_shards(animation.N_BOUNCE_FRAMES, world, rank)
        n_frames_total = sum(v[2] for v in seqs.values()) + animation.N_BOUNCE_FRAMES
        s0, s1 = 0, n * n
        units_per_step = n_frames_total
        seq_ms = {"rotate_zoom": [], "translate": [], "bounce": []}
        bounce_host_ms = []
        seq_ev = {k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for k in seq_ms}

        def frame(k=None):
            if k is not None:
                ev[k][0].record()
            for label, (ds_m, cams_m, _) in seqs.items():
                seq_ev[label][0].record()
                for c in cams_m:
                    ds_m.render_device(c, accum.data_ptr(), seed=1984, precision=abi.RT_PREC_F32, stream=stream)
                    ds_m.tonemap_device(accum.data_ptr(), 1.0 / (n * n), H * W, rgb.data_ptr(), stream=stream)
                seq_ev[label][1].record()
            torch.cuda.current_stream().synchronize()  # (so that the host's build times below are not waits for the frames above)
            seq_ev["bounce"][0].record()
            host_ms = 0.0
            for f in bounce_frames:
                t_h = time.perf_counter()
                sc_f = scenes.build("first_scene_moving", W, H, spp, depth, frame=f)
                try:
                    ds_f = ctx.upload(sc_f, device_build=True)  # hierarchy built on the GPU
                except lib.RtError:
                    ds_f = ctx.upload(sc_f)
                host_ms += (time.perf_counter() - t_h) * 1e3
                ds_f.render_device(sc_f.camera, accum.data_ptr(), seed=1984, precision=abi.RT_PREC_F32, stream=stream)
                ds_f.tonemap_device(accum.data_ptr(), 1.0 / (n * n), H * W, rgb.data_ptr(), stream=stream)
                torch.cuda.current_stream().synchronize()  # (the scene is freed below: its frame must have finished)
                ds_f.close()
            seq_ev["bounce"][1].record()
            if k is not None:
                ev[k][1].record()
                torch.cuda.current_stream().synchronize()
                for label in seq_ms:
                    seq_ms[label].append(seq_ev[label][0].elapsed_time(seq_ev[label][1]))
                bounce_host_ms.append(host_ms)
        my_frames = sum(len(v[1]) for v in seqs.values()) + len(bounce_frames)
        launches_per_step = max(my_frames, 1)
    else:
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
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        frame(k)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    dt = time.perf_counter() - t0
    # average duration of one render_kernel launch on this rank (HIP events on the stream the kernel runs on)
    if anim:  # (the events of a step also span tone mapping and, for the bounce frames, the host's scene builds)
        kern_ms = float(np.mean([sum(seq_ms[k][i] for k in seq_ms) for i in range(a.steps)])) / launches_per_step if a.steps else 0.0
    else:
        kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) / launches_per_step if a.steps else 0.0
    dt, kern_ms = max_over_ranks([dt, kern_ms])
    # rays traced per step (SURVEY 8d asks for rays/s too): one more, untimed, pass that reads the counters back
    rays_local = 0
    if anim:
        for label, (ds_m, cams_m, _) in seqs.items():
            for c in cams_m:
                rays_local += int(ds_m.render_device(c, accum.data_ptr(), seed=1984, precision=abi.RT_PREC_F32, stream=stream, want_stats=True)["rays"])
        for f in bounce_frames:
            sc_f = scenes.build("first_scene_moving", W, H, spp, depth, frame=f)
            ds_f = ctx.upload(sc_f)
            rays_local += int(ds_f.render_device(sc_f.camera, accum.data_ptr(), seed=1984, precision=abi.RT_PREC_F32, stream=stream, want_stats=True)["rays"])
            ds_f.close()
    else:
        st_count = ds.render_device(cam, accum.data_ptr(), seed=1984, s_begin=s0, s_end=s1, precision=abi.RT_PREC_F32, stream=stream, want_stats=True)
        rays_local = int(st_count["rays"])
        launch_chunks = int(st_count["sample_chunks"])
    rays = sum_over_ranks(rays_local)

    if rank != 0:
        return None

    def kernel_ms_of(scene_handle, camera, **kw):
        best = None
        for _ in range(2):
            st = scene_handle.render_device(camera, accum.data_ptr(), seed=1984, precision=abi.RT_PREC_F32, stream=stream, want_stats=True, **kw)
            best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
        return best

    # the same upload with the hierarchy built on the device (rt_scene_upload_device_build), and a frame rendered from that
    # stream: reported figures, the timed frames above use the host-built stream.  None: the device build refuses the scene
    device_build_ms = device_build_render_ms = None
    try:
        for _ in range(0 if a.no_extras else 2):  # (the first call also loads the build kernels' code object)
            t_d0 = time.perf_counter()
            ds_dev = ctx.upload(scene, device_build=True)
            dt_d = (time.perf_counter() - t_d0) * 1e3
            device_build_ms = dt_d if device_build_ms is None else min(device_build_ms, dt_d)
            if _ == 1 and not anim:
                device_build_render_ms = kernel_ms_of(ds_dev, cam, s_begin=s0, s_end=s1)
            ds_dev.close()
    except lib.RtError:
        pass
    headline = a.config == "final" and (W, H, spp) == (1920, 1080, 1000)
    if world == 1 and not anim and n * n >= 64 and not a.no_extras:
        # what each of 8 GPUs would render: one eighth of the strata of every pixel (north_star asks for the 8-GPU frame;
        # this is its kernel on one GPU -- the gap to 1/8 of the whole frame's kernel is the end-of-kernel tail)
        e_ms = kernel_ms_of(ds, cam, s_begin=0, s_end=(n * n) // 8)
        extra["eighth_range_ms"] = round(e_ms, 3)
        extra["eighth_range_vs_linear"] = round(kern_ms * ((n * n) // 8) / (n * n) / e_ms, 4) if e_ms > 0 else None

    samples = W * H * n * n * units_per_step
    sec_per_step = dt / max(a.steps, 1)
    # dominant kernel = render_kernel<float>; per launch this rank traces (s1 - s0) strata of every pixel
    launch_samples = W * H * (s1 - s0)
    achieved = launch_samples * cfg["flops"] / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    traffic, traffic_src = profiled_hbm_traffic() if world == 1 and headline else (None, None)
    if anim:
        workload = (f"first_scene animation, all three sequences: 62 rotate+zoom + 42 translate + 20 bounce = {units_per_step} frames {W}x{H}, "
                    f"{spp} spp (-> {n * n} stratified samples), depth {depth}, frames sharded f mod {world} over {world} GPU(s), no collective; "
                    "the bounce frames rebuild and re-upload their scene inside the timed region")
        metric = "Msamples/s (node), first_scene animation 1080p"
        extra["sequences"] = {
            label: {"frames": {"rotate_zoom": 62, "translate": 42, "bounce": animation.N_BOUNCE_FRAMES}[label],
                    "frames_on_rank0": len(bounce_frames) if label == "bounce" else len(seqs[label][1]),
                    "sec_per_frame_rank0": round(float(np.mean(ms)) * 1e-3 / max(len(bounce_frames) if label == "bounce" else len(seqs[label][1]), 1), 5)}
            for label, ms in seq_ms.items() if ms}
        if bounce_host_ms and bounce_frames:
            extra["sequences"]["bounce"]["host_build_upload_ms_per_frame"] = round(float(np.mean(bounce_host_ms)) / len(bounce_frames), 3)
            extra["sequences"]["bounce"]["host_share"] = round(float(np.mean(bounce_host_ms)) / max(float(np.mean(seq_ms["bounce"])), 1e-9), 4)
    else:
        how = "one GPU, whole sample range, no collective" if not grouped else (
            f"spp range split over {world} GPU(s) + " + ("one RCCL reduce onto rank 0" if backend == "nccl" else "gloo reduce through host memory (rehearsal: ranks share a device)"))
        workload = f"{cfg['scene']} {W}x{H}, {spp} spp (-> {n * n} stratified samples), depth {depth}, {how}"
        metric = f"Msamples/s (node), {cfg['scene']} {'1080p' if (W, H) == (1920, 1080) else f'{W}x{H}'} @{spp} spp"
    out = {
        "metric": metric, "value": round(samples / sec_per_step / 1e6, 3),
        "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(sec_per_step * 1e3, 3), "sec_per_frame": round(sec_per_step / units_per_step, 4),
        "grays_per_s": round(rays / sec_per_step / 1e9, 3), "rays_per_sample": round(rays / samples, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "samples_per_step": samples, "collective": backend if grouped and not anim else None},
        "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 5), "traffic": traffic,
                     "traffic_note": (f"HBM bytes per launch from the committed PMC passes ({traffic_src}); not collected live"
                                      if traffic else "HBM counters need their own profiler passes (profiles/)"),
                     "algorithmic_bytes": None if anim else partial_frame_bytes(W, H, launch_chunks), "sample_chunks": None if anim else launch_chunks,
                     "kernel": "render_kernel<float>", "kernel_ms": round(kern_ms, 3),
                     "flops_per_sample": round(cfg["flops"], 1),
                     "note": "non-MFMA FP32 vector roof (SURVEY 8d): the scene is cache-resident, HBM is not the bound"},
        "host": {"scene_build_ms": round(host_build_ms, 2), "flatten_upload_ms": round(flatten_upload_ms, 2),
                 "tune": None if tune is None else {"ms": round(tune_ms, 1), "chosen": tune["chosen"], "probe_ms": [round(x, 3) for x in tune["probe_ms"]],
                                                    "note": "rt_scene_tune: 6 settings of the hierarchy build probed on this camera; 0 = the default"},
                 "device_build_upload_ms": None if device_build_ms is None else round(device_build_ms, 2),
                 "device_build_render_ms": None if device_build_render_ms is None else round(device_build_render_ms, 3),
                 "note": "scene function + host BVH build (C++ scene API); flattening + H2D upload of the f32 stream; the same "
                         "upload with the hierarchy built on the device; render_kernel time of this rank's launch on that stream"},
    }
    out.update(extra)
    if not a.no_e2e and not anim and not a.no_extras:
        out["e2e"] = e2e_dropin(cfg, W, H, spp, depth, world, devices=rehearsal_devices)
    if world == 1 and not a.no_cpu_baseline and not anim and not a.no_extras:
        out["cpu_baseline"] = cpu_baseline(cfg["scene_id"], min(16, os.cpu_count() or 1))
    return out


if __name__ == "__main__":
    main()
