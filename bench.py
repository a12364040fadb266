#!/usr/bin/env python3
"""Benchmark of the ray-casting hot path: Mrays/s (primary + shadow rays started per second).

    python bench.py --gpus N --steps K --warmup W

One "step" is one make_step (main.cpp:439-477): process every live ray slot once (traversal /
intersection / shading / spawn), sort the slots by group, re-pack the queue.  The workload is the
headline configuration: the reference's scene (256 instances alternating bunny and dragon,
main.cpp:268-343) at 1920x1080 with a pool of 2,080,768 ray slots per GPU; dragon_vrip.ply cannot be
shipped, so a generated stand-in mesh of the same size takes its place.  With N > 1 (one process per
GPU, launched by torch.distributed.run) the frame is split into N row bands, the scene is replicated,
every GPU keeps a full-size pool (weak scaling) and the resolved RGBA8 bands are all-gathered with
RCCL once per frame, inside the timed region.  The timed region is K steps plus one resolve of the
frame into device memory (the reference resolves into a GL texture, main.cpp:479-490); the cost of
also copying the frame to host memory is reported separately (frame_resolve_to_host_ms).

Rank 0 prints ONE JSON line.  `value` is measured over K steps of the production schedule (the leaf
kernel on a second stream beside the box and shading kernels, no events between kernels).  `roofline`
is for the dominant kernel; its duration comes from HIP events on the kernel's stream over K more
steps run right after the timed ones with the kernels serialised (--stage-timers timed puts the
events inside the timed steps instead).  `cpu_baseline` is the CPU restatement of the reference
kernels (oracle/) on the host cores, on a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracer_amd import api, scene as S, tiling  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, spec
VALU_PEAK_TFLOPS = 157.3   # MI355X FP32 vector, spec
# Algorithmic cost model of the path: per ray traced 32 B in + 16 B out; per box test 25 flop, per
# triangle test 55 flop, per shaded sample about 150 flop.
BYTES_PER_RAY = 48
FLOP_PER_BOX, FLOP_PER_TRI, FLOP_PER_SHADE = 25, 55, 150


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=64)
    p.add_argument("--warmup", type=int, default=96)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--scene", default="bunny+dragon", choices=["bunny+dragon", "bunny", "dragon"])
    p.add_argument("--grid", type=int, default=0,
                   help="k > 0: a regular k*k*k grid of instances instead of the reference's random field of 256 (the "
                        "incoherent workload: --scene dragon --grid 5 --shading diffuse keeps 109 M triangles instanced)")
    p.add_argument("--shading", default="reference", choices=["reference", "mirror", "diffuse"],
                   help="what a material does with a sample: the reference's direct light + shadow ray, or --bounces "
                        "mirror / cosine-weighted diffuse bounces first (RT_FLAG_SHADING)")
    p.add_argument("--bounces", type=int, default=2)
    p.add_argument("--settle", type=int, default=96,
                   help="untimed steps run before the --warmup steps, whatever --warmup says: after rt_init_frame every ray "
                        "starts at the scene root in lock-step, and the rate only means something once the pool has mixed")
    p.add_argument("--no-extras", action="store_true", help="skip the extra configurations and the band prediction (N = 1 only)")
    p.add_argument("--warp-width", type=int, default=64)
    p.add_argument("--schedule", default="free", choices=["free", "reference"],
                   help="free: RT_FLAG_FREE_SCHEDULE (any processing order, same per-sample results); reference: every "
                        "step reproduces the reference's make_step state exactly")
    p.add_argument("--arith", default="mad", choices=["strict", "mad"],
                   help="mad: multiply-adds fused at the stated places (RT_FLAG_ARITH_MAD), which is how the reference builds its "
                        "kernels (-cl-mad-enable, main.cpp:253); strict: every product and sum rounded on its own.  Either way the "
                        "HIP path equals the oracle of the same mode bit for bit; the other mode's rate is in the extras")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg (0 disables it)")
    p.add_argument("--dragon", default=os.environ.get("RT_DRAGON_PLY", ""), help="path to dragon_vrip.ply if available")
    p.add_argument("--stage-timers", default="after", choices=["after", "timed"],
                   help="after: the timed steps run the production schedule untouched and the per-kernel HIP events are "
                        "taken over the same number of steps right after them, kernels serialised; timed: events inside "
                        "the timed steps (costs about 9% and the leaf span overlaps the other two)")
    p.add_argument("--aabb-threshold", type=int, default=128,
                   help="boxes per inner group of the hierarchy build (the reference builds with 128, main.cpp:290)")
    p.add_argument("--tri-threshold", type=int, default=128, help="triangles per leaf (the reference builds with 128)")
    p.add_argument("--serial", action="store_true",
                   help="keep every kernel of a step on one stream (as the stage-timing pass does): a rocprofv3 summary "
                        "of this run shows each kernel's duration alone")
    p.add_argument("--collective", action="store_true",
                   help="with --gpus 1: initialise torch.distributed (one rank) and run the frame gather of the N > 1 path -- the "
                        "resolved band all-gathered with RCCL -- inside the timed region, as every rank of a multi-GPU run does")
    p.add_argument("--spp", type=int, default=16,
                   help="samples per pixel of the fixed-work measurement in the extras (a frame and its eight bands rendered to "
                        "exactly this many samples per pixel, drain included)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo gathers through host memory: a rehearsal of the N > 1 path on a box with fewer GPUs than ranks")
    return p.parse_args()


def model_files(args, rank):
    if args.scene == "bunny":
        return [S.BUNNY, S.BUNNY], "bun_zipper x2"
    dragon = args.dragon
    label = "dragon_vrip"
    if not dragon or not os.path.exists(dragon):
        dragon = os.path.join(tempfile.gettempdir(), f"rt_dragon_standin_{os.getuid()}_{rank}.ply")
        if not os.path.exists(dragon):
            S.write_standin_ply(dragon + ".tmp", 934, 466)
            os.replace(dragon + ".tmp", dragon)
        label = "dragon stand-in (generated, 870,488 triangles)"
    if args.scene == "dragon":
        return [dragon, dragon], label + " x2"
    return [S.BUNNY, dragon], "bun_zipper + " + label


def cpu_baseline(args, files, seconds):
    """The oracle (CPU restatement of the reference kernels), multi-threaded over work-groups, on a
    bounded sample: a middle band of the same frame with a proportionally smaller pool."""
    from oracle import pyoracle as O
    threads = min(16, len(os.sched_getaffinity(0)))
    rows = max(8, args.height // 8)
    first = (args.height - rows) // 2
    pool = args.width * rows
    sc = S.build_scene(files, args.width, args.height, pool, band_first=first, band_rows=rows)
    o = O.Oracle(sc, warp_width=32, threads=threads, mad=args.arith == "mad")
    o.init_frame()
    t0 = time.time()
    warm = 0
    while warm < 48 and time.time() - t0 < seconds * 0.5:
        o.step(4)
        warm += 4
    c0 = o.counters()
    t1 = time.time()
    timed = 0
    while timed < 8 or (time.time() - t1 < seconds * 0.5 and timed < 64):
        o.step(4)
        timed += 4
    dt = time.time() - t1
    c1 = o.counters()
    rays = int(c1["primary_spawned"] - c0["primary_spawned"]) + int(c1["shadow_spawned"] - c0["shadow_spawned"])
    o.close()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"rows {first}..{first + rows - 1} of the {args.width}x{args.height} frame, pool {pool} slots, "
                      f"WARP_WIDTH 32, {warm} warm-up + {timed} timed steps, OpenMP over work-groups"}


def shading_flags(args):
    mode = {"reference": api.SHADE_REFERENCE, "mirror": api.SHADE_MIRROR, "diffuse": api.SHADE_DIFFUSE}[args.shading]
    return api.shading_flags(mode, args.bounces) if mode else 0


def settle(rt, steps):
    """Runs `steps` untimed steps, then more (up to as many again) until the number of samples started per
    step has stopped drifting: under 2 % between two windows of 8 steps.  Returns (steps run, steady)."""
    rt.make_step(steps)
    done, last = steps, None
    while done < 2 * steps + 16:
        c0 = rt.counters()
        rt.make_step(8)
        c1 = rt.counters()
        done += 8
        rate = int(c1["primary_spawned"]) - int(c0["primary_spawned"])
        if last is not None and abs(rate - last) <= 0.02 * max(rate, last, 1):
            return done, True
        last = rate
    return done, False


def quick_rate(files, width, height, args, device, band=(0, 0), steps=32, arith=None, aabb_threshold=None):
    """Mrays/s and ms/step of one more configuration on this GPU (settled, `steps` timed steps)."""
    pool = S.L.align_up(width * height, S.L.SORT_TILE)
    sc = S.build_scene(files, width, height, pool, band_first=band[0], band_rows=band[1], grid=args.grid,
                       tri_threshold=args.tri_threshold, aabb_threshold=aabb_threshold or args.aabb_threshold)
    rt = api.RayTracer(sc, device=device, warp_width=args.warp_width, free_schedule=args.schedule == "free",
                       shading=shading_flags(args), mad=(arith or args.arith) == "mad")
    rt.init_frame()
    _, steady = settle(rt, args.settle)
    rt.sync()
    c0 = rt.counters()
    t0 = time.perf_counter()
    rt.make_step(steps)
    rt.draw_frame_device()
    rt.sync()
    dt = time.perf_counter() - t0
    c1 = rt.counters()
    rt.close()
    rays = int(c1["primary_spawned"]) - int(c0["primary_spawned"]) + int(c1["shadow_spawned"]) - int(c0["shadow_spawned"])
    return {"mrays_per_s": rays / dt / 1e6, "ms_per_step": dt * 1e3 / steps, "steady": steady, "pool": pool, "groups": sc.group_count,
            "box_tests_per_ray": (int(c1["box_tests"]) - int(c0["box_tests"])) / max(rays, 1)}


def time_to_spp(files, width, height, args, device, band, spp, band_pool=False):
    """Fixed work: milliseconds to start spp * pixels samples of the band (or frame) and finish every one of them, frame
    set-up included.  The pool is as large as the frame has pixels, as in the steady-state measurement, so a band of
    a multi-GPU run keeps several samples of a pixel in flight at once.  Steps run at the production granularity
    until all samples have been started, then at a granularity that goes down to 1 with the number of rays left until
    none is in flight (the few samples of this scene that circle for ever in the reference's automaton excepted).  band_pool: a pool of as many slots as the
    band has pixels instead (shorter steps, a larger share of parked rays)."""
    pool = S.L.align_up(width * (band[1] if band_pool and band[1] else height), S.L.SORT_TILE)
    sc = S.build_scene(files, width, height, pool, band_first=band[0], band_rows=band[1], grid=args.grid)
    limit = spp * sc.width * sc.height
    rt = api.RayTracer(sc, device=device, warp_width=args.warp_width, sample_limit=limit, free_schedule=args.schedule == "free",
                       shading=shading_flags(args), mad=args.arith == "mad")
    rt.init_frame()
    rt.make_step(2)
    rt.sync()                                    # (first use of the context: graph capture)
    t0 = time.perf_counter()
    rt.init_frame()
    steps = 0
    while steps < 100000:
        rt.make_step(16)
        steps += 16
        g = rt.global_data()[0]
        # until every sample has been started -- or so few whole waves are left that the rest sits parked, which at this
        # granularity would never move again
        if int(g["pixel_offset"]) >= limit or int(g["ray_count"]) < rt.pool // 32:
            break
    # The drain: whole waves of one group become rare as the pool empties, so the granularity of the re-packing goes down
    # with the live count -- at granularity 1 from the start every wave would hold a dozen groups and visit them in turn.
    last, quiet, n, width = -1, 0, -1, args.warp_width
    while steps < 200000:
        rt.make_step(8)
        steps += 8
        n = int(rt.global_data()["ray_count"][0])
        if n == 0 and width == 1:
            break
        want = 64 if n > rt.pool // 4 else 16 if n > rt.pool // 16 else 4 if n > rt.pool // 128 else 1
        if want < width:
            width = want
            rt.set_warp_width(width)
        # the same few rays for 24 steps: those that never finish (about one sample in two million on this scene)
        quiet = quiet + 1 if n == last and width == 1 and n <= max(16, limit // 250000) else 0
        if quiet >= 3:
            break
        last = n
    rt.draw_frame_device()
    dt = time.perf_counter() - t0
    c = rt.counters()
    rt.close()
    return {"ms": dt * 1e3, "steps": steps, "rays": int(c["primary_spawned"]) + int(c["shadow_spawned"]), "left_in_flight": max(n, 0)}


def fixed_work(files, width, height, args, device, gather_ms):
    """A frame of --spp samples per pixel on this GPU, and each of its eight bands alone on this GPU as a rank of an
    8-GPU run would render it: what tiling buys for a frame of fixed size."""
    out = {"gather_ms": gather_ms,
           "note": "one GPU, bands one after the other; every run ends with a drain of 100-200 steps (a ray advances one node "
                   "per step, and with few rays left a wave holds many groups and visits them in turn), which is why few samples per "
                   "pixel scale badly; the gather time is this run's "
                   "resolve + all-gather of one band (a single rank unless --gpus > 1), so the prediction leaves out what eight "
                   "ranks on xGMI add"}
    for spp in sorted({args.spp, 8 * args.spp}):
        whole = time_to_spp(files, width, height, args, device, (0, 0), spp)
        row = {"frame_ms": whole["ms"], "frame_steps": whole["steps"], "frame_mrays_per_s": whole["rays"] / whole["ms"] / 1e3}
        bands = [time_to_spp(files, width, height, args, device, tiling.band_of(r, 8, height), spp) for r in range(8)]
        slowest = max(b["ms"] for b in bands)
        row.update({"band_ms": [round(b["ms"], 3) for b in bands], "band_steps": [b["steps"] for b in bands],
                    "predicted_8gpu_frame_ms": slowest + gather_ms,
                    "predicted_strong_scaling_efficiency": whole["ms"] / (8 * (slowest + gather_ms))})
        if spp == args.spp:  # a pool as small as the band: shorter steps, but two to five times as many of them
            small = [time_to_spp(files, width, height, args, device, tiling.band_of(r, 8, height), spp, True) for r in (0, 4)]
            row["band_sized_pool_ms_bands_0_and_4"] = [round(b["ms"], 3) for b in small]
        out[f"{spp}_spp"] = row
    return out


def library_digest():
    path = os.environ.get("RT_HIP_LIB", api.HIP_LIB_PATH)
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    dist = torch = None
    device = int(os.environ.get("RT_BENCH_DEVICE", local_rank))
    collective = world > 1 or args.collective
    if collective:
        import torch
        import torch.distributed as dist
        if world == 1:  # a single rank, set up as the launcher would for more
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL announces itself on stdout when its communicator is created (at the first collective); stdout is for the one
        # JSON line, so until then it is pointed at stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                torch.cuda.set_device(device)
                dist.init_process_group("nccl", device_id=torch.device("cuda", device))
                warm = torch.zeros(1, device="cuda")
                dist.all_reduce(warm)
                torch.cuda.synchronize()
            else:
                dist.init_process_group("gloo")
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    tensor_device = "cuda" if args.backend == "nccl" else "cpu"

    if args.serial:
        os.environ["RT_NO_OVERLAP"] = "1"  # read by rt_create
    files, label = model_files(args, rank)
    first, rows = tiling.band_of(rank, world, args.height)
    pool = S.L.align_up(args.width * args.height, S.L.SORT_TILE)
    sc = S.build_scene(files, args.width, args.height, pool, band_first=first, band_rows=rows,
                       tri_threshold=args.tri_threshold, aabb_threshold=args.aabb_threshold, grid=args.grid)
    rt = api.RayTracer(sc, device=device, warp_width=args.warp_width, free_schedule=args.schedule == "free",
                       shading=shading_flags(args), mad=args.arith == "mad")
    rt.init_frame()

    band = frame = None
    max_rows = tiling.max_band_rows(world, args.height)
    if collective:
        band = torch.zeros((max_rows, args.width, 4), dtype=torch.uint8, device=tensor_device)
        frame = torch.zeros((world, max_rows, args.width, 4), dtype=torch.uint8, device=tensor_device)
    host_band = np.zeros((rows, args.width, 4), np.uint8)

    def resolve_and_gather():
        if collective:
            if tensor_device == "cuda":
                rt.draw_frame_device(band.data_ptr())
            else:
                rt.draw_frame(host_band)
                band[:rows] = torch.from_numpy(host_band)
            dist.all_gather(list(frame.unbind(0)), band)
        else:
            rt.draw_frame_device()  # the image stays on the device, as the reference's stays in its GL texture

    def barrier():
        rt.sync()
        if collective:
            if tensor_device == "cuda":
                torch.cuda.synchronize()
            dist.barrier()
            if tensor_device == "cuda":
                torch.cuda.synchronize()

    settled, steady = settle(rt, args.settle)
    barrier()
    t_g = time.perf_counter()
    resolve_and_gather()
    barrier()
    gather_ms = (time.perf_counter() - t_g) * 1e3  # one resolve + gather, measured outside the timed region too
    keys = ("primary_spawned", "shadow_spawned", "box_tests", "tri_tests")
    # the warm-up steps come last before the timed region; the frame read-back to host memory, which leaves the GPU idle
    # for milliseconds, is measured after it
    rt.make_step(args.warmup)
    barrier()
    c0 = rt.counters()
    rt.set_profiling(1 if args.stage_timers == "timed" else 0)
    t0 = time.perf_counter()
    rt.make_step(args.steps)
    resolve_and_gather()
    barrier()
    dt = time.perf_counter() - t0
    c1 = rt.counters()
    t_h = time.perf_counter()
    rt.draw_frame(host_band)                       # the same resolve plus the copy of the band to host memory
    readback_ms = (time.perf_counter() - t_h) * 1e3
    d = {k: int(c1[k]) - int(c0[k]) for k in keys}
    if args.stage_timers == "timed":
        stage_ms, timed = rt.stage_times()
        dk = d
    else:
        # Kernel durations: the same number of steps again, right after the timed region, with HIP events
        # around every kernel and the three process kernels one after another on one stream.
        rt.set_profiling(2)
        rt.make_step(args.steps)
        rt.sync()
        stage_ms, timed = rt.stage_times()
        c2 = rt.counters()
        dk = {k: int(c2[k]) - int(c1[k]) for k in keys}
    rt.set_profiling(0)
    local = np.array([dt, d["primary_spawned"], d["shadow_spawned"], d["box_tests"], d["tri_tests"]], np.float64)
    if collective:
        t = torch.tensor(local, dtype=torch.float64, device=tensor_device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = t.cpu().numpy()
        total[0] = float(tmax[0])
    else:
        total = local
    elapsed = float(total[0])
    rays = float(total[1] + total[2])

    if rank == 0:
        # roofline of the dominant kernel on this rank
        per_launch = {k: v / max(1, timed) for k, v in stage_ms.items()}
        dominant = max(("traverse", "leaves", "shade"), key=lambda k: per_launch[k])
        kernel = {"traverse": "k_process_aabb", "leaves": "k_process_mesh", "shade": "k_process_shade"}[dominant]
        step_rays = (dk["primary_spawned"] + dk["shadow_spawned"]) / args.steps
        alg_bytes = BYTES_PER_RAY * step_rays
        if dominant == "traverse":
            alg_flop = FLOP_PER_BOX * dk["box_tests"] / args.steps
        elif dominant == "leaves":
            alg_flop = FLOP_PER_TRI * dk["tri_tests"] / args.steps
        else:
            alg_flop = FLOP_PER_SHADE * (dk["primary_spawned"] + dk["shadow_spawned"]) / args.steps
        seconds = max(per_launch[dominant] * 1e-3, 1e-12)
        # HBM bytes per launch from the hardware counters (tools/pmc.py + pmc_report.py, separate rocprofv3 --pmc
        # passes): only quoted when they were collected on this very build of the library and this schedule
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                t = json.load(open(tfile))
                if t.get("library_sha256") == library_digest() and t.get("schedule") == args.schedule and t.get("arithmetic", "strict") == args.arith:
                    traffic = t["kernels"].get(kernel)
            except Exception:
                traffic = None
        hbm = alg_bytes / seconds / 1e9
        valu = alg_flop / seconds / 1e12
        roof = [
            {"kernel": kernel, "bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": hbm / HBM_PEAK_GBS, "traffic": traffic,
             "note": "algorithmic bytes = 48 B per ray started in the step; the wavefront formulation moves far more "
                     "(ray slot + index per node visit)"},
            {"kernel": kernel, "bound": "valu", "achieved": valu, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": valu / VALU_PEAK_TFLOPS, "traffic": traffic,
             "note": "algorithmic flop = 25 per box test, 55 per triangle test, 150 per shaded sample; FP32 vector "
                     "peak, the path has no dense contraction for MFMA"},
        ]
        roof.sort(key=lambda r: -r["frac"])  # the bound that binds first
        out = {
            "metric": "Mrays/s (primary+secondary), dragon_vrip 1080p, at 1/2/4/8 MI355X",
            "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle_steps": settled, "steady": steady,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}, " + (f"{args.grid}^3 grid of instances" if args.grid else "256 instances (reference scene)")
                                   + f", {args.width}x{args.height}, {pool} ray slots per GPU, "
                                   + ("primary + shadow rays" if args.shading == "reference" else
                                      f"primary rays, {args.bounces} {args.shading} bounces, shadow rays"),
                       "tiling": f"{world} row band(s), RGBA8 all-gather per frame ({args.backend})" if collective else "single GPU",
                       "warp_width": args.warp_width, "schedule": args.schedule, "arithmetic": args.arith, "groups": sc.group_count,
                       "tri_threshold": args.tri_threshold, "aabb_threshold": args.aabb_threshold},
            "mrays_primary_only": float(total[1]) / elapsed / 1e6,
            "stage_ms_per_step": per_launch, "stage_timers": args.stage_timers, "schedule": "serial" if args.serial else "leaf kernel on a second stream", "frame_resolve_gather_ms": gather_ms,
            "frame_resolve_to_host_ms": readback_ms,
            "roofline": roof[0], "roofline_other_bound": roof[1],
            "scene_build": {"ingest_seconds": sc.ingest_seconds, "hierarchy_and_assembly_seconds": sc.build_seconds,
                            "model_files_from_binary_cache": sc.models_from_cache, "resident_triangles": sc.resident_triangles},
        }
        if world == 1 and not args.no_extras:
            rt.close()  # one full-size pool at a time
            extra = {"configs": {}}
            try:
                other = "strict" if args.arith == "mad" else "mad"
                extra["configs"][f"same workload, --arith {other}"] = quick_rate(files, args.width, args.height, args, device, arith=other)
                # a narrower hierarchy over the same leaves (inner groups of 8-15 and of 16-31 boxes instead of 128-255)
                for threshold in (8, 16):
                    if threshold != args.aabb_threshold:
                        extra["configs"][f"same workload, --aabb-threshold {threshold}"] = quick_rate(
                            files, args.width, args.height, args, device, aabb_threshold=threshold)
                if (args.width, args.height) != (3840, 2160):
                    extra["configs"]["3840x2160, same scene"] = quick_rate(files, 3840, 2160, args, device)
                if args.scene != "bunny":
                    extra["configs"][f"{args.width}x{args.height}, bun_zipper x2"] = quick_rate([S.BUNNY, S.BUNNY], args.width, args.height, args, device)
                # What eight GPUs would each get: the eight row bands of this frame, one after the other on this
                # GPU, each with a full-size pool as in the N = 8 run.  The slowest band sets the frame time.
                bands = [quick_rate(files, args.width, args.height, args, device, band=tiling.band_of(r, 8, args.height), steps=24)
                         for r in range(8)]
                ms = [b["ms_per_step"] for b in bands]
                total = sum(b["mrays_per_s"] * b["ms_per_step"] for b in bands) / max(ms)
                extra["eight_bands_on_one_gpu"] = {
                    "ms_per_step": ms, "mrays_per_s": [b["mrays_per_s"] for b in bands],
                    "predicted_8gpu_mrays_per_s": total,
                    "predicted_scaling_efficiency": total / 8 / out["value"],
                    "note": "rays of all bands per step of the slowest band, over 8 x the single-GPU rate; excludes the all-gather"}
                extra["fixed_work"] = fixed_work(files, args.width, args.height, args, device, gather_ms)
            except Exception as e:
                extra["error"] = str(e)
            out["extra"] = extra
        if world == 1 and args.cpu_seconds > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(args, files, args.cpu_seconds)
            except Exception as e:  # the GPU number stands on its own
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    rt.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
