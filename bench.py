#!/usr/bin/env python3
"""Benchmark of the wavefront path-tracing loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one launch group of the wavefront loop on every rank: --spp-per-step iterations (one sample per
pixel each, default 128) that share every kernel launch.  Ranks render disjoint, interleaved iteration
indices of the same replicated scene (iterations are independent given the iteration index, which seeds
every RNG stream); after the timed steps the per-rank running sums are summed with one all-reduce over
RCCL.  value = path-bounces of all ranks / time (path-bounce = one live
path entering the intersect stage for one bounce).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, resolution, depth, description)
    "dragon": ("dragon", (1920, 1080), 8,
               "dragon.json @1920x1080 depth 8 (procedural stand-in for the undistributed DragonAttenuation.glb: "
               "135200-triangle closed mesh, 6 PBR materials incl. one transmissive, cube light, "
               "2048x1024 procedural HDR env, intensity 0.6)"),
    "cornell": ("cornell", (800, 800), 8, "cornell.json @800x800 depth 8, env map absent (9 primitives, 13 materials)"),
    "chess": ("chess", (1920, 1080), 10, "chess.json @1920x1080 depth 10 (stand-in: 1.5M triangles, 18 textured PBR materials, HDR env)"),
    "sss": ("sss", (1920, 1080), 8, "sss.json @1920x1080 depth 8"),
    "sss32": ("sss", (1920, 1080), 32, "sss.json @1920x1080 depth 32"),
    "halo": ("halo", (3840, 2160), 8, "halo.json @3840x2160 depth 8 (stand-in: 3.0M triangles, 13 materials, two lights, HDR env)"),
}

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak (256 CUs x 128 lanes x 2 flop x 2.4 GHz)


TUNE_GROUPS = 5      # launch groups the library uses to pick its launch shape (one unmeasured, then 1 / 2 / 3 streams and one stream alone)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="iterations sharing one launch group = one step (1..512); 0 = the largest power of two whose path state "
                         "stays below 210 GB (512 at 1920x1080, 128 at 3840x2160)")
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (one GPU per rank); gloo only to rehearse several ranks on a box with fewer GPUs")
    ap.add_argument("--no-sort", action="store_true", help="MATERIAL_SORTING 0")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline: whole iterations until this much time has passed")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    pt = entry._load_package()
    scenes = importlib.import_module("cuda_path_tracer_amd.scenes")

    if args.backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)       # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    scene_name, res, depth, desc = WORKLOADS[args.workload]
    sc = scenes.make_scene(scene_name, res=res, depth=depth)
    tr = pt.Tracer(sc, device=local_rank)
    if args.no_sort:
        tr.set_option(pt.PT_OPT_MATERIAL_SORT, 0)
    npix = res[0] * res[1]
    image = torch.zeros(npix * 3, dtype=torch.float32, device="cuda")      # running sum lives in a torch tensor
    tr.bind_image(image.data_ptr())

    def barrier():
        tr.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # iterations per launch group: every kernel launch ends with the few longest rays of its bounce, a cost per launch
    # and not per iteration, so groups are as large as 210 GB of path state (about 190 B per pixel and iteration) allow,
    # in powers of two so that the persistent traversal grid divides evenly over the rows
    budget = min(210e9, 0.8 * torch.cuda.mem_get_info()[0])       # never more than four fifths of what is free now
    B = args.spp_per_step if args.spp_per_step > 0 else next(b for b in (512, 256, 128, 64, 32, 16, 8) if b == 8 or npix * 190.0 * b <= budget)
    B = max(1, min(B, 512))

    def it_index(step):          # interleaved sample split: rank r renders 1+r, 1+r+R, ...; B of them per step
        return 1 + rank + step * B * world

    # ---- untimed: context set-up (the library tries its launch shapes on the first five launch groups of a size and
    # keeps the fastest), warmup, then a counted pass over the timed iterations (exact node / triangle counts)
    while True:
        try:
            tr.set_option(pt.PT_OPT_BATCH, B)         # allocates the path state of B iterations
            break
        except pt.PtError:
            if args.spp_per_step > 0 or B <= 8:
                raise
            B //= 2                                   # a GPU with less free memory than planned for: smaller groups
    if world > 1:                                     # every rank uses the same group size (it fixes the iteration indices)
        t = torch.tensor([B], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) != B:
            B = int(t.item())
            tr.set_option(pt.PT_OPT_BATCH, B)
    tr.trace_range(it_index(0), it_index(TUNE_GROUPS), world)
    barrier()
    tr.clear()
    if args.warmup:
        tr.trace_range(it_index(0), it_index(args.warmup), world)
    barrier()
    tr.reset_stats()
    tr.clear()
    tr.set_option(pt.PT_OPT_COUNTERS, 1)
    count_steps = 1
    tr.trace_range(it_index(args.warmup), it_index(args.warmup + count_steps), world)
    cs = tr.stats()
    counted = dict(pb=cs.pathBounces, rays=cs.tracedRays, shadow_rays=cs.shadowRays,
                   boxes=cs.boxesTested - cs.shadeBoxesTested, tris=cs.trisTested - cs.shadeTrisTested, hits=cs.meshHits - cs.shadeMeshHits,
                   s_boxes=cs.shadeBoxesTested, s_tris=cs.shadeTrisTested)
    tr.set_option(pt.PT_OPT_COUNTERS, 0)
    tr.reset_stats()
    tr.clear()

    # ---- timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    tr.trace_range(it_index(args.warmup), it_index(args.warmup + args.steps), world)     # exactly K iterations per rank
    tr.sync()
    if world > 1:
        dist.all_reduce(image)                      # RCCL sum of the W*H*3 accumulation buffers
    barrier()
    t1 = time.perf_counter()
    total_stats = tr.stats()

    # ---- kernel durations for the roofline: the same K steps once more with every stage bracketed by HIP events
    # on the tracer's stream (with stage timing on, the library keeps a launch group on ONE stream, so a kernel's
    # duration is that of the kernel running alone; the timed region above runs each group on the one to three streams
    # the library picked)
    tr.reset_stats()
    tr.set_option(pt.PT_OPT_STAGE_TIMING, 1)
    p0 = time.perf_counter()
    tr.trace_range(it_index(args.warmup), it_index(args.warmup + args.steps), world)
    tr.sync()
    p1 = time.perf_counter()
    st = tr.stats()
    tr.set_option(pt.PT_OPT_STAGE_TIMING, 0)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    pbs = torch.tensor([float(total_stats.pathBounces)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(pbs, op=dist.ReduceOp.SUM)
    elapsed = float(elapsed.item())
    total_pb = float(pbs.item())

    out = None
    if rank == 0:
        value = total_pb / elapsed / 1e6
        # ---- roofline of the dominant kernel (by accumulated HIP-event time): the BVH traversal kernel in
        # closest-hit mode.  Algorithmic bytes per path-bounce of the whole loop = 192 streaming bytes
        # (path state read + written 96, hit record written + read 56, sort key + index 16, image 24)
        # + 32 B per child box tested + 36 B per triangle tested + 64 B of attributes per mesh hit.
        # The traversal kernel's share, per ray it is handed: 4 B queue entry + 48 B path state read
        # + 28 B hit record written, plus its own box / triangle / attribute bytes (exact counts from
        # the counted pass over the first iterations of the timed range).
        stages = {"k_trace(closest)": st.msTrace, "k_trace(shadow)": st.msShadowTrace,
                  "k_prims": st.msIntersect - st.msTrace,
                  "k_shade": st.msShade - st.msShadowTrace, "sort": st.msSort, "compact": st.msCompact}
        per_ray = 1.0 / max(counted["rays"], 1)
        boxes, tris, hits = counted["boxes"] * per_ray, counted["tris"] * per_ray, counted["hits"] * per_ray
        b_ray = 80.0 + 32.0 * boxes + 36.0 * tris + 64.0 * hits
        rays_per_launch = st.tracedRays / max(st.launchesTrace, 1)
        avg_ms = st.msTrace / max(st.launchesTrace, 1)
        achieved = (b_ray * rays_per_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # fp32 VALU view of the same kernel: 30 flop per box, 53 per triangle, 36 + 120 per ray (set-up, epilogue)
        flops_ray = 30.0 * boxes + 53.0 * tris + 36.0 + 120.0 * hits
        valu_tflops = (flops_ray * rays_per_launch) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        b_loop = 192.0 + (32.0 * (counted["boxes"] + counted["s_boxes"]) + 36.0 * (counted["tris"] + counted["s_tris"])
                          + 64.0 * counted["hits"]) / max(counted["pb"], 1)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_trace(closest)")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "k_trace<closest>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_gbs": (round(traffic / (avg_ms * 1e-3) / 1e9, 1) if (traffic and avg_ms > 0) else None),
                    "note": "algorithmic bytes follow the per-unit definition (32 B per child box tested, 36 B per triangle, 64 B per mesh "
                            "hit, 80 B of path state and hit record per ray); the BVH stays in L2 / Infinity Cache, so the HBM traffic the "
                            "counters see (traffic, traffic_gbs) is far below it and the kernel is bound by dependent-fetch latency and "
                            "SIMD lane divergence, not by HBM",
                    "alg_bytes_per_ray": round(b_ray, 1), "rays_per_launch": round(rays_per_launch, 1),
                    "avg_launch_ms": round(avg_ms, 5), "launches": st.launchesTrace,
                    "boxes_per_ray": round(boxes, 2), "tris_per_ray": round(tris, 2), "mesh_hit_rate": round(hits, 3),
                    "valu_fp32_tflops": round(valu_tflops, 2), "valu_fp32_peak_tflops": VALU_PEAK_TFLOPS,
                    "valu_frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4),
                    "loop_alg_bytes_per_path_bounce": round(b_loop, 1),
                    "loop_alg_gbs": round(b_loop * total_pb / world / elapsed / 1e9, 1),
                    "stage_ms_per_spp": {k: round(v / (args.steps * B), 4) for k, v in stages.items()},
                    "measured_in": "a second pass over the same %d steps with HIP events around every stage and the launch group on one "
                                   "stream (%.3f ms per spp; the timed region runs each group on the one to three streams the library picked during set-up)"
                                   % (args.steps, (p1 - p0) / (args.steps * B) * 1e3)}

        if st.launchesTrace == 0:
            # analytic scenes (no BVH mesh): the streaming kernels are all there is; the larger of the two stages is reported
            # with its share of the 192 streaming bytes per path-bounce (intersect 84 = path read 48 + hit written 28 + key and
            # index 8; shade 108 = path written 48 + hit read 28 + key and index read 8 + image 24)
            dom, ms_dom, launches, b_pb = (("k_shade", st.msShade, st.launchesShade, 108.0) if st.msShade >= st.msIntersect
                                           else ("k_prims", st.msIntersect, st.launchesIntersect, 84.0))
            avg_ms = ms_dom / max(launches, 1)
            units = st.pathBounces / max(launches, 1)
            achieved = b_pb * units / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                        "alg_bytes_per_path_bounce": b_pb, "path_bounces_per_launch": round(units, 1),
                        "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                        "stage_ms_per_spp": roofline["stage_ms_per_spp"], "measured_in": roofline["measured_in"]}

        cpu = None
        if not args.no_cpu_baseline and world == 1:       # reported at N = 1 only
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as orc                    # the checker, timed as the CPU baseline only
            o = orc.Oracle(pt, sc, orc.MATH_LIBM)
            c0 = time.perf_counter()
            cpu_spp = 0
            while cpu_spp < 1 or (time.perf_counter() - c0 < args.cpu_seconds and cpu_spp < 4096):
                cpu_spp += 1
                o.trace(cpu_spp)
            c1 = time.perf_counter()
            os_ = o.stats()
            cpu = {"value": round(os_.pathBounces / (c1 - c0) / 1e6, 4), "unit": "Mpaths*bounce/s", "cores": orc.threads(),
                   "kind": "port",
                   "sample": "%d spp of the same workload (%d path-bounces, %.1f s) on the CPU restatement of the reference's loop, "
                             "OpenMP over queue slots, libm arithmetic" % (cpu_spp, os_.pathBounces, c1 - c0)}
            o.close()

        out = {"metric": "Mpaths*bounce/s (%s)" % scene_name, "value": round(value, 3), "unit": "Mpaths*bounce/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": desc, "resolution": list(res), "depth": depth, "spp_per_step_per_gpu": B,
                          "ms_per_spp": round(elapsed / (args.steps * B) * 1e3, 5),
                          "material_sort": not args.no_sort, "sample_split": "iteration index interleaved over ranks, one RCCL all-reduce of the W*H*3 sum",
                          "path_bounces_per_spp": round(st.pathBounces / max(args.steps * B, 1), 1)},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
