#!/usr/bin/env python3
"""Benchmark of the wavefront path-tracing loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one launch group of the wavefront loop on every rank: --spp-per-step iterations (one sample per
pixel each) that share every kernel launch.  Ranks render disjoint, interleaved iteration indices of the same
replicated scene (iterations are independent given the iteration index, which seeds every RNG stream); after the
timed steps the per-rank running sums are summed with one all-reduce over RCCL.  value = path-bounces of all
ranks / time (path-bounce = one live path entering the intersect stage for one bounce).

--scaling weak (default): every rank renders --spp-per-step iterations per step, whatever the number of ranks.
--scaling strong --total-spp S: a step is S samples per pixel in total, S / ranks on every rank.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import math
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, resolution, depth, description)
    "dragon": ("dragon", (1920, 1080), 8,
               "dragon.json @1920x1080 depth 8 (procedural stand-in for the undistributed DragonAttenuation.glb: "
               "135200-triangle mesh, 6 PBR materials incl. one transmissive, cube light, "
               "2048x1024 procedural HDR env, intensity 0.6)"),
    "cornell": ("cornell", (800, 800), 8, "cornell.json @800x800 depth 8, env map absent (9 primitives, 13 materials)"),
    "chess": ("chess", (1920, 1080), 10, "chess.json @1920x1080 depth 10 (stand-in: 1.5M triangles, 18 textured PBR materials, HDR env)"),
    "sss": ("sss", (1920, 1080), 8, "sss.json @1920x1080 depth 8"),
    "sss32": ("sss", (1920, 1080), 32, "sss.json @1920x1080 depth 32"),
    "halo": ("halo", (3840, 2160), 8, "halo.json @3840x2160 depth 8 (stand-in: 3.0M triangles, 13 materials, two lights, HDR env)"),
    "p911": ("p911", (3840, 2160), 8, "911.json @3840x2160 depth 8 (stand-in: 3.0M triangles, 16 textured materials, light of emittance 1, HDR env intensity 3)"),
    "duck": ("duck_real", (1920, 1080), 8, "duck.json @1920x1080 depth 8: the reference's own gltf/Duck.glb (4212 triangles, one textured PBR "
                                            "material) through the glTF import, procedural HDR env in place of the undistributed one"),
}

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
CLOCK_GHZ = 2.4
# fp32 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-instructions/s (x 2 flop for an FMA, x 2 again
# for packed / dual-issue forms = the 157.3 TFLOP/s headline figure)
VALU_LANE_PEAK_GINST = 256 * 4 * 16 * CLOCK_GHZ
VALU_PEAK_TFLOPS = 157.3
L2_PEAK_GBS = 34500.0          # aggregate L2 bandwidth of the eight XCDs
TUNE_GROUPS = 5      # launch groups the library uses to pick its launch shape (one unmeasured, then 1 / 2 / 3 streams and one stream alone)


def iteration_range(rank, world, per_rank_per_step, step_begin, step_end):
    """(begin, end, stride) of pt_trace_range for the steps [step_begin, step_end) of one rank: rank r of R renders
    the iteration indices 1 + r, 1 + r + R, ...; a step is per_rank_per_step of them on every rank."""
    return (1 + rank + step_begin * per_rank_per_step * world, 1 + rank + step_end * per_rank_per_step * world, world)


def per_rank_iterations(scaling, world, spp_per_step, total_spp):
    """iterations one rank renders per step.  strong: total_spp samples per pixel are split evenly over the ranks.  The
    union of the ranks' iteration indices is 1 .. total_spp exactly as one GPU would render them, so the four
    jitter cells (iteration % 4) stay balanced when total_spp is a multiple of 4 — a single rank sees only some of the
    cells once the stride is even, which does not matter for the sum."""
    if scaling == "weak":
        return spp_per_step
    block = 4 * world // math.gcd(4, world)            # lcm(4, ranks)
    if total_spp <= 0 or total_spp % block:
        raise SystemExit("--total-spp must be a positive multiple of %d for %d ranks" % (block, world))
    return total_spp // world


def scene_directory(rank):
    """every rank writes its scene file into a directory of its own (ranks start at the same time)"""
    return os.path.join(tempfile.gettempdir(), "mi_pathtrace_bench_%d_rank%d" % (os.getuid(), rank))


def make_workload_scene(scenes, pt, name, res, depth, rank):
    if name == "duck_real":
        return scenes.make_duck_scene(res=res, depth=depth, directory=scene_directory(rank))
    return scenes.make_scene(name, res=res, depth=depth, directory=scene_directory(rank))


def _ray_sets(pt, tr, npix):
    """closest-hit stage alone on two fixed sets of one image's worth of rays (production kernels through the
    fixed-ray-set hook, device time of the traversal kernel): the primary rays of iteration 1, and rays leaving the
    third hit of a chain of cosine-weighted bounces that starts with them"""
    import numpy as np
    rng = np.random.default_rng(12345)

    def cosine_about(n):
        n = n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
        u1, u2 = rng.random(len(n), dtype=np.float32), rng.random(len(n), dtype=np.float32)
        r, phi = np.sqrt(u1), 2 * np.pi * u2
        a = np.where(np.abs(n[:, :1]) < 0.577, np.array([[1.0, 0, 0]], np.float32), np.array([[0, 1.0, 0]], np.float32))
        t = np.cross(n, a)
        t /= np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-20)
        b = np.cross(n, t)
        return (t * (r * np.cos(phi))[:, None] + b * (r * np.sin(phi))[:, None] + n * np.sqrt(np.maximum(0.0, 1.0 - u1))[:, None]).astype(np.float32)

    def timed(o, d):
        tr.intersect(o, d)
        tr.set_option(pt.PT_OPT_STAGE_TIMING, 1)
        tr.reset_stats()
        h = tr.intersect(o, d)
        st = tr.stats()
        tr.set_option(pt.PT_OPT_STAGE_TIMING, 0)
        ms = st.msTrace + st.msIntersect
        return h, (round(len(o) / ms / 1e3, 1) if ms > 0 else None), (round(len(o) / st.msTrace / 1e3, 1) if st.msTrace > 0 else None)

    o, d = tr.camera_rays(1)
    h, primary_stage, primary_trav = timed(o, d)
    out = {"rays_per_set": int(npix), "primary_Mrays_s": primary_stage, "primary_traversal_Mrays_s": primary_trav}
    if not (h["t"] > 0).any():
        return out
    for bounce in range(3):
        idx = np.flatnonzero(h["t"] > 0)
        sel = idx[rng.integers(0, len(idx), npix)]
        nrm = h["normal"][sel]
        o, d = (o[sel] + d[sel] * h["t"][sel, None] + nrm * 1e-3).astype(np.float32), cosine_about(nrm)
        if bounce < 2:
            h = tr.intersect(o, d)
    _h, stage, trav = timed(o, d)
    out.update({"bounce3_Mrays_s": stage, "bounce3_traversal_Mrays_s": trav})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="iterations sharing one launch group = one step (1..512); 0 = the largest power of two whose path state "
                         "stays below 210 GB (512 at 1920x1080, 128 at 3840x2160)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--total-spp", type=int, default=512, help="--scaling strong: samples per pixel of one step, split over the ranks")
    ap.add_argument("--workload", default="dragon", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-iteration and ray-set measurements after the timed region")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (one GPU per rank); gloo only to rehearse several ranks on a box with fewer GPUs")
    ap.add_argument("--no-sort", action="store_true", help="MATERIAL_SORTING 0")
    ap.add_argument("--no-collective-at-one", action="store_true", help="one rank: do not set up the process group / run the all-reduce")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline: whole iterations until this much time has passed")
    args = ap.parse_args()

    if os.environ.get("BENCH_WATCHDOG"):        # diagnosis: every thread's traceback on stderr after this many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=True)
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
    collective = None           # what reduces the per-rank sums: "rccl" / "gloo", or why there is none
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            collective = "rccl"
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            collective = "gloo"
    elif args.backend == "nccl" and not args.no_collective_at_one:
        # one rank: the process group and the all-reduce of the running sum run all the same (a sum over one rank), so that
        # the path the multi-GPU runs take — RCCL on a tensor the tracer is bound to — has executed on this machine too
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            collective = "rccl"
        except Exception as e:      # no usable network interface for the bootstrap, say: the single-GPU number does not depend on it
            collective = "unavailable at one rank: %s" % str(e)[:160]

    scene_name, res, depth, desc = WORKLOADS[args.workload]
    sc = make_workload_scene(scenes, pt, scene_name, res, depth, rank)
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
    # and not per iteration, so groups are as large as 220 GB of path state (203 B per pixel and iteration) allow,
    # in powers of two so that the persistent traversal grid divides evenly over the rows
    budget = min(220e9, 0.8 * torch.cuda.mem_get_info()[0])       # never more than four fifths of what is free now
    per_step = per_rank_iterations(args.scaling, world, args.spp_per_step, args.total_spp)
    B = args.spp_per_step if args.spp_per_step > 0 else next(b for b in (512, 256, 128, 64, 32, 16, 8) if b == 8 or npix * 203.0 * b <= budget)
    if args.scaling == "strong":
        B = min(B, per_step)
    B = max(1, min(B, 512))

    # ---- untimed: context set-up (the library tries its launch shapes on the first five full launch groups and
    # keeps the fastest), warmup, then a counted pass over the timed iterations (exact node / triangle counts)
    while True:
        try:
            tr.set_option(pt.PT_OPT_BATCH, B)         # allocates the path state of B iterations
            break
        except pt.PtError:
            if args.spp_per_step > 0 or B <= 8:
                raise
            B //= 2                                   # a GPU with less free memory than planned for: smaller groups
    if world > 1:                                     # every rank uses the same group size
        t = torch.tensor([B], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) != B:
            B = int(t.item())
            tr.set_option(pt.PT_OPT_BATCH, B)
    if args.scaling == "weak":
        per_step = B

    def steps_range(a, b):
        return iteration_range(rank, world, per_step, a, b)

    tr.trace_range(*iteration_range(rank, world, B, 0, TUNE_GROUPS))
    barrier()
    tr.clear()
    if args.warmup:
        tr.trace_range(*steps_range(0, args.warmup))
    barrier()
    tr.reset_stats()
    tr.clear()
    tr.set_option(pt.PT_OPT_COUNTERS, 1)
    tr.trace_range(*steps_range(args.warmup, args.warmup + 1))
    cs = tr.stats()
    counted = dict(pb=cs.pathBounces, rays=cs.tracedRays, shadow_rays=cs.shadowRays,
                   boxes=cs.boxesTested - cs.shadeBoxesTested, tris=cs.trisTested - cs.shadeTrisTested, hits=cs.meshHits - cs.shadeMeshHits,
                   s_boxes=cs.shadeBoxesTested, s_tris=cs.shadeTrisTested, max_stack=cs.maxStackDepth, ref_walks=cs.referenceOrderWalks)
    tr.set_option(pt.PT_OPT_COUNTERS, 0)
    tr.reset_stats()
    tr.clear()

    # ---- timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    tr.trace_range(*steps_range(args.warmup, args.warmup + args.steps))
    tr.sync()
    if collective in ("rccl", "gloo"):
        dist.all_reduce(image)                      # RCCL sum of the W*H*3 accumulation buffers (at one rank: over itself)
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
    tr.trace_range(*steps_range(args.warmup, args.warmup + args.steps))
    tr.sync()
    p1 = time.perf_counter()
    st = tr.stats()
    tr.set_option(pt.PT_OPT_STAGE_TIMING, 0)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    pbs = torch.tensor([float(total_stats.pathBounces)], dtype=torch.float64, device="cuda")
    if world > 1:
        if args.backend == "gloo":
            elapsed, pbs = elapsed.cpu(), pbs.cpu()
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(pbs, op=dist.ReduceOp.SUM)
    elapsed = float(elapsed.item())
    total_pb = float(pbs.item())
    spp_rank = args.steps * per_step               # iterations one rank rendered in the timed region

    out = None
    if rank == 0:
        value = total_pb / elapsed / 1e6
        stages = {"k_trace(closest)": st.msTrace, "k_trace(shadow)": st.msShadowTrace,
                  "k_prims": st.msIntersect - st.msTrace, "k_camera": st.msCamera,
                  "k_shade": st.msShade - st.msShadowTrace, "sort": st.msSort, "compact": st.msCompact}
        measured_in = ("a second pass over the same %d steps with HIP events around every stage and the launch group on one "
                       "stream (%.3f ms per spp; the timed region runs each group on the one to three streams the library picked during set-up)"
                       % (args.steps, (p1 - p0) / max(spp_rank, 1) * 1e3))
        pmc = {}
        ppath = os.path.join(ROOT, "profiles", "pmc_%s.json" % args.workload)
        if os.path.exists(ppath):
            try:
                pmc = json.load(open(ppath))
            except Exception:
                pmc = {}
        if st.launchesTrace > 0:
            # ---- roofline of the dominant kernel, the BVH traversal in closest-hit mode.
            # Its data (nodes, triangles) are served by L2 / Infinity Cache, so HBM is not what bounds it: the binding
            # resource is vector instruction issue under lane divergence.  achieved = fp32-lane instructions the kernel
            # retires per second = VALU wave-instructions per launch x active lanes per instruction (hardware counters
            # of a profiled run of this workload, profiles/pmc_<workload>.json) / this run's own launch duration (HIP
            # events); peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz.  frac is therefore issue utilisation x lane
            # utilisation.  traffic = HBM bytes per launch from the same profile.
            per_ray = 1.0 / max(counted["rays"], 1)
            boxes, tris, hits = counted["boxes"] * per_ray, counted["tris"] * per_ray, counted["hits"] * per_ray
            b_ray = 80.0 + 32.0 * boxes + 36.0 * tris + 64.0 * hits
            rays_per_launch = st.tracedRays / max(st.launchesTrace, 1)
            avg_ms = st.msTrace / max(st.launchesTrace, 1)
            alg_gbs = (b_ray * rays_per_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            flops_ray = 30.0 * boxes + 53.0 * tris + 36.0 + 120.0 * hits
            valu_tflops = (flops_ray * rays_per_launch) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            k = pmc.get("k_trace(closest)", {})
            # the counters are per ray handed to the kernel, so that a run with another group size can use them
            lane_inst_per_ray = k.get("valu_lane_instructions_per_ray")
            traffic_per_ray = k.get("hbm_bytes_per_ray")
            achieved = (lane_inst_per_ray * rays_per_launch) / (avg_ms * 1e-3) / 1e9 if (lane_inst_per_ray and avg_ms > 0) else None
            traffic = round(traffic_per_ray * rays_per_launch) if traffic_per_ray else None
            util = round(achieved / VALU_LANE_PEAK_GINST, 4) if achieved else None
            roofline = {"bound": "valu", "kernel": "k_trace<closest>",
                        "achieved": round(valu_tflops, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(valu_tflops / VALU_PEAK_TFLOPS, 4),
                        "utilisation": util,
                        "utilisation_detail": {"lane_instructions_G_per_s": round(achieved, 1) if achieved else None,
                                               "peak_G_per_s": round(VALU_LANE_PEAK_GINST, 1),
                                               "issue_utilisation": k.get("valu_issue_utilisation"),
                                               "lane_utilisation": k.get("valu_lane_utilisation")},
                        "l2_view": {"achieved": round(alg_gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(alg_gbs / L2_PEAK_GBS, 4)},
                        "traffic": traffic,
                        "traffic_gbs": (round(traffic / (avg_ms * 1e-3) / 1e9, 1) if (traffic and avg_ms > 0) else None),
                        "hbm_frac": (round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (traffic and avg_ms > 0) else None),
                        "l2_hit_rate": k.get("l2_hit_rate"),
                        "counters_from": ("profiles/pmc_%s.json (rocprofv3 --pmc passes of this workload)" % args.workload) if k else None,
                        "note": "frac = algorithmic flops (30 per child box slab-tested, 53 per triangle tested, 36 per ray-to-object-space transform, "
                                "120 per hit epilogue) / launch duration / the 157.3 TFLOP/s fp32 vector peak; utilisation = share of the machine's "
                                "fp32 lane-instruction slots the kernel fills (issue utilisation x lane utilisation, from the counters: it counts every "
                                "vector instruction, addressing and stack traffic included); l2_view = algorithmic bytes (32 per child box, 36 per "
                                "triangle, 64 per mesh hit, 80 of path state and hit record per ray) / launch duration / the 34.5 TB/s aggregate L2 "
                                "bandwidth - the BVH is served by L1 / L2 / Infinity Cache, so HBM is not the ceiling (hbm_frac = counter-measured HBM "
                                "traffic / 8 TB/s)",
                        "alg_bytes_per_ray": round(b_ray, 1), "alg_flops_per_ray": round(flops_ray, 1), "rays_per_launch": round(rays_per_launch, 1),
                        "avg_launch_ms": round(avg_ms, 5), "launches": st.launchesTrace,
                        "boxes_per_ray": round(boxes, 2), "tris_per_ray": round(tris, 2), "mesh_hit_rate": round(hits, 3),
                        "deepest_stack": counted["max_stack"], "reference_order_walks_per_million_rays": round(1e6 * counted["ref_walks"] / max(counted["rays"] + counted["shadow_rays"], 1), 2),
                        "stage_ms_per_spp": {k2: round(v / max(spp_rank, 1), 4) for k2, v in stages.items()},
                        "measured_in": measured_in}
        else:
            # analytic scenes (no BVH mesh): the streaming kernels are all there is; the larger of the two stages is reported
            # with its share of the 192 streaming bytes per path-bounce (intersect 84 = path read 48 + hit written 28 + key and
            # index 8; shade 108 = path written 48 + hit read 28 + key and index read 8 + image 24)
            dom, ms_dom, launches, b_pb = (("k_shade", st.msShade, st.launchesShade, 108.0) if st.msShade >= st.msIntersect
                                           else ("k_prims", st.msIntersect, st.launchesIntersect, 84.0))
            avg_ms = ms_dom / max(launches, 1)
            units = st.pathBounces / max(launches, 1)
            achieved = b_pb * units / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            # traffic = HBM bytes per launch from a profiled run of this workload (profiles/pmc_<workload>.json, per path-bounce)
            k = pmc.get(dom, {})
            traffic_pb = k.get("hbm_bytes_per_path_bounce")
            traffic = round(traffic_pb * units) if traffic_pb else None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "traffic_gbs": (round(traffic / (avg_ms * 1e-3) / 1e9, 1) if (traffic and avg_ms > 0) else None),
                        "hbm_frac": (round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (traffic and avg_ms > 0) else None),
                        "issue_utilisation": k.get("valu_issue_utilisation"), "lane_utilisation": k.get("valu_lane_utilisation"),
                        "l2_hit_rate": k.get("l2_hit_rate"),
                        "counters_from": ("profiles/pmc_%s.json (rocprofv3 --pmc passes of this workload)" % args.workload) if k else None,
                        "note": "frac prices the algorithmic bytes of the stage against HBM; the stage reaches its paths through a sorted "
                                "queue and gathers 16-byte records, each of which pulls a whole line, so the HBM traffic the counters see "
                                "(traffic, hbm_frac) is several times the algorithmic bytes",
                        "alg_bytes_per_path_bounce": b_pb, "path_bounces_per_launch": round(units, 1),
                        "avg_launch_ms": round(avg_ms, 5), "launches": launches,
                        "stage_ms_per_spp": {k2: round(v / max(spp_rank, 1), 4) for k2, v in stages.items()}, "measured_in": measured_in}

        live = [int(v) for v in st.lastLive[:depth + 1]]
        extras = {}
        if not args.no_extras and world == 1:
            # ---- the reference-shaped call: one iteration per call of pathtrace(pbo = NULL, frame, iteration)
            n_single = 48
            # the library's rule: four calls in flight, six / eight when the process has 8 / 16 hardware queues; PT_PIPELINE fixes it
            hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
            calls_in_flight = int(os.environ.get("PT_PIPELINE", "0") or 0) or (8 if hwq >= 16 else 6 if hwq >= 8 else 4)
            tr.clear()
            tr.trace(1)
            tr.sync()
            tr.reset_stats()
            s0 = time.perf_counter()
            for it in range(2, 2 + n_single):
                tr.trace(it)
            tr.sync()
            s1 = time.perf_counter()
            ss = tr.stats()
            w0 = time.perf_counter()
            for it in range(2 + n_single, 2 + n_single + 16):
                tr.trace(it)
                tr.sync()
            w1 = time.perf_counter()
            extras["single_iteration"] = {"ms_per_iter": round((s1 - s0) / n_single * 1e3, 4),
                                          "Mpaths_bounce_s": round(ss.pathBounces / (s1 - s0) / 1e6, 1), "iterations": n_single,
                                          "ms_per_iter_waiting_after_every_call": round((w1 - w0) / 16 * 1e3, 4),
                                          "path_state_bytes_in_flight": int(192 * npix) * calls_in_flight,
                                          "calls_in_flight": calls_in_flight,
                                          "how": "pt_trace(iteration) once per sample per pixel, no host copy, one wait at the end: up to calls_in_flight calls "
                                                 "overlap on the device (a row of the path state and a stream each, accumulation in call order); "
                                                 "the second figure has a host wait after every call, nothing overlaps"}
            # ---- the same loop at the rate of launch groups: pt_trace_lookahead renders the iterations behind the one asked for in the
            # same launch group and hands them out one per call; the running sum is copied into a page-locked host image after
            # every call, as the reference's pathtrace() does (src/pathtrace.cu:2913-2917).  This is the call sequence of
            # integration/pathtrace_mi355x.cpp (the file a maintainer of the reference adds), made through the C ABI.
            import numpy as np
            ahead = int(os.environ.get("PT_LOOKAHEAD", "128") or 128)
            ahead = max(2, min(ahead, B // 2))
            host = np.zeros((res[1], res[0], 3), np.float32)
            tr.set_option(pt.PT_OPT_LAUNCH_STREAMS, 1)
            tr.clear()
            tr.bind_host_image(host)
            n_calls = 4 * ahead
            f0 = time.perf_counter()
            tr.trace_lookahead(1, ahead)
            tr.host_image_wait()
            f1 = time.perf_counter()
            for it in range(2, 2 + n_calls):
                tr.trace_lookahead(it, ahead)
                tr.host_image_wait()                    # the host image is read after every call
            f2 = time.perf_counter()
            for it in range(2 + n_calls, 2 + 2 * n_calls):
                tr.trace_lookahead(it, ahead)           # the host copy is queued behind every call, waited for once
            tr.host_image_wait()
            f3 = time.perf_counter()
            check = tr.image()
            ok = bool((host.view(np.uint32) == check.view(np.uint32)).all())
            tr.sync()
            tr.bind_host_image(None)
            tr.set_option(pt.PT_OPT_LAUNCH_STREAMS, 0)
            extras["pathtrace_call"] = {"ms_per_call_host_image_current_after_every_call": round((f2 - f1) / n_calls * 1e3, 4),
                                        "ms_per_call_host_copy_queued": round((f3 - f2) / n_calls * 1e3, 4),
                                        "first_call_ms": round((f1 - f0) * 1e3, 2), "calls": n_calls, "iterations_rendered_ahead": ahead,
                                        "host_copy_bytes_per_call": int(npix * 12), "host_image_equals_device_sum": ok,
                                        "path_state_bytes": int(203 * npix) * 2 * ahead,
                                        "how": "pt_trace_lookahead(iteration, ahead) + pt_host_image_wait once per sample per pixel: the call sequence of "
                                               "pathtrace(pbo = NULL, frame, iteration) in integration/pathtrace_mi355x.cpp, device-to-host copy of the running "
                                               "sum (W*H*12 bytes) included; the first call renders a whole group before it returns"}
            extras["path_state_bytes_in_flight"] = int(192 * npix) * B
            if st.launchesTrace > 0:
                extras["ray_sets"] = _ray_sets(pt, tr, npix)

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
            # the reference's OWN intersection code (src/intersections.cu + src/bvh.cpp, compiled where they lie as
            # oracle/_ref/ref_tool) on the camera rays of one iteration: the closest-hit stage alone, in rays per second
            try:
                import ref_binding as refb
                if refb.available() and st.launchesTrace > 0:
                    ro, rd = tr.camera_rays(1)
                    m = min(len(ro), 400000)
                    timing = {}
                    refb.intersect(sc, ro[:m], rd[:m], timing=timing)
                    if timing:
                        cpu["reference_closest_hit"] = {"value": round(timing["rays"] / timing["seconds"] / 1e6, 3), "unit": "Mrays/s", "cores": timing["threads"],
                                                        "kind": "reference",
                                                        "sample": "%d camera rays of iteration 1 through the reference's computeIntersectionsBVH loop "
                                                                  "(its own boxIntersectionTest / meshIntersectionTestBVH / BVH::build), OpenMP over rays" % timing["rays"]}
            except Exception as e:          # the checker's tool is optional
                cpu["reference_closest_hit"] = {"error": str(e)[:200]}

        pb_spp = st.pathBounces / max(spp_rank, 1)
        out = {"metric": "Mpaths*bounce/s (%s)" % args.workload, "value": round(value, 3), "unit": "Mpaths*bounce/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": desc, "resolution": list(res), "depth": depth, "spp_per_step_per_gpu": per_step,
                          "iterations_per_launch_group": B,
                          "ms_per_spp": round(elapsed / max(spp_rank * world, 1) * 1e3, 5),
                          "material_sort": not args.no_sort, "sample_split": "iteration index interleaved over ranks, one RCCL all-reduce of the W*H*3 sum",
                          "collective": collective,
                          "path_bounces_per_spp": round(pb_spp, 1),
                          "live_paths_per_bounce_last_iteration": live,
                          "secondary_fraction_of_path_bounces": round(1.0 - npix / max(pb_spp, 1.0), 4),
                          "traversal_rays_per_spp": round(st.tracedRays / max(spp_rank, 1), 1),
                          "shadow_rays_per_spp": round(st.shadowRays / max(spp_rank, 1), 1)},
               "roofline": roofline, "cpu_baseline": cpu}
        out.update(extras)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
