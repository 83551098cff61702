#!/usr/bin/env python3
"""Per-kernel hardware-counter summary of one workload from the passes of tools/profile_r03.sh.

    python tools/pmc_workload.py <dir with sq/ sq2/ l2/ fetch/ write/ and *_bench.json> <workload>

Writes profiles/pmc_<workload>.json (per-ray figures bench.py reads for its roofline) and prints a table.

Only the dispatches of the last pass of bench.py are used: the event-bracketed pass that runs every launch group on
one stream (steps x depth launches of each traversal kernel).  Counter arithmetic:
  issue utilisation  = SQ_INSTS_VALU x 4 cycles / (SIMDs x GRBM_GUI_ACTIVE-equivalent cycles) is not used (clock varies);
                       instead SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES x SIMDs per SE share) is reported as measured by
                       the SQ itself: ACTIVE_INST_VALU / WAVE_CYCLES gives the share of a resident wave's time in which
                       it issues vector ALU work; multiplied by the resident waves per SIMD it is the SIMD's issue share
  lane utilisation   = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
  lane instructions  = SQ_INSTS_VALU x 64 x lane utilisation
  HBM bytes          = (2 x FETCH_SIZE + WRITE_SIZE) x 1024   (FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950)
  L2 hit rate        = TCC_HIT / (TCC_HIT + TCC_MISS)
  L1 tag pipe        = TCP_TOTAL_CACHE_ACCESSES / (256 CUs x cycles): the vector L1 looks up one line per cycle; a 16-byte-per-lane
                       load of a full wave costs at least 16 lookups however few lines it touches
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = [("k_trace<0, false>", "k_trace(closest)"), ("k_trace<1, false>", "k_trace(shadow)"), ("k_shade<false, false>", "k_shade"),
         ("k_prims<false, false>", "k_prims"), ("k_shade<false, true>", "k_shade(sparse form)"), ("k_prims<false, true>", "k_prims(sparse form)"),
         ("k_camera", "k_camera")]


def nice(kernel):
    for pat, n in NAMES:
        if pat in kernel:
            return n
    return None


def load(root):
    """{kernel: {dispatch id: {counter: value, '_ns': duration}}}"""
    d = defaultdict(lambda: defaultdict(dict))
    # one run per pass: gpurun merges the files of every run into the same local directory, so only the newest counts
    files = sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            k = nice(r["Kernel_Name"])
            if not k:
                continue
            e = d[k][int(r["Dispatch_Id"])]
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            e["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


def last_sum(per_dispatch, n):
    ids = sorted(per_dispatch)[-n:]
    out = defaultdict(float)
    for i in ids:
        for c, v in per_dispatch[i].items():
            out[c] += v
    out["_launches"] = len(ids)
    return out


def main():
    base, workload = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
    bench = json.loads(open(os.path.join(base, "sq_bench.json")).read().strip().splitlines()[-1])
    rl = bench["roofline"]
    depth, steps = bench["config"]["depth"], bench["steps"]
    n = steps * depth
    spp = bench["config"]["spp_per_step_per_gpu"] * steps
    pb_total = bench["config"].get("path_bounces_per_spp", 0.0) * spp
    rays = {}
    if "rays_per_launch" in rl:         # scenes without a BVH mesh have no traversal launches
        rays = {"k_trace(closest)": rl["rays_per_launch"] * rl["launches"], "k_trace(shadow)": bench["config"]["shadow_rays_per_spp"] * spp}
    passes = {p: load(os.path.join(base, p)) for p in ("sq", "sq2", "l2", "fetch", "write", "ea")}
    res = {"_workload": workload, "_bench_line_of_the_sq_pass": {k: bench[k] for k in ("value", "ms_per_step")},
           "_method": "rocprofv3 --pmc, one pass per counter group (tools/profile_r03.sh), sums over the last %d dispatches of each kernel "
                      "(the single-stream pass of bench.py --steps %d); see tools/pmc_workload.py for the arithmetic" % (n, steps)}
    lines = []
    for _pat, k in NAMES:
        if k not in passes["sq"]:
            continue
        nk = n if "trace" in k or k in ("k_shade", "k_prims") else steps
        sq, sq2, l2 = last_sum(passes["sq"][k], nk), last_sum(passes["sq2"].get(k, {}), nk), last_sum(passes["l2"].get(k, {}), nk)
        fe, wr = last_sum(passes["fetch"].get(k, {}), nk), last_sum(passes["write"].get(k, {}), nk)
        ea = last_sum(passes["ea"].get(k, {}), nk)
        e = {"launches_summed": int(sq["_launches"]), "ms_total_under_counters": round(sq["_ns"] / 1e6, 3)}
        act, wavecyc = sq.get("SQ_ACTIVE_INST_VALU", 0.0), sq.get("SQ_WAVE_CYCLES", 0.0)
        lane = sq.get("SQ_THREAD_CYCLES_VALU", 0.0) / (act * 64.0) if act else None
        e["valu_wave_instructions"] = sq.get("SQ_INSTS_VALU")
        e["valu_lane_utilisation"] = round(lane, 4) if lane else None
        e["valu_share_of_wave_time"] = round(act / wavecyc, 4) if wavecyc else None
        e["wait_any_share_of_wave_time"] = round(sq.get("SQ_WAIT_ANY", 0.0) / wavecyc, 4) if wavecyc else None
        e["wait_inst_share_of_wave_time"] = round(sq.get("SQ_WAIT_INST_ANY", 0.0) / wavecyc, 4) if wavecyc else None
        busy = sq.get("SQ_BUSY_CYCLES", 0.0)
        e["mean_resident_waves_x_busy"] = round(wavecyc / busy, 3) if busy else None
        # time-based issue utilisation: wave-instructions x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz)
        secs = sq["_ns"] * 1e-9
        e["valu_issue_utilisation"] = round(sq.get("SQ_INSTS_VALU", 0.0) * 4.0 / (1024.0 * secs * 2.4e9), 4) if secs else None
        if lane and k in rays and rays[k] > 0:
            e["valu_lane_instructions_per_ray"] = round(sq.get("SQ_INSTS_VALU", 0.0) * 64.0 * lane / rays[k], 1)
            e["valu_wave_instructions_per_ray_x64"] = round(sq.get("SQ_INSTS_VALU", 0.0) * 64.0 / rays[k], 1)
        hit, miss = l2.get("TCC_HIT_sum", 0.0), l2.get("TCC_MISS_sum", 0.0)
        e["l2_hit_rate"] = round(hit / (hit + miss), 4) if hit + miss else None
        e["l2_requests"] = l2.get("TCC_REQ_sum")
        e["ta_busy_avr"] = round(l2.get("TA_BUSY_avr", 0.0) / max(l2["_launches"], 1), 2) if l2.get("_launches") else None
        # vector L1: one tag lookup per cycle and CU; cycles of the l2 pass = its own kernel time x the clock the sq2 pass saw
        # (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        clock = sq2.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / sq2["_ns"] * 1e9 if sq2.get("_ns") else 2.4e9
        l2cyc = l2["_ns"] * 1e-9 * clock if l2.get("_ns") else 0.0
        lookups = l2.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
        e["shader_clock_ghz"] = round(clock / 1e9, 3)
        e["l1_tag_lookups"] = lookups
        e["l1_tag_busy"] = round(lookups / 256.0 / l2cyc, 4) if l2cyc else None
        e["l1_hit_rate"] = round(1.0 - l2.get("TCP_TCC_READ_REQ_sum", 0.0) / lookups, 4) if lookups else None
        e["ta_busy"] = round(l2.get("TA_BUSY_avr", 0.0) / l2cyc, 4) if l2cyc else None
        if sq2.get("SQ_INSTS_VMEM_RD"):
            e["l1_tag_lookups_per_load_instruction"] = round(lookups / sq2["SQ_INSTS_VMEM_RD"], 2)
        hbm = (2.0 * fe.get("FETCH_SIZE", 0.0) + wr.get("WRITE_SIZE", 0.0)) * 1024.0
        e["hbm_bytes"] = round(hbm)
        e["hbm_read_bytes(2xFETCH_SIZE)"] = round(2.0 * fe.get("FETCH_SIZE", 0.0) * 1024.0)
        e["hbm_write_bytes"] = round(wr.get("WRITE_SIZE", 0.0) * 1024.0)
        if ea.get("TCC_EA0_RDREQ_DRAM_32B"):
            # the L2's own tally of what it asked the memory side for, in 32-byte units: no assumption about request sizes
            rd32, wr32 = ea["TCC_EA0_RDREQ_DRAM_32B"] * 32.0, ea.get("TCC_EA0_WRREQ_WRITE_DRAM_32B", 0.0) * 32.0
            e["dram_read_bytes(32B units)"] = round(rd32)
            e["dram_write_bytes(32B units)"] = round(wr32)
            e["read_requests_128B"] = ea.get("TCC_EA0_RDREQ_128B")
            e["read_requests_64B"] = ea.get("TCC_EA0_RDREQ_64B")
            e["fetch_size_factor"] = round(rd32 / (fe.get("FETCH_SIZE", 0.0) * 1024.0), 3) if fe.get("FETCH_SIZE") else None
            e["dram_bytes"] = round(rd32 + wr32)
            if k in rays and rays[k] > 0:
                e["dram_bytes_per_ray"] = round((rd32 + wr32) / rays[k], 2)
            if k in ("k_shade", "k_prims") and pb_total > 0:
                e["dram_bytes_per_path_bounce"] = round((rd32 + wr32) / pb_total, 2)
        if k in rays and rays[k] > 0:
            e["hbm_bytes_per_ray"] = round(hbm / rays[k], 2)
        if k in ("k_shade", "k_prims") and pb_total > 0:
            # the one-thread-per-slot stages handle every path-bounce once
            e["hbm_bytes_per_path_bounce"] = round(hbm / pb_total, 2)
            e["valu_wave_instructions_per_64_path_bounces"] = round(sq.get("SQ_INSTS_VALU", 0.0) * 64.0 / pb_total, 1)
        fsec = fe["_ns"] * 1e-9 if fe.get("_ns") else 0
        e["hbm_gbs_under_counters"] = round(hbm / fsec / 1e9, 1) if fsec else None
        for c in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"):
            if c in sq2:
                e[c.lower()] = sq2[c]
        res[k] = e
        lines.append("%-22s launches %3d  %9.2f ms  VALU issue %5.1f%%  lanes %5.1f%%  VALU share of wave time %5.1f%%  waiting %5.1f%%  L1 tag pipe %5.1f%%  L2 hit %5.1f%%  HBM %8.2f GB (%6.1f GB/s)  DRAM 32-byte tally %8.2f GB (x%.2f of FETCH_SIZE)" % (
            k, e["launches_summed"], e["ms_total_under_counters"], 100 * (e["valu_issue_utilisation"] or 0), 100 * (lane or 0),
            100 * (e["valu_share_of_wave_time"] or 0), 100 * (e["wait_any_share_of_wave_time"] or 0), 100 * (e["l1_tag_busy"] or 0), 100 * (e["l2_hit_rate"] or 0),
            hbm / 1e9, e["hbm_gbs_under_counters"] or 0, (e.get("dram_bytes") or 0) / 1e9, e.get("fetch_size_factor") or 0))
    out = os.path.join(ROOT, "profiles", "pmc_%s.json" % workload)
    json.dump(res, open(out, "w"), indent=1)
    txt = os.path.join(ROOT, "profiles", "%s_units_%s.txt" % (tag, workload))
    open(txt, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    print("wrote", out, "and", txt)


if __name__ == "__main__":
    main()
