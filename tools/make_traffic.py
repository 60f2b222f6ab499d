#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of `bench.py --profile-pass` (tools/pmc.sh fetch | write | insts | cycles) into
profiles/pmc_traffic.json: per kernel, per launch -- HBM-side bytes and VALU instructions.

HBM bytes per launch = (FETCH_SIZE * 2 + WRITE_SIZE) * 1024 / launches: FETCH_SIZE / WRITE_SIZE are reported in KiB, and on gfx950
FETCH_SIZE reads half of the fetched bytes (MI355X_MICROARCH.md, HBM section; the factor was calibrated on wide streaming reads -- for
the divergent 16 B/lane gathers of the traversal kernels it is an upper-bound style correction, stated as such in DESIGN.md).
VALU instructions per launch = SQ_INSTS_VALU / launches (wave-level instructions: one per wave and instruction)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


PREFIX = ""


def read(tag, counter):
    tag = PREFIX + tag
    path = os.path.join(ROOT, "gpurun_out", "pmc_" + tag, "summary_%s.csv" % tag)
    out = {}
    if not os.path.exists(path):
        return out
    for row in csv.DictReader(open(path)):
        if counter in row:
            out[row["kernel"]] = (int(row["calls"]), float(row[counter]))
    return out


def library_build_id():
    """rptr_hip_build_id() of the library the passes ran on (the one in the tree, or RPTR_HIP_LIB): bench.py flags counters of another build"""
    import ctypes
    try:
        lib = ctypes.CDLL(os.environ.get("RPTR_HIP_LIB") or os.path.join(ROOT, "realtimepathtracingresearchframework_amd", "librptr_hip.so"))
        lib.rptr_hip_build_id.restype = ctypes.c_char_p
        return lib.rptr_hip_build_id().decode()
    except Exception:
        return None


def main():
    """make_traffic.py <build tag> [<workload key> [<bench args>]]: the passes gpurun_out/pmc_<key>_{fetch,write,insts,cycles,tcc} -> the
    entry `workloads[<key>]` of profiles/pmc_traffic.json (the other workloads' entries are kept)"""
    global PREFIX
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    key = sys.argv[2] if len(sys.argv) > 2 else "c2"
    bench_args = sys.argv[3] if len(sys.argv) > 3 else ""
    PREFIX = key + "_"
    fetch, write = read("fetch", "FETCH_SIZE"), read("write", "WRITE_SIZE")
    valu, salu, waves = read("insts", "SQ_INSTS_VALU"), read("insts", "SQ_INSTS_SALU"), read("insts", "SQ_WAVES")
    busy, gui = read("cycles", "SQ_BUSY_CYCLES"), read("cycles", "GRBM_GUI_ACTIVE")
    cyc = {c: read("cycles", c) for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")}
    tcc_hit, tcc_miss = read("tcc", "TCC_HIT_sum"), read("tcc", "TCC_MISS_sum")
    ta_busy = read("tcc", "TA_TA_BUSY_sum")
    doc = {"bench_args": bench_args,
           "source": "rocprofv3 --kernel-trace --pmc <counters> (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_* | SQ_*_CYCLES) over "
                     "bench.py --profile-pass --steps 3 --warmup 1 <bench_args> (frames one at a time, RPTR_TAIL_BOUNCE=2); tools/pmc.sh + tools/make_traffic.py",
           "build": tag, "library_build_id": library_build_id(),
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads 1/2, MI355X_MICROARCH.md)", "kernels": {}}
    for k in fetch:
        calls, f = fetch[k]
        w = write.get(k, (calls, 0.0))[1]
        e = {"launches": calls, "fetch_kib_raw": f, "write_kib_raw": w, "hbm_bytes_per_launch": (2 * f + w) * 1024 / max(calls, 1)}
        if k in valu:
            e["valu_insts_per_launch"] = valu[k][1] / max(valu[k][0], 1)
            e["salu_insts_per_launch"] = salu[k][1] / max(salu[k][0], 1)
            e["waves_per_launch"] = waves[k][1] / max(waves[k][0], 1)
        if k in gui:
            e["gui_active_cycles_per_launch"] = gui[k][1] / max(gui[k][0], 1)
        if k in cyc["SQ_WAVE_CYCLES"] and cyc["SQ_WAVE_CYCLES"][k][1] > 0:
            wc = cyc["SQ_WAVE_CYCLES"][k][1]
            # fractions of the wave-cycles (quad-cycles summed over resident waves): parked on s_waitcnt / barrier, issue-stalled, issuing
            for name, c in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_any_frac", "SQ_WAIT_INST_ANY"), ("active_inst_any_frac", "SQ_ACTIVE_INST_ANY"),
                            ("active_inst_valu_frac", "SQ_ACTIVE_INST_VALU")):
                if k in cyc[c]:
                    e[name] = round(cyc[c][k][1] / wc, 4)
            e["wave_quad_cycles_per_launch"] = wc / max(cyc["SQ_WAVE_CYCLES"][k][0], 1)
            if k in busy:
                e["sq_busy_cycles_per_launch"] = busy[k][1] / max(busy[k][0], 1)
        if k in tcc_hit and k in tcc_miss and tcc_hit[k][1] + tcc_miss[k][1] > 0:
            e["tcc_hit_rate"] = round(tcc_hit[k][1] / (tcc_hit[k][1] + tcc_miss[k][1]), 4)
        if k in ta_busy and k in gui and gui[k][1] > 0 and ta_busy[k][0] == gui[k][0]:
            # busy share of a CU's texture-address unit (the vector-memory path) while the kernel runs ALONE: TA_TA_BUSY_sum is the sum over
            # the chip's 256 units, GRBM_GUI_ACTIVE (as rocprofv3 reports it on this part) the sum over its 8 XCDs
            e["ta_busy_frac"] = round((ta_busy[k][1] / 256.0) / (gui[k][1] / 8.0), 4)
        doc["kernels"][k] = e
    # one frame = one rp_k_resolve launch: all VALU instructions of the pass / frames (where the tail kernel takes over does not change
    # the work of a frame, so this holds for the pipelined run too, whatever bounce its tail starts at)
    frames = max((v["launches"] for k, v in doc["kernels"].items() if "rp_k_resolve" in k), default=0)
    # (the one-time device build of a static tree -- rp_k_build_tris, rp_k_ploc_*, and the rp_k_lbvh_* steps it borrows -- is set_scene's work,
    # not a frame's)
    static_build = any("rp_k_ploc_" in k for k in doc["kernels"])
    per_frame = {k: v for k, v in doc["kernels"].items()
                 if not ("rp_k_ploc_" in k or "rp_k_build_tris" in k or (static_build and "rp_k_lbvh_" in k))}
    if frames:
        doc["frames"] = frames
        doc["valu_insts_per_frame"] = sum(v.get("valu_insts_per_launch", 0.0) * v["launches"] for v in per_frame.values()) / frames
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        full = json.load(open(path))
    except Exception:
        full = {}
    if "workloads" not in full:   # round 2's file held the default workload only
        full = {"workloads": {}}
    full["workloads"][key] = doc
    json.dump(full, open(path, "w"), indent=1)
    print("frames %d, VALU instructions per frame %.1f M" % (doc.get("frames", 0), doc.get("valu_insts_per_frame", 0) / 1e6))
    for k, v in doc["kernels"].items():
        print("%-48s %8.1f MB HBM  %8.1f M VALU insts per launch" % (k[:48], v["hbm_bytes_per_launch"] / 1e6, v.get("valu_insts_per_launch", 0) / 1e6))


if __name__ == "__main__":
    main()
