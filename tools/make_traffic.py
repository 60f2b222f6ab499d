#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE PMC passes (tools/pmc.sh fetch|write) into profiles/pmc_traffic.json.

HBM bytes per launch = (FETCH_SIZE * 2 + WRITE_SIZE) * 1024 / launches: FETCH_SIZE/WRITE_SIZE are reported
in KiB, and on gfx950 FETCH_SIZE reads half of the fetched bytes (MI355X_MICROARCH.md, HBM section; the
factor was calibrated on wide streaming reads -- for the divergent 16 B/lane gathers of the traversal
kernels it is an upper-bound style correction, stated as such in DESIGN.md)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path, counter):
    out = {}
    for row in csv.DictReader(open(path)):
        out[row["kernel"]] = (int(row["calls"]), float(row[counter]))
    return out


def main():
    fetch = read(os.path.join(ROOT, "gpurun_out", "pmc_fetch", "summary_fetch.csv"), "FETCH_SIZE")
    write = read(os.path.join(ROOT, "gpurun_out", "pmc_write", "summary_write.csv"), "WRITE_SIZE")
    doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --profile-pass --steps 3 --warmup 1 (frames one at a time, RPTR_TAIL_BOUNCE=2)",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads 1/2, MI355X_MICROARCH.md)", "kernels": {}}
    for k in fetch:
        calls, f = fetch[k]
        w = write.get(k, (calls, 0.0))[1]
        doc["kernels"][k] = {"launches": calls, "fetch_kib_raw": f, "write_kib_raw": w,
                             "hbm_bytes_per_launch": (2 * f + w) * 1024 / max(calls, 1)}
    ext = [v for k, v in doc["kernels"].items() if "rp_k_extend<false" in k]  # <false, true, ALPHA> (first bounce) + <false, false, ALPHA>
    if ext:
        doc["rp_k_extend_hbm_bytes_per_launch"] = (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ext) /
                                                   max(sum(v["launches"] for v in ext), 1))
    json.dump(doc, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in doc["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
