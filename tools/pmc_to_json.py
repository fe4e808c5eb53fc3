#!/usr/bin/env python3
"""Fold the text summaries of tools/pmc_run.sh (gpurun_out/<tag>/pass*.txt) into profiles/r06_traffic.json, the file
bench.py reads the HBM-side traffic and the issue counters of a workload from.
usage: pmc_to_json.py <dir with pass*.txt> <workload key> <kernel substring[+substring...]> <steps per pass> [note]"""
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(d, key, kernel, steps, note=""):
    steps = float(steps)
    kernels = kernel.split("+")     # several kernels that make up one pass (e.g. every round of the lane kernel + the
    c = {}                          # chain kernel): counters and TOTAL times are summed over them
    ms = []
    for f in sorted(glob.glob(os.path.join(d, "pass*.txt"))):
        tot = 0.0
        for line in open(f):
            parts = [p.strip() for p in line.split("|")]
            if len(parts) == 4 and any(k in parts[0] for k in kernels):
                try:
                    if parts[1].isdigit():
                        tot += float(parts[2])      # calls | total_ms | avg_ms
                    else:
                        c[parts[1]] = c.get(parts[1], 0.0) + float(parts[2])
                except ValueError:
                    pass
        if tot:
            ms.append(tot)
    k_ms = sum(ms) / len(ms)        # per pass (bench.py --steps 1 --warmup 0)
    cyc = k_ms * 1e-3 * 2.4e9
    rec = {
        "kernel": kernel,
        "fetch_bytes": c["FETCH_SIZE"] * 1024.0, "write_bytes": c["WRITE_SIZE"] * 1024.0,
        "kernel_ms_under_pmc": round(k_ms, 3),
        "l2_hit": c.get("TCC_HIT_sum"), "l2_miss": c.get("TCC_MISS_sum"),
        "issue": {
            "salu_per_step": round(c["SQ_INSTS_SALU"] / steps, 1), "valu_per_step": round(c["SQ_INSTS_VALU"] / steps, 1),
            "vmem_rd_per_step": round(c["SQ_INSTS_VMEM_RD"] / steps, 3), "smem_per_step": round(c.get("SQ_INSTS_SMEM", 0) / steps, 3),
            "salu_util": round(c["SQ_INSTS_SALU"] / 256 / cyc, 3),                 # one scalar instruction per cycle per CU
            "valu_util": round(c["SQ_INSTS_VALU"] * 4 / 1024 / cyc, 3),            # 4 cycles per wave instruction, 4 SIMDs per CU
            "wave_cycles_waiting_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
            "clock_assumed_ghz": 2.4,
        },
        "sectors_per_step": round((c.get("TCC_MISS_sum") or 0.0) / steps, 3),
        # memory side of the L2 (round 5): read requests it sends to the fabric, those addressed to local DRAM (rocprofv3 on gfx950
        # has no Infinity-Cache hit counter: both count a request whether the MALL or HBM serves it), and the average time one is
        # outstanding (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ, L2 clocks) -- the latency the memory side answers with under this load
        "ea_read_requests": c.get("TCC_EA0_RDREQ_sum"), "ea_read_requests_dram": c.get("TCC_EA0_RDREQ_DRAM_sum"),
        "ea_read_latency_cycles": (round(c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"], 1)
                                   if c.get("TCC_EA0_RDREQ_sum") and c.get("TCC_EA0_RDREQ_LEVEL_sum") else None),
        "tcp_pending_stall_frac": (round(c["TCP_PENDING_STALL_CYCLES_sum"] / 256 / cyc, 3) if c.get("TCP_PENDING_STALL_CYCLES_sum") else None),
        "note": note or ("FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over the same launch (tools/pmc_run.sh); "
                         "scattered <= 64-byte accesses: FETCH_SIZE is exact at one 64-byte sector per access "
                         "(profiles/r02_fetch_calibration.txt)"),
    }
    path = os.path.join(REPO, "profiles", "r06_traffic.json")
    try:
        doc = json.load(open(path))
    except (OSError, ValueError):
        doc = {"_comment": "per-launch HBM-side traffic and issue counters of the dominant kernel, from rocprofv3 --pmc passes "
                           "(tools/pmc_run.sh + tools/pmc_to_json.py); bench.py looks its workload up here", "workloads": {}}
    doc["workloads"][key] = rec
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
