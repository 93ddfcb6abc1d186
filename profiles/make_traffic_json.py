#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 --pmc passes of `bench.py --light` (profiles/run_profile.sh <tag> + profiles/run_mix.sh <tag>_mix): what the
bench line quotes as `roofline.traffic` (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) and `roofline.valu` (vector
instructions per cell by class, issue time at the clock the chip sustains under this arithmetic).
Usage: python profiles/make_traffic_json.py gpurun_out/prof_r4 gpurun_out/prof_r4_mix profiles/r4 [refs k preset]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def counters(root, kernel="lfd_dense_kernel"):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Kernel_Name"].startswith(kernel):
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    prof, mix, where = sys.argv[1], sys.argv[2], sys.argv[3]
    refs, k, preset = (int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]) if len(sys.argv) > 6 else (64, 3, "fast")
    c = counters(prof)
    m = counters(mix)
    grid = {"turbo": 320, "fast": 512, "base": 640, "high": 960, "precise": 1280}[preset]
    cells = refs * grid * grid
    per_cell = lambda name: m.get(name, 0.0) * 64.0 / cells
    f64 = sum(per_cell(n) for n in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    f32 = sum(per_cell(n) for n in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"))
    trans = per_cell("SQ_INSTS_VALU_TRANS_F64") + per_cell("SQ_INSTS_VALU_TRANS_F32")
    total = per_cell("SQ_INSTS_VALU")
    # real cycles per wave-instruction under load and the clock it runs at: profiles/microbench/clock_under_load.txt (profiles/microbench/clock_under_load.hip)
    cyc_f64, cyc_other, clock_ghz = 4.4, 2.3, 1.9
    cycles_per_cell = f64 * cyc_f64 + (total - f64) * cyc_other + trans * 8.0       # (reciprocals / roots are quarter rate: ~8 more cycles each)
    issue_ms = cells / 64.0 / 1024.0 * cycles_per_cell / (clock_ghz * 1e9) * 1e3
    out = {
        "comment": "HBM bytes per lfd_dense_kernel launch and its vector-instruction mix from rocprofv3 --pmc passes of `bench.py --light` (profiles/run_profile.sh, "
                   "profiles/run_mix.sh; separate FETCH_SIZE / WRITE_SIZE passes). FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE as reported (KB).",
        "workload": {"refs": refs, "k": k, "preset": preset},
        "fetch_size_kb": c.get("FETCH_SIZE"), "write_size_kb": c.get("WRITE_SIZE"),
        "traffic_bytes": (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0,
        "source": f"{where}/summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --light`; FETCH_SIZE x 2 per the gfx950 correction of MI355X_MICROARCH.md)",
        "valu": {
            "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU", m.get("SQ_INSTS_VALU")), "SQ_ACTIVE_INST_VALU": c.get("SQ_ACTIVE_INST_VALU"), "SQ_BUSY_CYCLES": c.get("SQ_BUSY_CYCLES"),
            "valu_busy_frac": (c["SQ_ACTIVE_INST_VALU"] * 4.0 / (c["SQ_BUSY_CYCLES"] * 32.0)) if c.get("SQ_BUSY_CYCLES") else None,
            "instructions_per_cell": total, "f64_per_cell": f64, "f32_per_cell": f32, "conversions_per_cell": per_cell("SQ_INSTS_VALU_CVT"),
            "integer_per_cell": per_cell("SQ_INSTS_VALU_INT32") + per_cell("SQ_INSTS_VALU_INT64"), "transcendental_per_cell": trans,
            "cycles_per_wave_instruction": {"f64": cyc_f64, "other": cyc_other}, "sustained_clock_GHz": clock_ghz,
            "issue_cycles_per_cell": cycles_per_cell, "issue_ms": issue_ms,
            "source": f"{where}/instruction_mix.txt (SQ_INSTS_VALU_* per launch x 64 lanes / cells); cycles per wave-instruction and the sustained clock from "
                      "profiles/microbench/clock_under_load.txt (profiles/microbench/clock_under_load.hip)",
            "note": "SQ_ACTIVE_INST_VALU counts 4-cycle units summed over the chip; SQ_BUSY_CYCLES is summed over the 32 shader engines (32 SIMDs each)"},
    }
    print(json.dumps(out, indent=2))


if __name__ == "__main__":
    main()
