#!/usr/bin/env python3
"""Turns the raw rocprofv3 outputs of one workload (kernel stats csv, FETCH_SIZE / WRITE_SIZE / SQ counter csvs) into
profiles/<round>_<tag>_pmc_summary.json, the file bench.py reads back for roofline.traffic.

usage: pmc_summary.py <dst_dir> <round> <tag> <workload_key> <kernel_regex> <bench_json_under_rocprof>

HBM-side bytes per step = sum over the workload's kernels of (2 x FETCH_SIZE + WRITE_SIZE) per dispatch.  The
factor 2: on gfx950 FETCH_SIZE tallies every 128-byte line an L2 miss brings in as 64 bytes -- for streaming reads of
any width, LDS-DMA and scattered 1..12-byte pieces alike (profiles/r02_fetch_calibration.json, measured with
tools/svoh_microbench on known byte counts)."""
import collections
import csv
import json
import os
import re
import sys


def per_kernel(path, counter_names=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if counter_names and r["Counter_Name"] not in counter_names:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    dst, rnd, tag, key, regex, bench_json = sys.argv[1:7]
    rx = re.compile(regex)
    out = {"workload_key": key, "kernel_regex": regex,
           "units": "FETCH_SIZE / WRITE_SIZE in KB (1024 B) per dispatch as rocprofv3 reports them; traffic_bytes_per_step = "
                    "sum over kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (calibration: r02_fetch_calibration.json)"}
    try:
        lines = [l for l in open(bench_json).read().splitlines() if l.startswith("{")]
        b = json.loads(lines[-1])
        out["bench_under_rocprof"] = {"kernel_ms": b.get("kernel_ms"), "value": b.get("value"), "unit": b.get("unit")}
    except Exception as e:
        out["bench_under_rocprof"] = {"error": str(e)}
    kernels = {}
    traffic = 0.0
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(dst, "%s_%s_pmc_%s.csv" % (rnd, tag, ctr))
        if not os.path.exists(f):
            continue
        for name, d in per_kernel(f, (ctr,)).items():
            if not rx.search(name):
                continue
            vals = sorted(d[ctr])
            top = [v for v in vals if v >= 0.5 * vals[-1]] if vals[-1] > 0 else vals   # the workload's own launches
            kernels.setdefault(name, {})[ctr] = {"dispatches": len(top), "mean_per_dispatch_KB_as_reported": sum(top) / len(top)}
    for name, d in kernels.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["traffic_bytes"] = (2.0 * d["FETCH_SIZE"]["mean_per_dispatch_KB_as_reported"] +
                                  d["WRITE_SIZE"]["mean_per_dispatch_KB_as_reported"]) * 1024.0
            traffic += d["traffic_bytes"]
    out["kernels"] = kernels
    out["traffic_bytes_per_step"] = traffic if kernels else None
    # kernel times of the same command (rocprofv3 --kernel-trace --stats)
    st = os.path.join(dst, "%s_%s_kernel_stats_svoh.csv" % (rnd, tag))
    if os.path.exists(st):
        rows = [r for r in csv.DictReader(open(st)) if rx.search(r["Name"].split("(")[0])]
        out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")} for r in rows]
        # the kernels of ONE STEP of the workload: those that move at least 1 % of the largest kernel's bytes per dispatch.  bench.py
        # also times side legs under the same regex (--workload pose: fifty single-bundle dispatches of pose_optimize_kernel<256,0>
        # for the latency figure); their average duration is not part of a step (it was added until round 5)
        big = max((d.get("traffic_bytes", 0.0) for d in kernels.values()), default=0.0)
        in_step = {n for n, d in kernels.items() if d.get("traffic_bytes", 0.0) >= 0.01 * big} if big > 0 else None
        step_rows = [r for r in rows if in_step is None or r["Name"].split("(")[0] in in_step]
        out["kernel_ms_per_step_rocprof"] = sum(float(r["AverageNs"]) for r in step_rows) * 1e-6
        out["kernels_not_part_of_a_step"] = [r["Name"].split("(")[0] for r in rows if r not in step_rows]
        if in_step is not None:
            out["traffic_bytes_per_step"] = sum(d["traffic_bytes"] for n, d in kernels.items() if n in in_step)
    # compute side: SQ counters of the dominant kernel (the one with the largest wave-cycle count)
    sq = {}
    for f in sorted(os.listdir(dst)):
        if not f.startswith("%s_%s_pmc_SQ" % (rnd, tag)) or not f.endswith(".csv"):
            continue
        for name, d in per_kernel(os.path.join(dst, f)).items():
            if not rx.search(name):
                continue
            for c, vals in d.items():
                vals = sorted(vals)
                top = [v for v in vals if v >= 0.5 * vals[-1]] if vals[-1] > 0 else vals
                sq.setdefault(name, {})[c] = sum(top) / len(top)
    if sq:
        dom = max(sq, key=lambda n: sq[n].get("SQ_WAVE_CYCLES", sq[n].get("SQ_INSTS_VALU", 0.0)))
        c = sq[dom]
        d = {"kernel": dom, "counters_mean_per_dispatch": c}
        if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
            # SQ_BUSY_CYCLES sums 32 shader engines' busy cycles; SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles
            kernel_cycles = c["SQ_BUSY_CYCLES"] / 32.0
            d["kernel_cycles"] = kernel_cycles
            d["valu_busy_fraction_of_simd_time"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * kernel_cycles)
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in c:
                    d[k + "_share_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
        if "SQ_INSTS_VALU_FMA_F64" in c:
            # wave-instructions x 64 lanes (an upper bound where lanes are masked off); FMA = 2 flop
            flop = 64.0 * (c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + 2.0 * c["SQ_INSTS_VALU_FMA_F64"])
            d["fp64_flop_per_dispatch_upper_bound"] = flop
            if "kernel_cycles" in d and out.get("kernel_ms_per_step_rocprof"):
                dom_ms = None
                for r in out.get("kernel_stats", []):
                    if r["Name"].split("(")[0] == dom:
                        dom_ms = float(r["AverageNs"]) * 1e-6
                if dom_ms:
                    d["fp64_tflops_upper_bound"] = flop / (dom_ms * 1e-3) / 1e12
                    d["fp64_fraction_of_measured_peak_63p8"] = d["fp64_tflops_upper_bound"] / 63.8
            if "SQ_INSTS_VALU" in c and c["SQ_INSTS_VALU"]:
                d["fp64_arith_share_of_valu_instructions"] = (c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                                                              c["SQ_INSTS_VALU_FMA_F64"]) / c["SQ_INSTS_VALU"]
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        out["compute_side"] = d
    json.dump(out, open(os.path.join(dst, "%s_%s_pmc_summary.json" % (rnd, tag)), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("workload_key", "traffic_bytes_per_step", "kernel_ms_per_step_rocprof") if k in out}))


if __name__ == "__main__":
    main()
