#!/usr/bin/env python3
"""attn_pmc_summary.py <out.json> [--kernels=a,b] [--lds=a:bytes,b:bytes] <pmc_dir> [<pmc_dir> ...]: per-launch averages of every counter rocprofv3 collected for
the attention kernels (one --pmc pass per directory), plus the ratios the roofline discussion needs.  SQ_* cycle
counters are in quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants); SQ_VALU_MFMA_BUSY_CYCLES in cycles."""
import csv
import glob
import json
import sys
from collections import defaultdict

KERNELS = ("prefix_attn_fwd", "prefix_attn_bwd_dq", "prefix_attn_bwd_dkv", "attn_delta")
args = sys.argv[2:]
if args and args[0].startswith("--kernels="):          # other kernels: --kernels=substr1,substr2
    KERNELS = tuple(args[0].split("=", 1)[1].split(","))
    args = args[1:]
LDS = {}                 # dynamic LDS per workgroup as the launch code computes it (the CSV's LDS_Block_Size holds the static part)
if args and args[0].startswith("--lds="):
    LDS = {kv.split(":")[0]: int(kv.split(":")[1]) for kv in args[0].split("=", 1)[1].split(",")}
    args = args[1:]
N_CU, SIMD_PER_CU, LDS_PER_CU = 256, 4, 160 * 1024


def vgpr_waves_per_simd(regs):
    """MI355X_MICROARCH.md, register files: allocation granule 8, waves per SIMD = min(8, 512 // alloc)."""
    alloc = max(8, -(-int(regs) // 8) * 8)
    return min(8, 512 // alloc)


acc = defaultdict(lambda: defaultdict(list))
meta = {}
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            key = next((k for k in KERNELS if k in kn), None)
            if key is None:
                continue
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta.setdefault(key, {"kernel_name": kn.split("(")[0], "grid": r.get("Grid_Size"), "workgroup": r.get("Workgroup_Size"),
                                  "vgpr": r.get("VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"),
                                  "sgpr": r.get("SGPR_Count"), "lds_bytes": r.get("LDS_Block_Size")})
out = {}
for key, ctrs in acc.items():
    avg = {c: sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) for c, v in ctrs.items()}      # drop the first third (warm-up)
    e = dict(meta[key])
    e["counters_per_launch"] = {c: round(x, 1) for c, x in sorted(avg.items())}
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS"):
            if c in avg:
                e[c + "_over_WAVE_CYCLES"] = round(avg[c] / wc, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CU_CYCLES" in avg and avg["SQ_BUSY_CU_CYCLES"]:
        e["mfma_busy_over_cu_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / avg["SQ_BUSY_CU_CYCLES"], 4)
    if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_over_idx_active"] = round(avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"], 4)
    if "SQ_INSTS_VALU" in avg and avg.get("SQ_INSTS_MFMA"):
        e["valu_insts_per_mfma"] = round((avg["SQ_INSTS_VALU"] - avg["SQ_INSTS_MFMA"]) / avg["SQ_INSTS_MFMA"], 2)
    # residency: what the registers and the LDS allow, and what the launch achieved on average
    try:
        regs = int(e["vgpr"] or 0) + int(e["accum_vgpr"] or 0)
        wg_waves = max(1, int(e["workgroup"]) // 64)
        by_vgpr = vgpr_waves_per_simd(regs)
        e["regs_per_lane"] = regs
        e["waves_per_simd_allowed_by_registers"] = by_vgpr
        lds = LDS.get(key, int(e["lds_bytes"] or 0))
        e["lds_bytes"] = lds
        wg_by_lds = LDS_PER_CU // lds if lds else 32
        wg_by_regs = by_vgpr * SIMD_PER_CU // wg_waves if wg_waves <= by_vgpr * SIMD_PER_CU else 0
        e["workgroups_per_cu_allowed_by_lds"] = wg_by_lds
        e["workgroups_per_cu_allowed"] = min(wg_by_lds, wg_by_regs, 32 // wg_waves)
        e["waves_per_simd_allowed"] = round(e["workgroups_per_cu_allowed"] * wg_waves / SIMD_PER_CU, 2)
        e["workgroups"] = int(e["grid"]) // int(e["workgroup"])
        e["workgroup_rounds"] = round(e["workgroups"] / (N_CU * max(1, e["workgroups_per_cu_allowed"])), 2)
        if wc and avg.get("GRBM_GUI_ACTIVE"):
            dur_cycles = avg["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
            e["kernel_cycles"] = round(dur_cycles)
            e["avg_resident_waves_per_simd"] = round(wc * 4.0 / (dur_cycles * N_CU * SIMD_PER_CU), 3)   # WAVE_CYCLES: quad-cycles
        if avg.get("SQ_VALU_MFMA_BUSY_CYCLES") and avg.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_over_all_simd_time"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8.0 * N_CU * SIMD_PER_CU), 4)
        if avg.get("TCC_HIT_sum") is not None and avg.get("TCC_MISS_sum") is not None and avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"]), 4)
    except (TypeError, ValueError, ZeroDivisionError) as ex:
        e["residency_error"] = str(ex)
    out[key] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
