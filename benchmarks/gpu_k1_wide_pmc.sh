#!/bin/bash
# counters of the K1 / K1b kernels, bf16 and f32: where do the waves' cycles go?  Default shape: wide (C = 1600, B = 2048,
# L = 31); SHAPE="8192 64 250" TAG=k1s gives the strip kernels at the headline roofline shape
set -o pipefail
export TMPDIR=/tmp
SHAPE=${SHAPE:-"2048 1600 31"}
TAG=${TAG:-k1w}
O=gpurun_out/r3/$TAG
mkdir -p $O
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA TCC_HIT_sum TCC_MISS_sum"
for dt in bf16 f32; do for dir in fwd bwd; do
  tag=${dt}_${dir}
  mkdir -p $O/$tag
  bash benchmarks/run_pmc.sh $O/$tag benchmarks/k1_pmc.py $dt $dir $SHAPE -- PASS "$P1" PASS "$P2" || exit $?
  python3 benchmarks/attn_pmc_summary.py $O/${TAG}_$tag.json --kernels=dwconv7_ln_fwd_wide,dwconv7_ln_bwd_wide,dwconv7_ln_fwd_strip,dwconv7_ln_bwd_strip,dwconv7_ln_bwd_reduce $O/$tag/pass1 $O/$tag/pass2 > /dev/null 2>&1
  find $O/$tag -name "*.db" -delete; rm -rf $O/$tag
done; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/${TAG}_*.json")):
    d = json.load(open(f))
    for k, e in d.items():
        keep = {kk: e[kk] for kk in e if kk.endswith("_over_WAVE_CYCLES") or kk in ("vgpr", "workgroup", "grid", "kernel_cycles", "avg_resident_waves_per_simd", "waves_per_simd_allowed_by_registers", "l2_hit_rate")}
        c = e["counters_per_launch"]
        keep["valu_per_wave"] = round(c.get("SQ_INSTS_VALU", 0) / max(1, c.get("SQ_WAVES", 1)))
        keep["lds_per_wave"] = round(c.get("SQ_INSTS_LDS", 0) / max(1, c.get("SQ_WAVES", 1)))
        keep["vmem_rd_per_wave"] = round(c.get("SQ_INSTS_VMEM_RD", 0) / max(1, c.get("SQ_WAVES", 1)))
        keep["salu_per_wave"] = round(c.get("SQ_INSTS_SALU", 0) / max(1, c.get("SQ_WAVES", 1)))
        print(f.split("/")[-1], k, json.dumps(keep))
PY
