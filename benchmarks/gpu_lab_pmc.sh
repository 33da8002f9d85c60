#!/bin/bash
# LDS conflict counters of a lab GEMM run: gpu_lab_pmc.sh <binary> <args...>
export TMPDIR=/tmp
O=gpurun_out/r2/labpmc; rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $O -o p -- "$@" > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r2/labpmc/**/p_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    if 'gemm' in k:
        print(k, {c: round(x) for c, x in v.items()}, 'conflict/active = %.3f' % (v['SQ_LDS_BANK_CONFLICT'] / max(1, v['SQ_LDS_IDX_ACTIVE'])))
PY
