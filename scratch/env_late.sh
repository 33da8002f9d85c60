cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_env -- python3 scratch/env_late.py 2>&1 | grep "env"
python3 - <<EOF
import csv,glob,collections
f=sorted(glob.glob('gpurun_out/prof_env/*/*_kernel_trace.csv'))[-1]
rows=[(int(r["Start_Timestamp"]), r["Kernel_Name"][:30], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])) for r in csv.DictReader(open(f)) if 'Cijk' in r["Kernel_Name"]]
rows.sort()
print([g for _,_,g in rows])
EOF
