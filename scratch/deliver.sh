set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01/bench -- python3 bench.py > gpurun_out/r01/bench_default.json 2> gpurun_out/r01/bench_default.err
tail -c 1500 gpurun_out/r01/bench_default.json
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01/pmc_fetch -- python3 benchmarks/k1_pmc.py > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01/pmc_write -- python3 benchmarks/k1_pmc.py > /dev/null 2>&1
python3 benchmarks/pmc_summary.py gpurun_out/r01/pmc_fetch gpurun_out/r01/pmc_write dwconv7_ln_fwd_strip gpurun_out/r01/k1_pmc.json
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01/pmc4_fetch -- python3 benchmarks/decode_pmc.py > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01/pmc4_write -- python3 benchmarks/decode_pmc.py > /dev/null 2>&1
python3 benchmarks/pmc_summary.py gpurun_out/r01/pmc4_fetch gpurun_out/r01/pmc4_write decode_attn_k gpurun_out/r01/k4_pmc.json
