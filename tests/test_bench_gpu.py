"""The DEFAULT path of bench.py -- the one the driver runs -- end to end on the GPU, every extra leg at its smallest step
count (--quick) and a one-step, batch-2 CPU baseline, so the whole file costs well under two minutes.  Round 2's line was
lost to an exception after the last leg; this test fails on any leg that raises."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_default_bench_path_prints_a_complete_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--quick", "--cpu-steps", "1",
                        "--cpu-batch", "2"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert "failed_legs" not in line, (line["failed_legs"], r.stderr[-4000:])
    assert line["metric"] == "train spectra/sec" and line["n_gpus"] == 1 and line["dtype"] == "bf16" and line["value"] > 0
    assert "workload" in line["config"] and "optimizer_updates" in line["config"]
    ro = line["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and 0 < ro["frac"] < 1 and ro["peak"] == 8000.0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    for leg in ("greedy", "roofline_mfma", "roofline_k2", "roofline_attn", "roofline_attn_d512", "roofline_decode",
                "roofline_k1_all_shapes", "train_batch256_bf16", "train_batch32_fp32", "train_own_gemms_bf16",
                "train_real_config_mup_bf16", "test_config_forward_greedy", "hip_graph_step", "rccl_world1"):
        assert leg in line, leg
