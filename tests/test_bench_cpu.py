"""bench.py's launch contract on CPU (no GPU, backend gloo, --dry-run): `--gpus N` alone starts N ranks itself, the
driver's torch.distributed.run form works too, and a world size that disagrees with --gpus is an error -- the line can
never say n_gpus: 1 for an N-GPU request again."""
import json
import os
import socket
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _last_json(out: str):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_flag_alone_spawns_the_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and len(line["per_rank_ms_per_step"]) == 2
    assert line["allreduce_calls"] >= 3 and line["replicas_equal"]


def test_driver_form_under_torch_distributed_run():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 2


def test_world_size_must_match_gpus():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--backend", "gloo", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_json_line_assembly_without_gpu():
    """The JSON line is assembled from plain values (round 2 lost its line to a read of a deleted object): the
    assembly, with a failed leg recorded, must serialise and carry the contract's keys."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(batch=32, steps=20, warmup=5, dtype="bf16", config="local_config")
    extras = {}
    bench.run_leg(extras, "roofline", lambda: {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1,
                                                 "traffic": None})
    bench.run_leg(extras, "broken", lambda: 1 / 0)
    out = bench.assemble_line(a, bench.CONFIGS["local_config"], 1, 0.5, 6.9, {"ranks": 1}, "all inside opt.step()", extras)
    line = json.loads(json.dumps(out))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["value"] == 1280.0 and line["ms_per_step"] == 25.0 and "workload" in line["config"]
    assert line["failed_legs"] == ["broken"] and "ZeroDivisionError" in line["broken"]["error"]
