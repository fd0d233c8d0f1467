"""bench.py as the driver launches it: a child process (never this one: the GPU stays with the test runner), the JSON contract checked.
The N > 1 form -- `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` -- is exercised with N = 1: same launcher, same
RANK / WORLD_SIZE / MASTER_* environment, process group over RCCL, barriers and the MAX / SUM reductions of the timing."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "roofline_seed_lookup", "host")


def _check(line, steps, warmup, cpu):
    d = json.loads(line)
    for k in KEYS:
        assert k in d, k
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["distinct_blocks_in_timed_region"] == steps          # no read mapped twice in the timed region
    for r in (d["roofline"], d["roofline_seed_lookup"]):
        assert r["bound"] in ("hbm", "valu") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["roofline_seed_lookup"]["bound"] == "hbm" and d["roofline_seed_lookup"]["unit"] == "GB/s"
    assert d["host"]["cpu_us_per_read"] > 0
    if cpu:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["parity"]["mismatching_reads"] == 0 and c["parity"]["reads"] > 0
    return d


def _run(cmd, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line
    return lines[0]


def test_bench_json_contract_single_process():
    line = _run([sys.executable, "bench.py", "--workload", "ecoli", "--reads", "4096", "--steps", "2", "--warmup", "1", "--cpu-seconds", "3"])
    _check(line, 2, 1, True)


def test_bench_under_the_distributed_launcher():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                 "bench.py", "--gpus", "1", "--workload", "ecoli", "--reads", "2048", "--steps", "1", "--warmup", "1", "--no-cpu", "--no-pcie"])
    _check(line, 1, 1, False)
