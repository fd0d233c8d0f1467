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


def _check(line, steps, warmup, cpu, n_gpus=1):
    d = json.loads(line)
    for k in KEYS:
        assert k in d, k
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert "PCIe-inclusive" in d["value_basis"] and (d["resident_mbases_per_s"] is None or d["resident_mbases_per_s"] > 0)
    assert d["roofline"]["kernel"] == max(d["kernel_ms_per_step_all"], key=d["kernel_ms_per_step_all"].get)   # the rule: largest summed duration, no exclusions
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["distinct_blocks_in_timed_region"] == steps          # no read mapped twice in the timed region
    for r in (d["roofline"], d["roofline_seed_lookup"], d["roofline_seed_lookup"]["four_term_B_seed"]):
        assert r["bound"] in ("hbm", "valu") and r["peak"] > 0
        assert r["achieved"] is None or abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["roofline_seed_lookup"]["bound"] == "hbm" and d["roofline_seed_lookup"]["unit"] == "GB/s"
    assert d["host"]["cpu_us_per_read"] > 0
    if cpu:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["parity"]["mismatching_reads"] == 0 and c["parity"]["reads"] > 0
        # the real minimap2 / mappy is looked for on every run with a CPU leg: null when the box has neither, else timed + diffed
        assert "reference_on_box" in d and (d["reference_on_box"] is None or "error" in d["reference_on_box"] or d["reference_on_box"]["parity"]["mismatching_reads"] == 0)
    return d


def _run(cmd, timeout=900, extra_env=None, may_fail=False):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if may_fail and p.returncode != 0:
        return None, p.stderr[-3000:]
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


def test_two_ranks_share_the_one_gpu():
    """the N > 1 code path with more than one rank on hardware (the boxes have one GPU: both ranks use GPU 0, `--oversubscribe`): process
    group of two, the read set cut into two shards balanced by bases, every rank maps its own shard against the index replica of its device,
    MAX over ranks of the time and SUM of the bases on rank 0's line.  RCCL first; if it refuses two ranks on one device, the same run over
    gloo (the collective is only the timing barrier / reduction, never on the data path)."""
    out = None
    for backend in ("nccl", "gloo"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
               "bench.py", "--gpus", "2", "--oversubscribe", "--workload", "ecoli", "--reads", "2048", "--steps", "2", "--warmup", "1", "--no-cpu", "--streams", "2", "--depth", "1"]
        out = _run(cmd, extra_env={"MM355_BENCH_DIST_BACKEND": backend, "MM355_HOST_THREADS": "4"}, may_fail=backend == "nccl")
        if isinstance(out, str):
            break
        print("RCCL with two ranks on one device failed, falling back to gloo:", out[1][-400:])
    d = _check(out, 2, 1, False, n_gpus=2)
    assert d["oversubscribed"] is True
    assert abs(d["config"]["reads_total"] - 2 * 2048 * 2) < 0.25 * 2 * 2048 * 2      # world x reads per step x timed blocks (rank 0's shard: cut by bases, not by count)
    assert "2 contiguous shard(s)" in d["config"]["read_set"]
    # SUM over ranks: the whole job's input rate covers both shards (the bases of the timed blocks of both ranks over the MAX time)
    assert d["input_mbases_per_s"] >= d["value"] > 0
