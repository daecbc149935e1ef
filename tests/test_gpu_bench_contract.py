"""-m gpu: bench.py keeps its contract — exactly ONE JSON line on stdout (native libraries' banners go to stderr),
the required keys, the roofline / cpu_baseline objects, and consistency between value and ms_per_step."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", "--no-large-n"],
                       cwd=ROOT, capture_output=True, text=True, timeout=550)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 10 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "MPPI N=1024" in d["config"]["workload"] and "H=50" in d["config"]["workload"]
    assert abs(d["value"] - 1024 * 50 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["value"] > 1e8                                            # far above the 1e6 target on any healthy box
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel"].startswith("ctk_mppi_rollout")
    assert rf["kernel_timed_launches"] >= 3 and 5.0 < rf["kernel_us"] < 100.0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 1e5        # oracle/ctk_cpu.c on the box's host cores (C + OpenMP)
    assert "ctk_cpu.c" in cb["sample"] and d["cpu_baseline_numpy"]["cores"] == 1     # the single-thread NumPy oracle stays beside it
    res = d.get("resident")                                                          # the opt-in resident form: its own field, never `value`
    assert res is not None and "error" not in res, res
    assert res["value"] > 1e8 and res["kernel_launches_total"] >= 1 and res["boundary"] == "controller_mpc.step"


@pytest.mark.timeout(600)
def test_bench_gpus_2_as_a_bare_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment (the shape of the driver's 1-GPU command): the script starts one
    process per rank as a child (torch.distributed.run), relays rank 0's line and the return code.  Rehearsed on the one-GPU box with
    both ranks on device 0 and gloo instead of RCCL (CTK_BENCH_SINGLE_DEVICE / CTK_BENCH_BACKEND are never set by the driver)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CTK_BENCH_SINGLE_DEVICE="1", CTK_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       cwd=ROOT, capture_output=True, text=True, timeout=550, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["priming"] == 64 and d["scaling"] == "strong"
    assert "mppi_cfg5" in d["config"]["workload"] and d["config"]["global_rollouts"] == 65536
    assert d["exchange_us"] is not None and len(d["roofline_per_rank"]) == 2
    assert abs(d["value"] - 65536 * 100 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
