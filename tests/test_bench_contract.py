"""The committed bench line of the current round (profiles/r05_bench_line.json, produced by `python bench.py` on the GPU box) carries every
field of the driver's contract: metric/value/unit/n_gpus/steps/warmup/ms_per_step/higher_is_better/scaling/vs_baseline/
dtype/data/config.workload plus the roofline and cpu_baseline objects."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = open(os.path.join(ROOT, "profiles", "r05_bench_line.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    assert "configs[1]" in d["config"]["workload"] and d["dtype"] == "f16" and d["metric"] == "samples/sec Wide&Deep Criteo batch16384"
    e = d["roofline_embedding_path"]
    assert abs(e["frac"] - e["algorithmic_bytes"] / (e["sum_ms"] * 1e-3) / 1e9 / 8000.0) < 2e-3
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 16384 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_bench_gpus_n_self_launches_ranks_before_any_gpu_call():
    """`python bench.py --gpus 2` typed as is (no launcher environment) must start two ranks itself, through
    torch.distributed.run as a child process, and propagate their failure -- on this GPU-less machine each rank ends
    with "needs an MI355X", which proves both were started and that the parent reports a non-zero code."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        import torch
        if torch.cuda.is_available():
            import pytest
            pytest.skip("GPU present: covered by the driver's multi-GPU bench")
    except ImportError:
        pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--repeats", "1",
                        "--vocab", "1000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert (r.stdout + r.stderr).count("needs an MI355X") >= 2, (r.stdout + r.stderr)[-2000:]
