"""CPU-only: the committed bench line of the round (profiles/r01_bench_s1m_v*.json, written by bench.py on the MI355X) carries
every field of the driver's contract, and bench.py's argument defaults are the contract's."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = glob.glob(os.path.join(ROOT, "profiles", "r01_bench_s1m_v*.json"))
    return max(files, key=lambda f: int(re.search(r"_v(\d+)\.json$", f).group(1)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(_latest()))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "iters/s" and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "1000000 Gaussians @ 1600x1063" in d["config"]["workload"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.01 * d["value"]            # whole-job throughput = 1 view per step at N = 1
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["kernel"] == "blend_backward" and r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_bench_defaults_are_single_gpu_and_quick():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1\b', src)
    steps = int(re.search(r'"--steps", type=int, default=(\d+)', src).group(1))
    warm = int(re.search(r'"--warmup", type=int, default=(\d+)', src).group(1))
    assert 1 <= warm < steps <= 100
    assert 'default=1_000_000' in src and 'default=1600' in src and 'default=1063' in src     # BASELINE.json's S1M configuration
