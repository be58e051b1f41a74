"""CPU-only: the committed bench line of the round (profiles/rNN_bench_s1m_v*.json, written by bench.py on the MI355X) carries
every field of the driver's contract, bench.py's argument defaults are the contract's, `--gpus N` really starts N ranks, and the
committed PMC counter pass the bench line quotes was collected on the kernels as they are now."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r0[0-9]_bench_s1m_v*.json")) if re.search(r"_v(\d+)\.json$", f)]     # (a file named otherwise is not a bench line of the series)
    return max(files, key=lambda f: (int(re.search(r"r(\d+)_bench", f).group(1)), int(re.search(r"_v(\d+)\.json$", f).group(1))))


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
    rnd = int(re.search(r"r(\d+)_bench", os.path.basename(_latest())).group(1))
    if rnd >= 2:                                          # round 2 on: the blend kernels' VALU roofline + where the PMC figures come from
        for k in ("blend_forward", "blend_backward"):
            v = r["valu"][k]
            assert v["bound"] == "valu" and v["peak"] == 157.3 and abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-3
            assert abs(v["achieved"] - v["flop_per_pair"] * v["pairs"] / (v["avg_ms"] * 1e-3) / 1e12) < 0.02 * v["achieved"]
        if os.path.exists(os.path.join(ROOT, "profiles", "r%02d_pmc_traffic_s1m.json" % rnd)) and r.get("traffic_source"):
            src = r["traffic_source"]
            assert src["file"].startswith("profiles/") and (src.get("kernel_sha16") or src.get("sha16_by_kernel_at_collection"))
        assert d["steps"] >= 100
        assert "integrate" in d and d["integrate"]["later_call_of_the_view"]["wall_ms"] > 0
    if rnd >= 3:                                          # round 3: the heavy-tailed leg, the workspace footprint, the PyTorch-CPU render beside the HIP forward
        cl = d["clustered"]
        assert "S1M-clustered" in cl["workload"] and cl["ms_per_step"] > 0 and cl["num_rendered"] > 20_000_000
        assert cl["entries_walked_per_tile_pct_0_50_90_99_100"][4] > 5 * cl["entries_walked_per_tile_pct_0_50_90_99_100"][1]      # heavy-tailed indeed
        ws = r["workspace"]
        assert ws["reference_per_instance_bytes"] == 24
        if rnd == 3:
            assert 100 <= ws["per_instance_total_bytes"] <= 130
        elif "staged_fraction_of_instances" in ws:        # round 4: record pool sized for the staged entries (+ the mask pool, reported beside it)
            assert ws["per_instance_total_bytes"] <= 80 and 0 < ws["staged_fraction_of_instances"] < 1
        tc = d["cpu_baseline"]["torch_cpu"]
        assert "400x400" in tc["workload"] and tc["torch_cpu_float32_s"] > 0 and tc["hip_forward_ms"] > 0
    if rnd >= 4 and "views" in d:                          # round 4: posed cameras cycled inside a timed region; the reference's own kernels on this GPU
        assert d["views"]["ms_per_step"] > 0 and len(d["views"]["per_view_alone_ms"]) == 8
        rs = d["reference_same_gpu"]
        assert rs["available"] and rs["ms_per_step"] > 5 * d["ms_per_step"] and abs(rs["product_speedup"] - rs["ms_per_step"] / d["ms_per_step"]) < 0.05 * rs["product_speedup"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_bench_defaults_are_single_gpu_and_quick():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1\b', src)
    steps = int(re.search(r'"--steps", type=int, default=(\d+)', src).group(1))
    warm = int(re.search(r'"--warmup", type=int, default=(\d+)', src).group(1))
    assert 1 <= warm < steps and steps >= 100
    assert 'default=1_000_000' in src and 'default=1600' in src and 'default=1063' in src     # BASELINE.json's S1M configuration


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gof_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_n_without_a_launcher_spawns_n_ranks(monkeypatch):
    """`python bench.py --gpus 8` (RANK unset) must not silently run one rank: it re-runs itself under torch.distributed.run."""
    import subprocess
    import sys
    b = _bench_module()
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7"])
    assert b.spawn_ranks(8) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_world_size_that_contradicts_gpus_is_refused():
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but the launcher started 1 rank" in (r.stderr + r.stdout)


def test_the_committed_pmc_pass_was_collected_on_the_current_blend_kernels():
    """bench.py quotes HBM traffic / VALU issue figures from profiles/<PMC_FILE> (rocprofv3 cannot run in-process); the file records
    the hashes of the kernel sources it was collected on.  An edit of the dominant kernel without a fresh counter pass fails here."""
    b = _bench_module()
    f = os.path.join(ROOT, "profiles", b.PMC_FILE)
    if not os.path.exists(f):
        import pytest
        pytest.skip("no PMC pass of this round committed yet")
    pmc = json.load(open(f))
    assert b.pmc_pass_is_current(pmc, "blend_backward")        # the dominant kernel, whose traffic the bench line quotes (bench.py)
