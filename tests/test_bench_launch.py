"""bench.py --gpus N must start its N ranks by itself: the driver's `python bench.py --gpus 8` has no launcher
around it.  Checked here without a GPU: JV_BENCH_LAUNCH_CHECK=1 makes every rank report itself and stop before torch/HIP."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = dict(os.environ, JV_BENCH_LAUNCH_CHECK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_plain_invocation_with_gpus_2_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    # (the ranks share one stdout pipe: their lines may arrive glued together — look for the reports, not for whole lines)
    import re
    reports = sorted(re.findall(r"launch-check rank \d+ of \d+ local_rank \d+ gpus \d+", r.stdout))
    assert reports == ["launch-check rank 0 of 2 local_rank 0 gpus 2", "launch-check rank 1 of 2 local_rank 1 gpus 2"], r.stdout


def test_single_gpu_invocation_does_not_spawn():
    r = _run([])
    assert r.returncode == 0 and r.stdout.strip() == "launch-check rank 0 of 1 local_rank 0 gpus 1"


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 3


def test_kept_bench_line_has_the_contract_fields():
    """the line `bench.py` printed on the final tree (kept under profiles/ next to the rocprofv3 summary of the same command)
    carries what the driver's contract names: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
    scaling / vs_baseline / dtype / data / config.workload, a roofline object whose fraction is achieved / peak, and a CPU
    baseline with its unit, core count, kind and sample."""
    import glob
    import json
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_c3_10m", "bench_final_tree.json")))
    assert paths, "no kept bench line"
    line = open(paths[-1]).read().strip().splitlines()[-1]
    j = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["vs_baseline"] is None and j["data"].startswith("synthetic")
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.9
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"]
    # value = queries of the timed steps / time: steps x batch / (steps x ms_per_step)
    assert abs(j["value"] - j["config"]["queries_per_step"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 0.01


def test_roofline_arithmetic_of_the_bench_line():
    """the numbers `bench.py` derives, on made-up counters (no GPU): SURVEY 8(d)'s bytes per query for the three layouts, the
    roofline object (fraction = achieved / peak on the KERNEL's time, the whole call's fraction beside it)"""
    sys.path.insert(0, ROOT)
    import bench
    nq, launches, d, R, M = 1000, 2, 768, 32, 32
    visited, reranked, expanded = 4000.0 * nq, 1200.0 * nq, 1377.0 * nq
    fused = bench.algorithmic_bytes(visited, reranked, expanded, nq, launches, M, d, R, True)
    assert fused == expanded * R * (M + 4) + reranked * 4 * d + launches * 1024 * d
    plain = bench.algorithmic_bytes(visited, reranked, expanded, nq, launches, M, d, R, False)
    assert plain == visited * M + expanded * 4 * (R + 1) + reranked * 4 * d + launches * 1024 * d
    assert bench.algorithmic_bytes(visited, 0.0, expanded, nq, launches, 0, d, R, False) == visited * 4 * d + expanded * 4 * (R + 1)
    # per query at the headline's counters: 1 377 x 32 x 36 + 1 200 x 3 072 (+ the shared codebook read)
    assert abs(fused / nq - (1377 * 32 * 36 + 1200 * 3072 + launches * 1024 * d / nq)) < 1e-6
    r = bench.roofline_object(1.382e12, 464.0, 489.0, 1.5e12, "kept", "jv_search_pqw_kernel", M, True)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # the fraction is on the WHOLE call's time: the bytes are counted over every row, whichever launch finished it
    assert abs(r["achieved"] - 1.382e12 / 0.489 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["frac"] == r["frac_whole_call"]
    assert r["kernel_avg_ms"] == 464.0 and r["call_avg_ms"] == 489.0 and r["traffic"] == 1.5e12 and "reranked*4d" in r["formula"]
    assert abs(r["first_launch"]["share_of_call"] - 464.0 / 489.0) < 1e-3
    # a step whose rows mostly finish in a heavy second launch (round 5's mixtureB leg printed 1.04 on the first launch's time):
    # 0.9 TB counted over all rows, a 100 ms first launch, a 900 ms call -> 1 000 GB/s = 0.125, never above 1
    h = bench.roofline_object(0.9e12, 100.0, 900.0, None, None, "jv_search_pqw_kernel", M, True)
    assert abs(h["frac"] - 0.125) < 1e-4 and h["frac"] <= 1.0 and abs(h["first_launch"]["share_of_call"] - 1 / 9) < 1e-3
