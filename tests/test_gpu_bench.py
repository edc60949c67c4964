"""bench.py's contract on the GPU box: the JSON line of a 1-rank run and of a 2-rank run launched exactly as the driver
launches it (torch.distributed.run, one process per rank; both ranks share this box's one GPU and talk over gloo -- the
test hooks MST_BENCH_BACKEND / MST_BENCH_ONE_DEVICE -- so the N > 1 control flow is exercised before an 8-GPU node is)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.strip().split("\n") if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def _fragments(err):
    """the RANK_FRAGMENT JSON objects on stderr (two ranks may write to the same line)"""
    import re
    return [json.loads(m.group(1)) for m in re.finditer(r"RANK_FRAGMENT (\{[^{}]*\})", err)]


def _check_line(j, n_gpus, steps, warmup):
    assert j["metric"].startswith("scale-space Mpix/s") and j["unit"] == "Mpix/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == n_gpus and j["steps"] == steps and j["warmup"] == warmup
    assert j["dtype"] == "f64" and j["data"] == "synthetic" and j["scaling"] == "strong" and j["vs_baseline"] is None
    assert abs(j["value"] - j["config"]["megapixels_per_step"] / (j["ms_per_step"] * 1e-3)) < 1e-3 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and r["peak"] == 39.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the roofline prices the pixels of the workgroups that RAN (a tile shared by two blocks once), not the block pixels
    assert abs(r["achieved"] - r["computed_pixels_per_launch"] * r["flops_per_pixel"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-2
    assert abs(r["computed_pixels_per_launch"] - r["pixels_per_launch"] * r["work_items"] / r["tiles"]) <= 1
    assert r["block_pixel_view"]["frac"] >= r["frac"] and j["tile_sharing"]["speedup"] > 1.0
    assert abs(j["tile_sharing"]["work_items"] - r["work_items"]) == 0
    assert r["executed_flops_per_pixel"] > r["flops_per_pixel"] * 0.9
    assert r["hbm_model"]["bytes_per_pixel_model"] == 592.0
    assert r["kernel_ms_per_step"] <= j["ms_per_step"] * 1.02          # the kernel fits inside the step it dominates
    rk = j["ranks"]
    assert rk["ms_per_step_max"] <= j["ms_per_step"] * 1.001 and rk["ms_per_step_min"] <= rk["ms_per_step_max"]
    assert rk["blocks_per_rank_max"] >= rk["blocks_per_rank_min"] and rk["imbalance_bound"] >= 1.0
    assert 0.0 < rk["efficiency_bound"] <= 1.0 and set(rk["efficiency_bound_at"]) == {"1", "2", "4", "8"}


KEPT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "ranks", "roofline", "tile_sharing", "band_skip", "no_share", "normalize_ms_untimed",
        "end_to_end", "chr21_5kb", "diff_chr21_5kb"}


def test_bench_one_rank_small():
    """the driver's line (no flags but the size): exactly the kept legs; then the same with --extra: the side legs beside them"""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--small", "--no-cpu",
                        "--no-file"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    # the labelled single-GPU projection of the strong split (every rank's range timed alone on this GPU)
    rk = j["ranks"]
    assert rk["kind"] == "single-GPU share timing, not a multi-GPU run"
    assert set(rk["projected_step_ms_at"]) == set(rk["projected_efficiency_at"]) == {"2", "4", "8"}
    for k in ("2", "4", "8"):
        assert len(rk["projected_step_ms_per_rank"][k]) == int(k) and rk["projected_step_ms_at"][k] == max(rk["projected_step_ms_per_rank"][k])
        assert 0.3 < rk["projected_efficiency_at"][k] <= 1.05
    _check_line(j, 1, 2, 1)
    assert set(j) == KEPT, sorted(set(j) ^ KEPT)
    assert j["scaling"] == "strong" and j["config"]["partition"] == "blocks" and "rank 0 = [0, 12)" in j["config"]["sharding"]
    assert j["band_skip"]["value"] > j["value"] and j["chr21_5kb"]["value"] > 0 and j["end_to_end"]["loops"] > 0
    assert j["diff_chr21_5kb"]["value"] > 0 and j["diff_chr21_5kb"]["ms_per_call"]["calls"] == 30
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--small", "--no-cpu",
                        "--no-file", "--extra"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    _check_line(j, 1, 2, 1)
    assert set(j) - KEPT == {"fma_mode", "normalize_roofline", "row2_scatter", "genome_5kb", "diff_genome_5kb", "variants"}
    assert abs(j["band_skip"]["roofline"]["frac"] - j["band_skip"]["roofline"]["achieved"] / 39.3) < 1e-3
    assert j["band_skip"]["roofline"]["block_pixel_view"]["frac"] >= j["band_skip"]["roofline"]["frac"]
    assert j["row2_scatter"]["band_rebuilt_identical"] is True and 0 < j["row2_scatter"]["value_from_coo"] < j["value"]
    g = j["genome_5kb"]
    assert g["blocks"] > 300 and g["chromosomes"] == 24 and g["value"] > 0 and g["end_to_end"]["loops"] > 0
    assert j["diff_genome_5kb"]["block_pairs"] == g["blocks"] and j["diff_genome_5kb"]["value"] > 0


def test_bench_two_ranks_gloo_one_device():
    env = dict(os.environ, MST_BENCH_BACKEND="gloo", MST_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--small",
           "--no-cpu"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    _check_line(j, 2, 2, 1)
    # default = the metric: STRONG scaling, ONE chromosome's 12 blocks in two contiguous ranges, the same megapixels per step
    # as at N = 1 (mustache.py:913-937: one process per block of one chromosome); the genome partition is timed beside it
    assert j["scaling"] == "strong" and j["config"]["partition"] == "blocks"
    assert j["ranks"]["blocks_per_rank_max"] == 6 and j["ranks"]["blocks_per_rank_min"] == 6
    assert "rank 0 = [0, 6), rank 1 = [6, 12)" in j["config"]["sharding"]
    assert abs(j["config"]["megapixels_per_step"] - 12 * 16.0) < 1e-6
    o = j["other_scaling"]
    assert o["scaling"] == "weak" and o["blocks_on_this_rank"] == 12 and abs(o["megapixels_per_step"] - 2 * 12 * 16.0) < 1e-6
    assert o["value"] > 0
    frags = _fragments(r.stderr)
    assert sorted(f["rank"] for f in frags) == [0, 1] and all(f["backend"] == "gloo" for f in frags)
    assert "cpu_baseline" not in j and "chr21_5kb" not in j        # rank-0-at-N=1-only legs stay out of the N > 1 line


def test_bench_two_ranks_self_launched_with_file_leg():
    """`python bench.py --gpus 2 --small` WITHOUT torchrun: bench.py starts its two ranks itself (free port), both on this
    box's one GPU over gloo.  The N-rank file leg runs: each rank inflates its share of the `.hic` blocks (the file is read
    once between the ranks), the shares are exchanged, the loops of the 2-rank run equal the 1-rank run's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MST_BENCH_BACKEND="gloo", MST_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--small", "--no-cpu"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    _check_line(j, 2, 2, 1)
    f = j["end_to_end_from_file"]
    assert f["ranks"] == 2 and len(f["read_s_per_rank"]) == 2 and j["ranks"]["read_s"] == f["read_s_per_rank"]
    assert all(c > 0 for c in f["records_per_rank"]) and sum(f["records_per_rank"]) == f["records"]
    assert abs(f["records_per_rank"][0] - f["records_per_rank"][1]) < 0.2 * f["records"]      # shares of the blocks, balanced
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0", "--small", "--no-cpu",
                          ], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert one.returncode == 0, one.stderr[-3000:]
    f1 = _last_json(one.stdout)["end_to_end_from_file"]
    assert f1["ranks"] == 1 and f1["records"] == f["records"] and f1["loops"] == f["loops"] > 0 and f1["n"] == f["n"]


def test_bench_four_ranks_self_launched_strong_split_and_file_leg():
    """The driver's N = 4 on this box's one GPU over gloo (`--small`: 12 blocks in four contiguous ranges of three, the `.hic`
    inflated in four shares and exchanged): same megapixels per step, same records and loops as one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MST_BENCH_BACKEND="gloo", MST_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1", "--small", "--no-cpu"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 4 and j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0 and 0 < j["roofline"]["frac"] < 1
    assert j["scaling"] == "strong" and j["config"]["partition"] == "blocks"
    assert "rank 0 = [0, 3), rank 1 = [3, 6), rank 2 = [6, 9), rank 3 = [9, 12)" in j["config"]["sharding"]
    assert j["ranks"]["blocks_per_rank_max"] == j["ranks"]["blocks_per_rank_min"] == 3
    assert abs(j["config"]["megapixels_per_step"] - 12 * 16.0) < 1e-6
    f = j["end_to_end_from_file"]
    assert f["ranks"] == 4 and len(f["records_per_rank"]) == 4 and all(c > 0 for c in f["records_per_rank"])
    assert sum(f["records_per_rank"]) == f["records"] and f["loops"] > 0
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "0", "--small", "--no-cpu"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert one.returncode == 0, one.stderr[-3000:]
    f1 = _last_json(one.stdout)["end_to_end_from_file"]
    assert f1["records"] == f["records"] and f1["loops"] == f["loops"] and f1["n"] == f["n"]


def test_bench_two_ranks_rccl_one_device_if_rccl_allows_it():
    """bench.py --gpus 2 with the REAL backend (nccl = RCCL) and both ranks on this box's one GPU.  RCCL may refuse two ranks
    on one device ("Duplicate GPU detected" / invalid usage) -- then this test says so and skips: the gloo run above keeps
    covering the N > 1 control flow and the 1-rank RCCL run below the RCCL calls.  Either way each rank's RANK_FRAGMENT line
    must be on stderr up to the point of failure."""
    env = dict(os.environ, MST_BENCH_BACKEND="nccl", MST_BENCH_ONE_DEVICE="1", NCCL_DEBUG="WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29549", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--small",
           "--no-cpu"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL with two ranks on one device did not complete (hang): gloo covers the N > 1 control flow")
    if r.returncode != 0:
        tail = (r.stderr + r.stdout)[-3000:]
        if any(k in tail for k in ("Duplicate GPU", "invalid usage", "invalid argument", "ncclInvalidUsage", "NCCL error",
                                   "ncclUnhandledCudaError", "DistBackendError")):
            pytest.skip("RCCL refuses two ranks on one device here: " + tail.strip().splitlines()[-1][:200])
        raise AssertionError(tail)
    j = _last_json(r.stdout)
    _check_line(j, 2, 2, 1)
    frags = _fragments(r.stderr)
    assert sorted(f["rank"] for f in frags) == [0, 1] and all(f["backend"] == "nccl" and f["blocks"] == 6 for f in frags)


def test_rccl_calls_run_in_a_one_rank_group(tmp_path):
    """No multi-GPU box is available to these tests, but the RCCL code path itself can run: a 1-rank `nccl` process group on
    this GPU through (a) bench.py's N > 1 control flow (MST_BENCH_FORCE_DIST: init_process_group(..., device_id), barrier,
    all_gather of the timings on device tensors) and (b) sharding.gather_records with force=True (all_gather of the counts
    and of the padded record block, device tensors)."""
    env = dict(os.environ, MST_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29563")
    # (two timed steps after a warm-up one: a single cold step also pays for the capture of the launch's hipGraph, and the
    # line's consistency checks compare that step with warmed-up ones)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--small", "--no-cpu"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_line(_last_json(r.stdout), 1, 2, 1)
    script = tmp_path / "g.py"
    script.write_text(
        "import numpy as np, torch, torch.distributed as dist\n"
        "from mustache_amd.sharding import gather_records, gather_loops\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "rec = np.arange(12, dtype=np.float64).reshape(3, 4)\n"
        "out = gather_records(rec, force=True)\n"
        "assert len(out) == 1 and np.array_equal(out[0], rec)\n"
        "assert gather_records(np.zeros((0, 4)), force=True)[0].shape == (0, 4)\n"
        "dist.barrier(); dist.destroy_process_group(); print('rccl ok')\n")
    env2 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29565", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-3000:]
