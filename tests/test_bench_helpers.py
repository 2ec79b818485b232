"""CPU checks of bench.py's workload generator and CPU-baseline leg (no GPU needed)."""
import json
import os
import sys

import numpy as np
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_workload_generator_matches_explicit_projection():
    C, D, K = 5, 40, 4
    S, scale = bench.make_feature_scatters(C, D, K, "smsqfa", torch.device("cpu"), torch.float64)
    assert S.shape == (C, K, K) and scale == 1.0
    # explicit restatement: Sigma_c = A A^T + 0.05 I, Psi = Sigma + mu mu^T, S = F Psi F^T + 0.01 I
    gen = torch.Generator().manual_seed(1234)
    R = min(D, 128)
    F = torch.randn(K, D, generator=torch.Generator().manual_seed(7), dtype=torch.float64)
    F = F / F.norm(dim=1, keepdim=True)
    A = torch.randn(C, D, R, generator=gen, dtype=torch.float32).double() / R ** 0.5
    mu = 0.1 * torch.randn(C, D, generator=gen, dtype=torch.float32).double()
    Psi = A @ A.transpose(1, 2) + 0.05 * torch.eye(D, dtype=torch.float64) + mu[:, :, None] * mu[:, None, :]
    expect = F @ Psi @ F.T + 0.01 * torch.eye(K, dtype=torch.float64)
    assert torch.allclose(S, expect, atol=1e-12)
    E, scale = bench.make_feature_scatters(C, D, K, "sqfa", torch.device("cpu"), torch.float64)
    assert E.shape == (C, K + 1, K + 1) and scale == 0.5
    assert torch.linalg.eigvalsh(E).min() > 0


def test_cpu_baseline_leg_runs_the_oracle():
    S, scale = bench.make_feature_scatters(12, 30, 3, "smsqfa", torch.device("cpu"), torch.float32)
    out = bench.cpu_baseline(S, scale, 12)
    assert out["kind"] == "port" and out["unit"] == "evals/s" and out["value"] > 0 and out["cores"] >= 1
    assert out["full_size"] is True and out["cores"] == bench._entitled_threads() and out["cpu_model"]
    one = out["one_thread"]
    assert one["cores"] == 1 and one["full_size"] is False and one["value"] > 0
    json.dumps(out)


def test_pmc_summary_is_consistent():
    with open(os.path.join(ROOT, "profiles", bench.PMC_FILE)) as fh:
        pmc = json.load(fh)["kernels"]
    # the c3 closure projects from the block-triangular packed statistics since round 4: 332 800 elements per class at D=784
    proj = pmc["project_packed_kernel"]
    assert abs(proj["algorithmic_bytes_per_launch"] - 4 * 1000 * 332800) < 1
    assert 0.9 < proj["hbm_bytes_per_launch"] / proj["algorithmic_bytes_per_launch"] < 1.3
    assert np.isfinite(pmc["pair_tile_kernel"]["hbm_bytes_per_launch"])
    assert bench.pmc_traffic("project_packed_kernel", "c3", "f32") == proj["hbm_bytes_per_launch"]
    assert bench.pmc_traffic("project_packed_kernel", "c4", "f32") is None        # only quoted for the workload it was measured on


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` must work without torch.distributed.run: the parent starts one
    child per rank with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set and relays rank 0's line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launcher-selftest"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["world"] == 3 and line["rank"] == 0 and line["master"] == "127.0.0.1" and line["port"] > 0


def test_bench_launcher_stops_the_job_when_a_rank_dies():
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SQFA_BENCH_SELFTEST_FAIL_RANK"] = "1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "rank 1 exited with code 7" in out.stderr
    assert time.time() - t0 < 45          # did not sit out the other rank's wait


def test_class_shard_generator_is_the_same_data_set_for_every_rank_count():
    """bench.py's c4 closure leg: every rank generates only its own class shard; the union over the ranks must be the
    data set a single rank generates (one generator seed per chunk of 50 classes), for even and uneven splits."""
    C, D = 130, 24
    whole = bench.make_class_shard_statistics(C, D, 0, C, torch.device("cpu"))
    assert whole.shape == (C, D, D) and torch.allclose(whole, whole.transpose(1, 2))
    for world in (2, 3, 8):
        parts = [bench.make_class_shard_statistics(C, D, r * C // world, (r + 1) * C // world, torch.device("cpu"))
                 for r in range(world)]
        assert torch.equal(torch.cat(parts), whole)
    # second moments: Sigma + mu mu^T with Sigma = A A^T + 0.05 I  ->  positive definite
    assert torch.linalg.eigvalsh(whole.double()).min() > 0.04


def test_replicas_agree_on_one_rank():
    """parallel.replicas_agree compares an exact checksum of the bit patterns; on one gloo rank it must hold trivially
    and must be sensitive to a one-ulp change (the two-rank behaviour is tested in tests/test_distributed_gloo.py)."""
    import torch.distributed as dist
    from sqfa_amd.parallel import replicas_agree
    port = 34500 + os.getpid() % 2000
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        a = torch.randn(7, 5)
        assert replicas_agree([a, a.double()])
    finally:
        dist.destroy_process_group()
