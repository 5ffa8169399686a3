"""bench.py's byte accounting and its use of the committed PMC summary -- no GPU needed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_stage_bytes_add_up_to_the_iteration_formula():
    import bench
    P, R, N, Tn, K = 500_000, 3_650_000, 1600 * 1062, 100 * 67, 16.0
    sb = bench.stage_bytes(P, R, N, Tn, K)
    # B_iter = P(718 + 36 K) + 280 R + 40 N + the 8 Tn ranges term; 24 R x 6 passes is the reference's 45-bit sort
    total = sum(sb.values())
    assert abs(total - (P * (718 + 36 * K) + 280 * R + 40 * N + 8 * Tn)) < 1e-6 * total
    assert set(sb) == set(bench.STAGE_KERNELS) | {"depth_sort_scan"}


def test_pmc_traffic_uses_the_committed_counters():
    import bench
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
    for stage, kernels in bench.STAGE_KERNELS.items():
        for name, launches in kernels:
            assert name in pmc, name
    t = bench.pmc_traffic("blend_bwd", "metric_500k_1600x1062")
    want = sum((2 * pmc[k]["FETCH_SIZE"] + pmc[k]["WRITE_SIZE"]) * 1024 for k in ("r3::blend_bwd_kernel<4>", "r3::pair_reduce_kernel"))
    assert t == int(want) and 3e8 < t < 7e8
    assert bench.pmc_traffic("blend_bwd", "some_other_workload") is None
