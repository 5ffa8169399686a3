"""bench.py's byte accounting and its use of the committed PMC summary -- no GPU needed."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_stage_bytes_add_up_to_the_iteration_formula():
    import bench
    P, R, N, Tn, K = 500_000, 3_650_000, 1600 * 1062, 100 * 67, 16.0
    sb = bench.stage_bytes(P, R, N, Tn, K)
    # B_iter = P(718 + 36 K) + 280 R + 40 N + the 8 Tn ranges term; 24 R x 6 passes is the reference's 45-bit sort
    total = sum(sb.values())
    assert abs(total - (P * (718 + 36 * K) + 280 * R + 40 * N + 8 * Tn)) < 1e-6 * total
    assert set(sb) == set(bench.STAGE_KERNELS)


def test_pmc_traffic_uses_the_committed_counters():
    import bench
    path = os.path.join(ROOT, bench.PMC_SUMMARY)
    if not os.path.exists(path):
        pytest.skip("no PMC summary committed for this round yet")
    pmc = json.load(open(path))
    for stage, kernels in bench.STAGE_KERNELS.items():
        for name, launches, wide in kernels:   # names are prefixes (the pair-word policy depends on the workload)
            assert bench.find_kernel(pmc, name) is not None, name
    t = bench.pmc_traffic("blend_bwd", "metric_500k_1600x1062")
    # FETCH_SIZE x2 for the kernels whose loads are 16 B / lane (STAGE_KERNELS' wide flag), WRITE_SIZE as reported
    want = sum(launches * ((2 if wide else 1) * bench.find_kernel(pmc, name)["FETCH_SIZE"] + bench.find_kernel(pmc, name)["WRITE_SIZE"])
               for name, launches, wide in bench.STAGE_KERNELS["blend_bwd"]) * 1024
    assert t == int(want) and 1e8 < t < 7e8
    assert bench.pmc_traffic("blend_bwd", "some_other_workload") is None
    v = bench.pmc_valu("blend_bwd", "metric_500k_1600x1062", 0.4)
    assert v and v["kernel"] == bench.STAGE_KERNELS["blend_bwd"][0][0]
    assert abs(sum(v["insts_by_class"].values()) - v["insts"]) <= 8 and 0.3 < v["frac"] < 1.0
    assert 3.5 < v["sq_active_cycles_per_inst"] < 5.0
    assert bench.pmc_valu("blend_bwd", "some_other_workload", 0.4) is None


def test_cgroup_probe_never_raises():
    import bench
    cg = bench.cgroup_cpu()
    assert cg is None or {"quota_cpus", "nr_throttled", "throttled_usec", "usage_usec"} <= set(cg)
    assert bench.effective_cpus() >= 1


def test_find_kernel_refuses_an_ambiguous_prefix():
    """a profile holding several instantiations behind one open prefix must not be resolved by picking the first."""
    import bench
    table = {"r3::radix_scatter_kernel<r3::IoNarrow, 7>": {"FETCH_SIZE": 1.0},
             "r3::radix_scatter_kernel<r3::IoNarrow, 8>": {"FETCH_SIZE": 2.0},
             "r3::unit_order_kernel<1>": {"FETCH_SIZE": 3.0}, "r3::pair_reduce_kernel": {"FETCH_SIZE": 4.0}}
    del bench.AMBIGUOUS_KERNELS[:]
    assert bench.find_kernel(table, "r3::radix_scatter_kernel<") is None
    assert bench.AMBIGUOUS_KERNELS == ["r3::radix_scatter_kernel<"]
    assert bench.find_kernel(table, "r3::radix_scatter_kernel<r3::IoNarrow, 7>")["FETCH_SIZE"] == 1.0   # exact name wins
    assert bench.find_kernel(table, "r3::unit_order_kernel")["FETCH_SIZE"] == 3.0                        # one match
    assert bench.find_kernel(table, "r3::pair_reduce_kernel")["FETCH_SIZE"] == 4.0
    assert bench.find_kernel({"r3::tile_order_kernel": {"x": 1}}, "r3::unit_order_kernel") == {"x": 1}    # round-4 name
    assert bench.find_kernel(table, "r3::nothing") is None
    del bench.AMBIGUOUS_KERNELS[:]


def test_own_algorithm_bytes_stay_below_the_reference_algorithms():
    """own_stage_bytes prices what THIS build has to move: with fewer pairs binned than the reference counts and no 64-bit key
    sort, the binning stage must come out far below SURVEY 8d's figure, and no stage may be negative or missing."""
    import bench
    P, R, Rb, N, Tn, K = 500_000, 3_650_000, 2_370_000, 1600 * 1062, 100 * 67, 16.0
    ref, own = bench.stage_bytes(P, R, N, Tn, K), bench.own_stage_bytes(P, 0.9 * P, Rb, N, Tn, K)
    assert set(own) == set(ref) and all(v > 0 for v in own.values())
    assert own["tile_binning"] < 0.25 * ref["tile_binning"]          # 29 B per binned pair against 164 B per reference pair
    assert own["preprocess_bwd"] < ref["preprocess_bwd"]              # no 300 P of fills
    assert own["blend_fwd"] < ref["blend_fwd"] * 1.05
    split = bench.own_stage_bytes(2_000_000, 1_800_000, 14_000_000, N, Tn, 9.0, word_bytes=6, key_bytes=2)
    assert split["tile_binning"] > 14_000_000 * 30


def test_exchange_model_arithmetic():
    """multiview.exchange_model (bench.py --gpus N): bytes per row, per link and phase, and the
    serialised speed-up they allow."""
    sys.path.insert(0, os.path.join(ROOT, "reduced-3dgs_amd"))
    from multiview import exchange_model as m
    d = m(500_000, 8, 0.67)
    assert d["bytes_per_row"] == 248 and d["buffer_bytes"] == 124_000_000 and d["bytes_per_link_per_phase"] == 15_500_000
    # two phases of 15.5 MB at 153 GB/s + two collective latencies
    assert abs(d["exchange_ms"] - (2 * (15.5e6 / 153e9 * 1e3 + 0.025))) < 1e-3
    assert abs(d["speedup"] - 8 * 0.67 / (0.67 + d["exchange_ms"])) < 0.01
    b = m(500_000, 8, 0.67, sh_rest_bf16=True)
    assert b["bytes_per_row"] == 158 and b["speedup"] > d["speedup"]
    s = m(2_000_000, 8, 1.25, sparse=True, union_frac=0.5, sh_rest_bf16=True)
    assert s["rows"] == 1_000_000 and s["exchange_ms"] < m(2_000_000, 8, 1.25, sh_rest_bf16=True)["exchange_ms"]
    assert m(500_000, 1, 0.67)["speedup"] == 1.0
