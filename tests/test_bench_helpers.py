"""CPU checks of bench.py's measurement plumbing (no GPU): the committed profile reference resolves for every leg the line quotes, the
agreement check flags what it should, the PMC traffic files exist for every engine / geometry the line prices, and the closed-form issued-flop
count of the F(4x4) gates launch matches the library's formula (conv_issued_flops, conv3x3_mfma.hip) re-stated here."""
import importlib.util
import json
import os

import pytest

from tests.helpers import ROOT

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_profile_reference_covers_every_leg_of_the_line():
    ref, path = bench.profile_ref()
    assert ref is not None and path.startswith("profiles/")
    for leg in ("isolated", "live"):
        for key, prec in (("w172_l4_fp32", "fp32"), ("w172_l4_fp16", "fp16"), ("w172_l4_bf16", "bf16"), ("w168_l12_fp32", "fp32"), ("w168_l12_fp16", "fp16")):
            ent = ref[leg][key]
            assert os.path.exists(os.path.join(ROOT, ent["stats_file"])), ent["stats_file"]
            rows = [k for k in ent["kernels"] if k.startswith(bench.GATES_KERNEL[prec])]
            assert len(rows) == 1, (leg, key, rows)
            assert ent["kernels"][rows[0]]["calls"] >= 12 and ent["kernels"][rows[0]]["avg_us"] > 100
            assert "--profile-leg %s" % leg in ent["command"]


def test_profile_check_agrees_and_disagrees(capsys):
    ref, _ = bench.profile_ref()
    row = [v for k, v in ref["isolated"]["w172_l4_fp32"]["kernels"].items() if k.startswith(bench.GATES_KERNEL["fp32"])][0]
    ok = bench.profile_check("isolated", "w172_l4_fp32", "fp32", row["avg_us"] * 1.05e-3)
    assert ok["agree"] is True and abs(ok["ratio_bench_over_profile"] - 1.05) < 1e-9 and ok["stats_file"].endswith("_kernel_stats.md")
    bad = bench.profile_check("isolated", "w172_l4_fp32", "fp32", row["avg_us"] * 1.25e-3)
    assert bad["agree"] is False
    assert "WARNING" in capsys.readouterr().err
    assert bench.profile_check("isolated", "w999_l1_fp32", "fp32", 1.0)["agree"] is None          # a leg nobody profiled: reported as such, never a crash


def test_pmc_traffic_files_exist_for_every_priced_leg():
    for prec, win, length in (("fp32", 172, 4), ("fp16", 172, 4), ("bf16", 172, 4), ("fp32", 168, 12), ("fp16", 168, 12)):
        traffic, src = bench.pmc_traffic(prec, win, length)
        assert traffic and src.startswith("profiles/r"), (prec, win, length)
        d = json.load(open(os.path.join(ROOT, src)))
        assert 1.0 <= d["traffic_over_algorithmic"] < 1.5 and os.path.exists(os.path.join(ROOT, d["raw"]))


def test_issued_flops_closed_form_matches_the_library_formula():
    """conv_issued_flops for the F(4x4) gates launch (conv3x3_wino4.hip launch_w4 geometry), W = 172, 36 windows x 2 directions, mean over L = 4"""
    W, n, n_per_set, Cout, L = 172, 72, 36, 64, 4
    RR = ((W + 15) // 16) ** 2
    ntiles = ((RR * min(n, n_per_set) + 1) // 2) * (n // n_per_set) * (Cout // 64)

    def launch(cin_run):
        nrun = max(3, (cin_run + 7) // 8)
        rem = min(49, nrun * 8) - 8 * (nrun - 1)
        return ntiles * 8 * ((nrun - 1) * 72 + ((rem + 3) // 4) * 36) * 2.0 * 16 * 16 * 4
    mean = ((L - 1) * launch(49) + launch(17)) / L
    r = bench.roofline("fp32", W, 36, 0.459, 12, L)
    assert abs(r["mfma_flops_issued_per_launch"] / mean - 1.0) < 1e-12
    assert abs(mean / 1e9 - 28.904) < 0.01                                                          # what ttc_debug_kernel_flops reported on the GPU (r06_g_bench.json)
    # the line's frac is issued / time / peak
    assert abs(r["frac"] - mean / 0.459e-3 / 157.3e12) < 1e-12


def test_table_totals():
    t = {"conv_gates": {"ms": 0.5, "n": 8, "flops": 2e9}, "dsen2_conv": {"ms": 0.4, "n": 24, "flops": 1e9}}
    ms, fl = bench.table_totals(t, 2)
    assert ms == pytest.approx((0.5 * 8 + 0.4 * 24) / 2) and fl == pytest.approx((2e9 * 8 + 1e9 * 24) / 2)
