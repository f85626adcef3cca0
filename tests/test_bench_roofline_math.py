"""bench.py's roofline arithmetic, without a GPU: the fractions in the line must be what anybody recomputes from the files under
profiles/ -- algorithmic flops (or record bytes) of the launch / the MEAN launch of the committed rocprofv3 kernel trace / the
peak -- and the committed round-6 line must agree with its own trace to 1 %."""
import csv
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (numpy only at import time; torch is imported inside main())

HYP_256 = 256 * 32 * 21 * 16368


def _trace_mean_ns(csv_name, kernel_substring):
    with open(os.path.join(ROOT, "profiles", csv_name)) as f:
        rows = [r for r in csv.DictReader(f) if kernel_substring in r["Name"]]
    assert len(rows) == 1, [r["Name"] for r in rows]
    return float(rows[0]["AverageNs"]), int(rows[0]["Calls"])


def test_headline_fraction_is_flops_over_the_committed_traces_mean_launch():
    counters = bench._kernel_counters("k_acq_mx<0>", 256, 1)
    assert counters and counters["trace_source"].startswith("profiles/") and counters["kernel_trace_calls"] >= 50
    mean_ns, calls = _trace_mean_ns(os.path.basename(counters["trace_source"]), "k_acq_mx<0>")
    assert calls == counters["kernel_trace_calls"] and mean_ns == pytest.approx(counters["kernel_trace_avg_ns"], rel=1e-9)
    roof = bench._mx_roofline(HYP_256, launch_ms=2.17, counters=counters, clk_khz=2400000)
    flops = 17 * 2 * 2.0 * 32 * 1024 * 1024 * 256 * 21          # 17 passes x 2 streams x 2 M N K per (capture, Doppler) pair
    assert bench.MFMA_FLOPS_PER_HYP * HYP_256 == pytest.approx(flops, rel=1e-12)
    assert roof["frac"] == pytest.approx(flops / (mean_ns * 1e-9) / 1e16, rel=1e-9)      # against 10 PF dense FP4
    assert roof["frac_live"] == pytest.approx(flops / 2.17e-3 / 1e16, rel=1e-9)
    assert roof["achieved"] == pytest.approx(roof["frac"] * 10000.0) and "mean launch of the committed" in roof["frac_basis"]
    # without a committed trace of the kernel and launch shape the live figure is the fraction, and says so
    live = bench._mx_roofline(HYP_256, 2.17, None, 2400000)
    assert live["frac"] == live["frac_live"] and live["frac_basis"].startswith("live")


def test_ten_block_fraction_is_record_bytes_over_the_committed_traces_mean_launch():
    counters = bench._kernel_counters("k_acq_mx<3>", 256, 10)
    mean_ns, _ = _trace_mean_ns(os.path.basename(counters["trace_source"]), "k_acq_mx<3>")
    roof, mfma = bench._walk_roofline("k_acq_mx<3>", 10 * HYP_256, 10, 27.0, counters)
    rec_bytes = 10 * HYP_256 * 4.0 * 9 / 10                     # 2 B read + 2 B written per hypothesis and block, 9 of 10 blocks
    assert roof["algorithmic_bytes"] == pytest.approx(rec_bytes) and roof["bound"] == "hbm"
    assert roof["frac"] == pytest.approx(rec_bytes / (mean_ns * 1e-9) / 8e12, rel=1e-9)
    assert 0.95 < roof["traffic_over_algorithmic"] < 1.05 and roof["traffic_over_compulsory"] > 1000
    assert mfma["frac"] == pytest.approx(bench.MFMA_FLOPS_PER_HYP * 10 * HYP_256 / (mean_ns * 1e-9) / 1e16, rel=1e-9)


def test_vector_alu_fraction_uses_this_rounds_counters_and_cannot_exceed_one_at_the_measured_rate():
    per_hyp, src, ent = bench._poly_counters(256, "k_acq_poly<8,16,0>")
    assert ent["searches_per_launch"] == 256 and "r06" in src and 150 < per_hyp < 260
    rates = bench._valu_class_rates()
    assert rates["and_bcnt_pair_tlane_ops"] > rates["four_cycle_class_tlane_ops"] > 30 and "r06" in rates["source"]
    mean_ns, _ = _trace_mean_ns("r06_poly_kernel_stats.csv", "k_acq_poly")
    roof = bench._poly_valu_roofline(per_hyp, HYP_256 / (mean_ns * 1e-9), src)
    t_peak = 128.0 / rates["and_bcnt_pair_tlane_ops"] + (per_hyp - 128.0) / rates["four_cycle_class_tlane_ops"]     # ps per hypothesis
    assert roof["frac"] == pytest.approx(t_peak * 1e-12 * HYP_256 / (mean_ns * 1e-9), rel=1e-9)
    assert 0.8 < roof["frac"] <= 1.0


def test_the_committed_line_agrees_with_its_own_trace_to_one_per_cent():
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_1gpu.json")))
    basis = line["roofline"]["frac_basis"]
    csv_name = basis[basis.index("profiles/") + len("profiles/"):basis.index(".csv") + 4]
    mean_ns, _ = _trace_mean_ns(csv_name, "k_acq_mx<0>")
    assert line["roofline"]["frac"] == pytest.approx(bench.MFMA_FLOPS_PER_HYP * HYP_256 / (mean_ns * 1e-9) / 1e16, rel=0.01)
    assert line["roofline"]["profiled_kernel_ms_mean"] == pytest.approx(mean_ns * 1e-6, rel=1e-4)
    assert abs(line["roofline"]["frac_live"] / line["roofline"]["frac"] - 1) < 0.06      # the live figure of the same box is near it
