"""bench.py's last stdout line (stm32f4_sdr_gps_amd/benchline.py): compact, strict JSON, the contract's keys -- checked on canned
detail records without a GPU.  Round 5's 22 KB line (profiles/r05_bench_final.json) is one of the canned records: the driver could
not read it back; its condensed form must fit."""
import copy
import json
import os

import pytest

from stm32f4_sdr_gps_amd import benchline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    with open(os.path.join(ROOT, "profiles", "r05_bench_final.json")) as f:
        return json.load(f)


def test_round5_record_condenses_to_a_short_strict_line():
    detail = _canned()
    assert len(json.dumps(detail)) > 20000            # the record that was too long for the driver
    text = benchline.render(detail)
    assert len(text.encode()) <= benchline.MAX_LINE_BYTES and "\n" not in text
    back = json.loads(text, parse_constant=lambda name: pytest.fail(f"non-finite constant {name}"))
    for k in benchline.CONTRACT_KEYS:
        assert k in back, k
    assert back["value"] == pytest.approx(detail["value"], rel=1e-5)
    assert back["ms_per_step"] == pytest.approx(detail["ms_per_step"], rel=1e-5)
    assert back["roofline"]["frac"] == pytest.approx(detail["roofline"]["frac"], rel=1e-5)
    assert back["roofline"]["kernel"] == "gpsx::k_acq_mx<0>"
    assert back["cpu_baseline"]["kind"] == "reference" and back["cpu_baseline"]["cores"] == 1
    assert back["cpu_baseline_multicore"]["cores"] == 128
    assert back["tracking"]["value"] == detail["tracking"]["value"]
    assert back["tracking"]["device_loop"] == detail["tracking"]["closed_loop"]["device_loop"]["value"]
    assert back["config5"]["real_time"] is True
    assert back["configs3_one_gpu"]["value"] == pytest.approx(detail["configs3_one_gpu"]["value"], rel=1e-5)
    assert back["letter_compliant"]["kernel"].startswith("gpsx::k_acq_poly")
    # no ladder, no paragraph survives
    def walk(x):
        if isinstance(x, dict):
            for v in x.values():
                walk(v)
        elif isinstance(x, list):
            assert len(x) <= 4
        elif isinstance(x, str):
            assert len(x) <= 200
    walk(back)


def test_multi_gpu_record_keeps_the_communicator():
    detail = _canned()
    for k in ("native_grid", "configs3_one_gpu", "letter_compliant", "weighted_2bit_extension", "tracking", "cpu_baseline",
              "cpu_baseline_multicore", "pcie_inclusive"):
        detail.pop(k, None)
    detail.update({"n_gpus": 8,
                   "communicator": {"backend": "nccl", "library": "RCCL", "rccl_ranks": 8, "distinct_devices": 8,
                                    "devices": [{"rank": r, "uuid": "x" * 32, "name": "AMD Instinct MI355X"} for r in range(8)]},
                   "parity": {"parity_checked": True, "hypotheses_checked": 5237760, "against": "CPU oracle", "searches": [1, 2046]},
                   "single_search": {"ms_per_search": 1.3, "value": 8.4e10, "note": "n" * 400},
                   "per_gpu_unsharded": {"value": 1.06e12, "ms_per_step": 26.4, "note": "n" * 400},
                   "roofline_mfma": {"bound": "mfma", "achieved": 4400.0, "peak": 10000.0, "unit": "TFLOP/s", "frac": 0.44}})
    back = benchline.check(benchline.render(detail))
    assert back["communicator"] == {"backend": "nccl", "rccl_ranks": 8, "distinct_devices": 8}
    assert back["parity"]["parity_checked"] is True
    assert back["single_search"]["ms_per_search"] == 1.3 and back["per_gpu_unsharded"]["value"] == 1.06e12
    assert back["roofline_mfma"]["frac"] == 0.44


def test_check_refuses_long_nonfinite_and_incomplete_lines():
    good = benchline.render(_canned())
    benchline.check(good)
    with pytest.raises(ValueError, match="bytes"):
        benchline.check(good[:-1] + "," + json.dumps("pad")[:-1] + "x" * 5000 + '":1}')
    with pytest.raises(ValueError, match="non-finite"):
        benchline.check(good.replace('"vs_baseline":null', '"vs_baseline":NaN'))
    with pytest.raises(ValueError, match="lacks"):
        benchline.check(json.dumps({"metric": "m", "value": 1.0}))
    with pytest.raises(ValueError, match="newline"):
        benchline.check(good + "\n")


def test_nonfinite_numbers_and_errors_of_secondary_legs_do_not_take_the_line():
    detail = copy.deepcopy(_canned())
    detail["native_grid"] = {"error": "RuntimeError('" + "x" * 1000 + "')"}
    detail["letter_compliant"]["roofline_valu"]["frac"] = float("nan")
    detail["tracking"]["closed_loop"] = {"error": "boom"}
    back = benchline.check(benchline.render(detail))
    assert len(back["native_grid"]["error"]) <= 100
    assert back["letter_compliant"].get("frac") is None
    assert back["tracking"]["value"] == detail["tracking"]["value"] and "config5" not in back


def test_render_safe_falls_back_to_the_contract_keys():
    detail = _canned()
    detail["config"]["workload"] = None
    detail["roofline"].pop("frac")                   # the full record no longer passes check()
    text = benchline.render_safe(detail, err=open(os.devnull, "w"))
    back = json.loads(text)
    assert back["value"] == pytest.approx(detail["value"], rel=1e-5) and "line_error" in back
    assert len(text) < benchline.MAX_LINE_BYTES


def test_detail_file_is_written_beside_the_script(tmp_path):
    (tmp_path / "gpurun_out").mkdir()
    written = benchline.write_detail(_canned(), str(tmp_path))
    assert written == ["bench_detail.json", os.path.join("gpurun_out", "bench_detail.json")]
    assert json.load(open(tmp_path / "bench_detail.json"))["tracking"]["ladder"]
