"""The record bench.py hands the driver: ONE compact, strict-JSON line on stdout, everything else in a file beside it.

Round 5's line had grown to 22 KB (ladders, per-count dicts, paragraph-long notes) and the driver could not read it
back (BENCH_r05.json: "parsed": null).  From round 6 on bench.py builds the full record as before -- the DETAIL -- and
this module
  * writes the detail to bench_detail.json (repo root, and gpurun_out/ when that exists: gpurun merges it back),
  * condenses it to the contract's keys plus one or two numbers per secondary leg (`compact`),
  * refuses to let a line out that is long (MAX_LINE_BYTES), not strict JSON (NaN / Infinity) or missing a contract key
    (`check`) -- tests/test_benchline.py runs both on a canned detail record without a GPU.
Nothing here measures anything.
"""
import json
import math
import os

MAX_LINE_BYTES = 4096
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline")


def _pick(src, keys):
    """{k: src[k]} for the keys present with a scalar (or short-list) value"""
    if not isinstance(src, dict):
        return None
    out = {}
    for k in keys:
        if k in src and src[k] is not None:
            out[k] = src[k]
    return out


def _sig(x, digits=6):
    """floats to `digits` significant digits (a 17-digit repr is 2 KB of the line for no information); non-finite -> None"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        if x == 0.0:
            return 0.0
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return str(x)


def _short(s, n=100):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def compact(detail):
    """The driver's line from the full record.  Every string is bounded, every ladder stays behind in the detail file."""
    d = detail
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                  "scaling", "vs_baseline")}
    line["dtype"] = _short(d.get("dtype"), 60)
    line["data"] = _short(d.get("data"), 90)
    cfg = d.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload"), 200),
                      **(_pick(cfg, ("searches_per_gpu_per_step", "hypotheses_per_step", "blocks_per_search")) or {}),
                      "parallelism": _short(cfg.get("parallelism"), 120), "inputs": _short(cfg.get("inputs"), 60)}
    roof = d.get("roofline") or {}
    line["roofline"] = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "frac_basis", "frac_live", "frac_profiled_mean",
                                    "kernel", "kernel_ms", "profiled_kernel_ms_mean", "profiled_launches", "traffic",
                                    "traffic_gbs", "traffic_frac_of_hbm_peak", "algorithmic_bytes", "traffic_over_algorithmic", "mfma_busy_frac",
                                    "counter_source"))
    for k in ("roofline_valu", "roofline_mfma"):
        if isinstance(d.get(k), dict):
            line[k] = _pick(d[k], ("bound", "achieved", "peak", "unit", "frac"))
    if "value_pcie_inclusive" in d:
        line["value_pcie_inclusive"] = d["value_pcie_inclusive"]
    cb = d.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {**(_pick(cb, ("value", "unit", "cores", "kind", "error")) or {}),
                                "cpu": _short(cb.get("cpu"), 60), "sample": _short(cb.get("sample"), 100)}
    if isinstance(d.get("cpu_baseline_multicore"), dict):
        line["cpu_baseline_multicore"] = _pick(d["cpu_baseline_multicore"], ("value", "cores", "kind"))

    def leg(name, keys, roof_key=None):
        src = d.get(name)
        if not isinstance(src, dict):
            return
        if "error" in src:
            line[name] = {"error": _short(src["error"], 100)}
            return
        out = _pick(src, keys) or {}
        r = src.get(roof_key) if roof_key else None
        if isinstance(r, dict):
            out.update({k: r[k] for k in ("bound", "frac", "frac_live", "traffic", "traffic_over_algorithmic", "counter_source")
                        if r.get(k) is not None})
        line[name] = out

    leg("native_grid", ("value", "ms_per_launch", "kernel"), "roofline")
    leg("configs3_one_gpu", ("value", "ms_per_step", "kernel"), "roofline")
    leg("configs3_sharded", ("value", "ms_per_step", "kernel"), "roofline")
    leg("letter_compliant", ("value", "ms_per_launch", "kernel", "keys_identical_to_the_matrix_core_path"), "roofline_valu")
    leg("weighted_2bit_extension", ("value", "ms_per_launch"), "roofline")
    trk = d.get("tracking")
    if isinstance(trk, dict):
        if "error" in trk:
            line["tracking"] = {"error": _short(trk["error"], 100)}
        else:
            t = {"value": trk.get("value"), "unit": "channels (E/P/L step p99 < 1 ms)"}
            cl = trk.get("closed_loop") if isinstance(trk.get("closed_loop"), dict) else {}
            dl = cl.get("device_loop") if isinstance(cl.get("device_loop"), dict) else {}
            # (the host-mode ladder -- one host round trip per millisecond on a shared host: a latency reading that moves with
            #  the other tenants -- stays in the detail file; the line carries the device loop's counts)
            if dl.get("value") is not None:
                t["device_loop"] = dl["value"]
            if isinstance(dl.get("mux17"), dict) and dl["mux17"].get("value") is not None:
                t["device_loop_mux17"] = dl["mux17"]["value"]
            for k in ("error",):
                if k in cl:
                    t["closed_loop_error"] = _short(cl[k], 100)
            line["tracking"] = t
            c5 = cl.get("config5")
            if isinstance(c5, dict):
                line["config5"] = _pick(c5, ("channels", "ms", "loops", "ms_per_launch", "deadline_us", "launches_over_deadline",
                                             "launch_max_us", "real_time", "code_and_carrier_lock", "figure_from"))
    for k, keys in (("communicator", ("backend", "rccl_ranks", "distinct_devices")),
                    ("parity", ("parity_checked", "hypotheses_checked", "against")),
                    ("single_search", ("ms_per_search", "value")),
                    ("per_gpu_unsharded", ("value", "ms_per_step"))):
        if isinstance(d.get(k), dict):
            line[k] = _pick(d[k], keys)
    if isinstance(d.get("device"), dict):
        line["device"] = _pick(d["device"], ("name", "compute_units", "clock_khz"))
    for k in ("fits_in_driver_run", "bench_wall_s", "detail"):
        if k in d:
            line[k] = d[k]
    return _sig(line)


def _no_constants(name):
    raise ValueError(f"non-finite number in the bench line: {name}")


def check(text):
    """What must hold for the string bench.py prints last: one line, short, strict JSON, the contract's keys."""
    if "\n" in text:
        raise ValueError("the bench line contains a newline")
    n = len(text.encode())
    if n > MAX_LINE_BYTES:
        raise ValueError(f"the bench line is {n} bytes (limit {MAX_LINE_BYTES}): move the excess to bench_detail.json")
    back = json.loads(text, parse_constant=_no_constants)
    missing = [k for k in CONTRACT_KEYS if k not in back]
    if missing:
        raise ValueError(f"the bench line lacks {missing}")
    if not isinstance(back["value"], (int, float)) or not back["value"] > 0:
        raise ValueError(f"value = {back['value']!r}")
    for k in ("bound", "achieved", "peak", "unit", "frac"):
        if k not in back["roofline"]:
            raise ValueError(f"roofline lacks {k}")
    return back


def render(detail):
    """detail -> the checked line (a str).  Raises instead of returning something the driver cannot parse."""
    text = json.dumps(compact(detail), allow_nan=False, separators=(",", ":"))
    check(text)
    return text


def render_safe(detail, err=None):
    """`render`, but a record that cannot be rendered (a bug in a secondary leg's bookkeeping) still leaves a line: the contract's
    keys alone plus the reason -- printed to `err` (default sys.stderr) as well.  The job must never end without its line."""
    try:
        return render(detail)
    except Exception as exc:       # noqa: BLE001
        import sys
        print(f"bench line: falling back to the contract keys only: {exc!r}", file=err or sys.stderr, flush=True)
        roof = detail.get("roofline") or {}
        minimal = {k: detail.get(k) for k in CONTRACT_KEYS if k not in ("config", "roofline")}
        minimal["dtype"], minimal["data"] = _short(minimal.get("dtype"), 60), _short(minimal.get("data"), 90)
        minimal["config"] = {"workload": _short((detail.get("config") or {}).get("workload"), 200)}
        minimal["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms", "traffic")}
        cb = detail.get("cpu_baseline")
        if isinstance(cb, dict):
            minimal["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        minimal["line_error"] = _short(repr(exc), 200)
        return json.dumps(_sig(minimal), allow_nan=False, separators=(",", ":"))


def write_detail(detail, root):
    """bench_detail.json beside bench.py and under gpurun_out/ (when present); returns the paths written.  Failures to write
    are reported in the return value, never raised: the line matters more than its appendix."""
    written = []
    text = json.dumps(_sig(detail, 9), indent=1)
    for path in (os.path.join(root, "bench_detail.json"), os.path.join(root, "gpurun_out", "bench_detail.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(text + "\n")
                written.append(os.path.relpath(path, root))
        except OSError:
            pass
    return written
