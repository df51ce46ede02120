"""The one JSON line `bench.py` prints is the driver's record of the round: its compact form must carry every key the contract names,
the figures a reader of BENCH_rNN.json needs inside `roofline`, and stay well inside the 8 KB the driver keeps.  Checked on the verbose
record of the round's last GPU run (profiles/r5_bench_cfg3_final_full.json) -- no GPU needed."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "profiles", "r5_bench_cfg3_final_full.json")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


@pytest.mark.skipif(not os.path.exists(FULL), reason="the verbose record of the last GPU run is not in this tree")
def test_compact_record_carries_the_contract():
    full = json.load(open(FULL))
    rec = _bench().compact_record(full)
    line = json.dumps(rec)
    assert len(line) < 4096, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["unit"] == "frames/s" and rec["higher_is_better"] is True and rec["vs_baseline"] is None and rec["dtype"] == "bf16"
    assert "workload" in rec["config"] and "model" not in rec["config"]
    r = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "encode_ms", "decode_ms",
              "encode_frac_of_mfma_peak", "mfma_busy", "tolerance_mode", "tolerance_mode_mixed", "fastest_mode_inside_tolerance"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(rec["value"] * rec["ms_per_step"] / 1e3 - 17.0) < 0.05          # frames/s x s per clip = the clip's 17 frames
    for k in ("tolerance_mode", "tolerance_mode_mixed"):
        assert r[k]["meets_north_star_tolerance"] is True and r[k]["latent_max_abs"] <= 1e-3
    best = r["fastest_mode_inside_tolerance"]
    assert best["value"] == max(r["tolerance_mode"]["value"], r["tolerance_mode_mixed"]["value"]) and best["mode"] in r
    cb = rec["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1
