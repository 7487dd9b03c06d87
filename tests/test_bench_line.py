"""bench.py's stdout contract, checked without a GPU: the ONE line the driver parses is built from canned numbers and must
carry the contract keys, `roofline` and `cpu_baseline` in under 4 KB whatever the long record holds (round 5's 25.8 KB line
left BENCH_r05.json's `parsed` null)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("mrhip_bench_py", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _canned(nrows=60, long_text=400):
    pad = "x" * long_text
    base = {t: {"kernel": "rational_opair_kernel", "kernel_ms": 1.2345, "wall_ms": 1.3456, "frac": 0.4321, "frac_wall": 0.4012}
            for t in ("C1", "C2", "C2s", "C2r", "C2rp", "C2rd", "C3a", "C3b", "C4", "C4f", "C5")}
    return {
        "metric": "Msamples/s in (Float32, 147//160, 24*147 taps) + achieved HBM GB/s vs roofline",
        "value": 650000.123, "unit": "Msamples/s (input samples, all channels, all GPUs)", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 9.8461, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "FIRRational 147//160 3528 taps f32, 64 ch x 1e+08 samples/GPU, one filt! per pass, HBM-resident " + pad,
                   "channels_per_gpu": 64, "samples_per_channel": 100000000, "chunk": 100000000, "numerics": "strict",
                   "parallelism": "channel-shard x1, no collective", "backend": None},
        "output_msamples_s": 597187.5, "kernel": "rational_opair_kernel", "parity_pin": "algorithmic",
        "roofline": {"bound": "hbm", "achieved": 4988.7, "peak": 8000.0, "unit": "GB/s", "frac": 0.6236, "traffic": 49434277376.0,
                     "traffic_source": {"file": "profiles/traffic_latest.json", "how": pad}, "algorithmic_bytes_per_launch": 49120000000.0,
                     "avg_launch_ms": 9.84612, "launches_timed": 20, "whole_step_GBps": 4988.1, "baseline_configs": base},
        "parity": {"kind": "algorithmic", "text": pad * 4},
        "fused": {"note": pad, "headline": {"frac": 0.70}, "C3a": {"frac": 0.53}},
        "streamed_1e6_chunks": {"Msamples_per_s": 590000.0, "frac": 0.57, "chunk": 1000000},
        "configs": [{"name": f"X row {i} " + pad, "kernel": "k", "kernel_ms": 1.0, "note": pad} for i in range(nrows)],
        "cpu_baseline": {"value": 121.8, "unit": "Msamples/s", "cores": 1, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                         "host_logical_cores": 256, "runs": 5, "sample": "1 ch x 3e+08 f32 samples " + pad, "note": pad},
        "cpu_baseline_all_cores": {"value": 1386.7, "cores": 256, "sample": pad},
        "cpu_baseline_simd": {"value": 170.1, "flags": pad},
    }


def test_compact_line_carries_the_contract_and_stays_small():
    b = _bench_module()
    full = _canned()
    assert len(json.dumps(full)) > 25000          # the long record is as long as round 5's line
    text = b.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in text and len(text) < 4096
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "kernel", "parity_pin"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["dtype"] == "f32"
    assert line["config"]["workload"].startswith("FIRRational 147//160") and len(line["config"]["workload"]) <= 110
    assert "model" not in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_timed"):
        assert rf[k] == full["roofline"][k], k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert set(rf["baseline_configs"]) == set(full["roofline"]["baseline_configs"])
    assert rf["baseline_configs"]["C4"] == {"kernel_ms": 1.2345, "frac": 0.4321, "frac_wall": 0.4012}
    cb = line["cpu_baseline"]
    assert cb["value"] == 121.8 and cb["cores"] == 1 and cb["kind"] == "port" and cb["unit"] == "Msamples/s" and len(cb["sample"]) <= 100
    assert line["cpu_baseline_simd"]["value"] == 170.1
    assert line["full_record"] == "gpurun_out/bench_full.json"
    # the timed region the line claims fits any run that contains it
    assert line["ms_per_step"] * line["steps"] / 1e3 < 1.0


def test_compact_line_drops_optional_keys_before_it_grows():
    b = _bench_module()
    full = _canned()
    # a multi-GPU record: c5 object beside the headline
    full["c5"] = {"value": 1.0e6, "unit": "Msamples/s " + "y" * 300, "scaling": "strong", "ms_per_step": 4.1, "n_gpus": 8, "channels_total": 4096,
                  "channels_per_gpu": 512, "samples_per_channel": 1000000, "kernel": "rational_opair_kernel",
                  "roofline": {"achieved": 4400.0, "frac": 0.55, "avg_launch_ms": 1.78, "note": "z" * 500},
                  "gather": {"root": {"ms": 30.0, "GBps_into_one_gpu": 900.0}, "all": {"ms": 40.0, "GBps_into_one_gpu": 700.0}}}
    line = json.loads(b.compact_line(full, "f.json"))
    assert line["c5"]["n_gpus"] == 8 and line["c5"]["gather"]["root"]["ms"] == 30.0 and line["c5"]["roofline"]["frac"] == 0.55
    # absurdly many BASELINE rows: optional keys go, the contract stays
    full["roofline"]["baseline_configs"] = {f"C{i}": {"kernel_ms": 1.0, "frac": 0.5, "frac_wall": 0.4} for i in range(200)}
    was = b.BASELINE_ROW_TAGS
    text = b.compact_line(full, "f.json")
    assert len(text) < 4096 and b.BASELINE_ROW_TAGS is was
    line = json.loads(text)
    assert "baseline_configs" not in line["roofline"] and "cpu_baseline" in line and line["roofline"]["frac"] == 0.6236


def test_emit_prints_one_line_and_writes_the_long_record(tmp_path, capsys):
    import argparse
    b = _bench_module()
    full = _canned(nrows=5)
    out = tmp_path / "sub" / "bench_full.json"
    b.emit(full, argparse.Namespace(full_out=str(out), full_stderr=False))
    printed = capsys.readouterr().out.splitlines()
    assert len(printed) == 1 and len(printed[0]) < 4096
    assert json.loads(printed[0])["value"] == full["value"]
    assert json.load(open(out))["configs"][4]["name"].startswith("X row 4")
