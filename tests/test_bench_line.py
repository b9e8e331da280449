"""CPU: the line bench.py hands the driver.  Round 4's 20.6 KB line did not parse on the driver's side: the last
stdout line is now a compact, flat record held to LINE_LIMIT bytes and to strict JSON, built by a pure function that is
tested here on canned measurements (a full line of round 4, committed under profiles/) — no GPU, no oracle."""
import copy
import io
import json
import os
from contextlib import redirect_stdout

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.fixture(scope="module")
def full():
    return json.load(open(os.path.join(ROOT, "profiles", "round4c_bench_line.json")))


def strict(text):
    def no_constant(name):
        raise ValueError("non-JSON constant " + name)

    return json.loads(text, parse_constant=no_constant)


def test_compact_line_fits_and_round_trips(full):
    assert len(json.dumps(full)) > 4 * bench.LINE_LIMIT  # the canned tree really is the one that did not parse
    line = bench.compact_line(full)
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.LINE_LIMIT <= 4096
    assert "\n" not in text
    assert strict(text) == line
    for k in REQUIRED:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert set(line["config"]) == {"workload", "rays_per_step_per_gpu", "parallelism"}


def test_compact_line_keeps_the_roofline_and_the_cpu_baseline_recomputable(full):
    line = bench.compact_line(full)
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms", "rays_per_launch", "algorithmic_flops_per_ray",
              "traffic", "traffic_source", "path_frac_of_binding_ceiling", "hbm_frac"):
        assert k in r, k
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    # achieved = algorithmic flops per ray x rays per launch / the launch's duration
    assert r["achieved"] == pytest.approx(r["algorithmic_flops_per_ray"] * r["rays_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12, rel=1e-4)
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "cpu_model", "single_thread", "physical_cores", "sample"):
        assert k in c, k
    assert isinstance(c["single_thread"], float) and isinstance(c["physical_cores"], int)  # numbers, not nested samples
    assert line["speedup_vs_cpu"] == pytest.approx(line["value"] / c["value"], rel=1e-3)
    assert set(line["parity"]) == {"rgb_mae", "thermal_mae"}


def test_compact_line_shows_config3_and_every_variant_as_numbers(full):
    line = bench.compact_line(full)
    v = line["variants"]
    c3 = v["train_config3_S192"]
    for k in ("value", "ms_per_step", "frac", "steps", "rgb_psnr_db", "thermal_mae_degC", "parity_rgb_mae", "parity_thermal_mae"):
        assert isinstance(c3[k], (int, float)), k
    assert c3["steps"] == 30000 and c3["ms_per_step"] == pytest.approx(full["variants"]["train_config3_S192"]["ms_per_step"], rel=1e-4)
    # a training variant's `frac` is SURVEY 8(d)'s fraction (3 x B(S) bytes per ray against 8 TB/s), recomputable from the line itself;
    # the fp32-MFMA view and the builder's serial-phase number sit beside it under their own names
    roof3 = full["variants"]["train_config3_S192"]["roofline"]
    assert c3["frac"] == pytest.approx(3 * 309320 * 4096 / (c3["ms_per_step"] * 1e-3) / 8e12, rel=2e-3) == pytest.approx(roof3["frac"], rel=1e-3)
    assert c3["mfma_frac"] == pytest.approx(roof3["mfma_view"]["frac"], rel=1e-3)
    assert c3["serial_phase_frac"] == pytest.approx(roof3["serial_phase_view"]["frac"], rel=1e-3) and c3["serial_phase_frac"] != c3["frac"]
    for name, c in v.items():
        assert all(not isinstance(x, (dict, list)) for x in c.values()), name
    for name in full["variants"]:
        assert name in v or name == "shard_proxy"
    assert v["shard_proxy_800x800_S192"]["n8_efficiency"] == pytest.approx(
        full["variants"]["shard_proxy"]["frames"]["800x800_S192"]["N8"]["implied_efficiency"], rel=1e-3)
    assert v["bf16x6"]["frac"] == pytest.approx(full["variants"]["bf16x6"]["roofline"]["frac"], rel=1e-3)


def test_compact_line_at_eight_ranks_with_non_finite_numbers(full):
    big = copy.deepcopy(full)
    big["n_gpus"] = 8
    big["rccl"] = {"backend": "nccl", "world_size": 8, "rccl_version": "2.26.6",
                   "devices": ["AMD Instinct MI355X (cuda:%d)" % i for i in range(8)]}
    big["variants"]["strong_frame_1080p_S48"] = {"metric": "x", "value": 4.1e8, "ms_per_step": 5.0, "n_gpus": 8, "scaling": "strong",
                                                 "roofline": {"frac": 0.5}}
    big["roofline"]["traffic"] = float("nan")
    big["variants"]["bf16x6"]["value"] = float("inf")
    line = bench.compact_line(big)
    text = json.dumps(line, allow_nan=False)  # raises on NaN / inf
    assert len(text) < bench.LINE_LIMIT
    assert line["roofline"]["traffic"] is None and line["variants"]["bf16x6"]["value"] is None
    assert line["rccl"]["world_size"] == 8 and line["rccl"]["device_ids"] == list(range(8))
    assert line["rccl"]["devices"] == ["AMD Instinct MI355X"]
    assert line["variants"]["strong_frame_1080p_S48"] == {"value": 4.1e8, "ms_per_step": 5.0, "frac": 0.5, "n_gpus": 8, "scaling": "strong"}


def test_compact_line_sheds_optional_parts_rather_than_exceed_the_limit(full):
    huge = copy.deepcopy(full)
    for i in range(200):
        huge["variants"]["extra_%03d" % i] = {"value": 1.0 + i, "ms_per_step": 2.0, "roofline": {"frac": 0.25}}
    huge["config"]["workload"] = "w" * 5000
    line = bench.compact_line(huge)
    assert len(json.dumps(line, allow_nan=False)) <= bench.LINE_LIMIT
    for k in REQUIRED:
        assert k in line, k
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)


def test_train_mode_line(full):
    t = full["variants"]["train_config3_S192"]
    line = {"metric": "rays/sec (train step: forward + losses + backward + Adam) @ 4096 rays/step", "value": t["value"], "unit": "rays/s",
            "n_gpus": 1, "steps": t["steps"], "warmup": 1, "ms_per_step": t["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": t["what"], "parallelism": "scene replica per rank x1"},
            "roofline": t["roofline"], "held_out": t["held_out"], "trained_weights_parity": t["trained_weights_parity"],
            "ms_per_step_by_window": t["ms_per_step_by_window"]}
    c = bench.compact_line(line)
    assert len(json.dumps(c, allow_nan=False)) < bench.LINE_LIMIT
    assert c["roofline"]["serial_phase_frac"] == pytest.approx(t["roofline"]["serial_phase_view"]["frac"], rel=1e-4)
    assert c["held_out"]["rgb_psnr_db"] == pytest.approx(t["held_out"]["rgb_psnr_db"], rel=1e-3)
    assert c["trained_weights_parity"]["rgb_mae"] == pytest.approx(t["trained_weights_parity"]["rgb_mae"], rel=1e-2)


def test_emit_prints_the_detail_first_and_the_compact_line_last(full, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "gpurun_out")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(copy.deepcopy(full))
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2
    assert strict(lines[0])["bench_detail"]["variants"]["train_config3_S192"]["ms_per_step_by_window"]
    last = strict(lines[-1])
    assert len(lines[-1]) < bench.LINE_LIMIT and last["detail"] == "bench_detail.json"
    assert last == bench.compact_line(full)
    for d in (tmp_path, tmp_path / "gpurun_out"):
        assert strict(open(d / "bench_detail.json").read())["value"] == full["value"]
