"""The ONE JSON line `bench.py` prints, checked against the driver's contract on the committed evidence of the last round (no GPU needed): every
required key with the right type, the roofline / cpu_baseline objects, and the arithmetic that ties the fields together (value = samples x
denoise steps / step time; frac = achieved / peak; traffic close to the algorithmic bytes) -- a contract regression in bench.py would show up in
the next evidence file this test is pointed at."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PROFILES = os.path.join(os.path.dirname(HERE), "profiles")
LINES = ["r04_final_bench_driver_form.json", "r04_final_bench_b8_ncfg1.json", "r04_final_bench_b4_ncfg2.json"]


def _load(name):
    path = os.path.join(PROFILES, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} is not committed")
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"{name}: bench.py prints exactly one JSON line"
    return json.loads(lines[0])


@pytest.mark.parametrize("name", LINES)
def test_bench_line_follows_the_driver_contract(name):
    d = _load(name)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert key in d and isinstance(d[key], typ), (key, type(d.get(key)))
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert d["metric"].startswith("denoise-steps/sec") and d["unit"] == "denoise-steps/sec" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["n_gpus"] == 1
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg
    # value = samples x denoise steps per rollout / seconds per rollout, summed over the ranks
    per_step = cfg["global_batch"] * cfg["denoise_steps"] / (d["ms_per_step"] * 1e-3)
    assert abs(per_step - d["value"]) <= 2e-3 * d["value"], (per_step, d["value"])
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 5e-4 and 0.2 < r["frac"] < 1.0
    # achieved = algorithmic FLOPs per launch / live average launch duration
    assert abs(r["flops_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e12 - r["achieved"]) <= 2e-3 * r["achieved"]
    assert isinstance(r["traffic"], int) and 0.8 < r["traffic"] / 435.4e6 < 1.5          # PMC bytes per launch vs the algorithmic 435 MB
    assert "pmc_attention.json@" in r["traffic_source"]
    fwd = r["forward"]
    assert abs(fwd["achieved"] / 2500.0 - fwd["frac"]) < 5e-4


def test_default_bench_line_carries_the_cpu_baseline_and_the_round_4_objects():
    d = _load("r04_final_bench_driver_form.json")
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and 0 < c["value"] < 1.0
    assert isinstance(d["roofline"]["mfma_busy"], float) and 0.3 < d["roofline"]["mfma_busy"] < 1.0
    assert d["roofline"]["gemm"]["frac"] > 0.3
    p = d["power"]
    assert p["watts"]["median"] <= p["cap_watts"] * 1.02 and p["joules_per_denoise_step"] > 0
    for leg in ("optimize_step", "optimize_step_flux1", "optimize_step_qwen_image"):
        assert d[leg]["ratio_is_one"] is True and d[leg]["ms_forward_backward"] > d[leg]["ms_forward_train"] > 0, leg
    fam = d["families"]
    for leg in ("flux1_dev_b8_1024", "wan21_t2v_1p3b_b2_cfg_480x832x49", "qwen_image_b2_cfg_1328"):
        assert fam[leg]["finite"] is True and 0.3 < fam[leg]["forward_frac"] < 1.0, leg
