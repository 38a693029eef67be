"""CPU: the committed bench line (profiles/r02_bench_line.json, written by `python bench.py` on an MI355X) carries every
field the bench contract names, and bench.py's constants agree with BASELINE.json's configs[1]."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r02_bench_line.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["n_gpus"] == 1
    assert "workload" in d["config"] and "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and r["bound"] in ("hbm", "mfma")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    c = d["cpu_baseline"]
    assert set(c) >= {"value", "unit", "cores", "kind", "sample"} and c["kind"] in ("port", "reference") and c["cores"] >= 1
    assert abs(d["value"] - d["n_gpus"] * 250 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]   # 250 tokens per step per GPU


def test_committed_cb_lines_have_the_contract_fields():
    for name, cfgname in (("r02_cb_configs2.json", "configs[2]"), ("r02_cb_fp8_bs64.json", "configs[4]")):
        d = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, (name, k)
        assert cfgname in d["config"]["workload"] and d["scaling"] == "weak" and d["vs_baseline"] is None
        r = d["roofline"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 8000.0
        assert abs(d["value"] - d["tokens_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" in json.loads(open(os.path.join(ROOT, "profiles", "r02_cb_configs2.json")).read().strip().splitlines()[-1])


def test_bench_constants_match_the_named_workload():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert (b.N_PROMPT_PH, b.N_TEXT_PH, b.N_PROMPT_TOK, b.N_NEW) == (40, 60, 100, 250) and b.FRAMES == 500
    assert b.HBM_PEAK_GBS == 8000.0 and b.GPT_CACHE == [(1, 512), (1, 1024)] and (b.CB_REQUESTS_PER_GPU, b.CB_SLOTS) == (256, 32)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "bs=1" in base["configs"][1] and "V2Pro" in base["configs"][1]
