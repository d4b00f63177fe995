"""The bench line the driver parses: the committed round bench lines (profiles/r*_bench_cfg2.json,
produced by `python bench.py` on the GPU box) carry every field of the contract, with the
roofline and cpu_baseline objects next to them. Runs on CPU (reads the committed JSON)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def test_committed_bench_line_has_every_contract_field():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cfg2.json")))
    assert files, "no committed bench line"
    j = json.load(open(files[-1]))
    for k in REQUIRED:
        assert k in j, k
    assert j["metric"].startswith("task-to-servant assignments/sec")
    assert j["unit"] == "assignments/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["data"] == "synthetic" and j["dtype"] in ("u32", "u64")
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - j["stats"]["granted"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    r = j["roofline"]
    for k in ROOFLINE:
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = j["cpu_baseline"]
    for k in CPU:
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] == 1
    assert j["parity_vs_cpu_baseline"] is True
    if files[-1].endswith("r01_bench_cfg2.json"):
        return
    # round 2 on: which rate `value` is, the SURVEY 8(d) end-to-end figure beside it, p99 over
    # at least 100 batches
    assert "HBM-resident" in j["value_definition"]
    e = j["end_to_end"]
    assert e["batches"] >= 100 and e["p99_ms"] >= e["p50_ms"] > 0
    assert abs(e["assignments_per_s"] - j["stats"]["granted"] / (e["ms_per_batch"] * 1e-3)) < 1e-6 * e["assignments_per_s"]
    assert j["latency_samples"] >= 100


def test_streaming_bench_line_has_reference_beside_it():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cfg5.json")))
    j = json.load(open(files[-1]))
    if files[-1].endswith("r01_bench_cfg5.json"):
        return
    assert j["cpu_baseline"]["kind"] == "reference" and j["parity_vs_cpu_baseline"] is True
    assert j["parity_ticks"] >= 20 and "roofline" in j


def test_bench_baseline_json_agrees():
    b = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "assignments/sec" in b["metric"]
    assert b.get("published") == {}  # nothing published for this path: vs_baseline stays null
