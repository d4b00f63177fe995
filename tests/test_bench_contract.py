"""The bench line the driver parses. The -m gpu tests RUN bench.py on the GPU box (a few steps
of cfg2, the streaming ticks of cfg5, the group path of one rank) and validate the line that
comes out — a change that breaks the contract fails here. The CPU part checks what can be
checked without a device: BASELINE.json and the validator itself on the committed lines."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def validate(j, steps=None, streaming=False, cpu_baseline=True):
    for k in REQUIRED:
        assert k in j, k
    assert j["metric"].startswith("task-to-servant assignments/sec")
    assert j["unit"] == "assignments/s" and j["higher_is_better"] is True
    assert j["scaling"] in ("weak", "strong") and j["vs_baseline"] is None
    assert j["data"] == "synthetic" and j["dtype"] in ("u32", "u64")
    assert "workload" in j["config"] and "model" not in j["config"]
    if steps is not None:
        assert j["steps"] == steps
    assert j["value"] > 0 and j["ms_per_step"] > 0
    r = j["roofline"]
    for k in ROOFLINE:
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0 < r["frac"] < 1
    if cpu_baseline:
        c = j["cpu_baseline"]
        for k in CPU:
            assert k in c, k
        assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0
        assert j["parity_vs_cpu_baseline"] is True
    if not streaming:
        # value = granted requests x steps / wall time of the timed region
        granted = j.get("granted_all_ranks", j["stats"]["granted"])  # (lines before round 4: one rank)
        assert abs(j["value"] - granted / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
        if j["n_gpus"] == 1:
            assert granted == j["stats"]["granted"]
        assert "HBM-resident" in j["value_definition"]
        assert j["latency_samples"] >= 100
        assert j["p99_dispatch_latency_ms"] >= j["p50_dispatch_latency_ms"] > 0


def run_bench(*args, env=None, timeout=420):
    e = dict(os.environ)
    for k, v in (env or {}).items():
        if v is None:
            e.pop(k, None)
        else:
            e[k] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT,
                         env=e, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-800:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly ONE JSON line: %r" % out.stdout[-800:]
    assert out.stdout.strip().splitlines()[-1] == lines[0]  # ... and it is the last thing on stdout
    # ... short enough for the driver to ingest (BENCH_r05: a 23.8 KB line came back "parsed": null)
    assert len(lines[0]) < 8192, len(lines[0])
    line = json.loads(lines[0])
    validate_line(line)
    # everything else that was measured is in the file the line names
    detail = json.load(open(os.path.join(ROOT, line["detail_file"])))
    for k in REQUIRED:
        assert detail[k] == line[k], k
    detail["_line"] = line
    return detail


def validate_line(j, cpu_baseline=None):
    """What the driver reads: the contract keys, `roofline`, `cpu_baseline`."""
    for k in REQUIRED:
        assert k in j, k
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ROOFLINE:
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    if cpu_baseline if cpu_baseline is not None else "cpu_baseline" in j:
        for k in CPU:
            assert k in j["cpu_baseline"], k
        assert j["parity_vs_cpu_baseline"] is True
    assert j["detail_file"] == "bench_detail.json"


@pytest.mark.gpu
def test_bench_cfg2_line():
    j = run_bench("--steps", "3", "--warmup", "2")
    validate(j, steps=3)
    assert j["n_gpus"] == 1 and j["warmup"] == 2
    assert "cfg2: 100000 pending requests x 2000 servants" in j["config"]["workload"]
    assert j["stats"]["granted"] + j["stats"]["timeouts"] + j["stats"]["env_not_found"] == 100000
    e = j["end_to_end"]
    assert e["batches"] >= 100 and e["p99_ms"] >= e["p50_ms"] > 0
    assert abs(e["assignments_per_s"] - j["stats"]["granted"] / (e["ms_per_batch"] * 1e-3)) < 1e-6 * e[
        "assignments_per_s"]
    assert j["roofline"]["kernel"] in j["kernels_us_per_step"]
    # the three rates side by side, whichever of them `value` is
    assert j["value_pipelined"] == j["value"] and j["value_synchronous"] > 0
    assert j["value_end_to_end"] == e["assignments_per_s"]
    # steady state: every batch COMMITs and its grants are released again
    s = j["steady_state_commit"]
    assert s["registry_restored"] is True and s["same_placement"] is True and s["ms_per_step"] > 0
    # the other single-GPU configurations of BASELINE.json, timed in the same run and pinned to
    # the committed fixtures of the verbatim reference
    c = j["configs"]
    assert set(c) == {"cfg3", "cfg4", "cfg5", "cfg2_150_digests"}
    d = c["cfg2_150_digests"]
    assert "with 150 digests" in d["workload"] and d["parity_vs_oracle"] is True and d["conservation"] is True
    assert d["ms_per_step"] < 40  # (84 ms with the lone walker of round 3)
    assert "1000000 pending requests x 8000 servants" in c["cfg3"]["workload"]
    assert "4000000 pending requests x 16000 servants" in c["cfg4"]["workload"]
    for k in ("cfg3", "cfg4"):
        r = c[k]
        assert r["parity_vs_reference_fixture"] is True
        assert r["fixture_requests"] == {"cfg3": 400_000, "cfg4": 200_000}[k]
        assert r["conservation"] is True and r["value"] > 0 and r["ms_per_step"] > 0
        assert r["p99_dispatch_latency_ms"] >= r["p50_dispatch_latency_ms"] > 0
        assert r["roofline"]["kernel"] in r["kernels_us_per_step"] and 0 < r["roofline"]["frac"] < 1
        assert r["end_to_end_ms"] > 0
        assert r["latency_samples"] >= 100
        assert r["cpu_baseline"]["kind"] == "reference" and r["cpu_baseline"]["requests"] == 50_000
        assert r["parity_vs_cpu_baseline"] is True
    assert c["cfg5"]["parity_vs_reference_fixture"] is True and c["cfg5"]["fixture_ticks"] == 200
    assert c["cfg5"]["steps"] == 1000 and c["cfg5"]["ms_per_step"] > 0
    assert c["cfg5"]["latency_samples"] >= 1000 and c["cfg5"]["parity_vs_cpu_baseline"] is True
    # the driver's line carries a short record of each, and its own baseline
    ln = j["_line"]
    assert set(ln["configs"]) == set(c) and "cpu_baseline" in ln and "roofline" in ln
    for k in ("cfg3", "cfg4", "cfg5"):
        assert ln["configs"][k]["cpu_baseline"]["value"] > 0 and ln["configs"][k]["roofline"]["frac"] > 0
    assert ln["value_end_to_end"] > 0 and ln["td_surface"]


@pytest.mark.gpu
def test_bench_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as the driver would type it, no torch.distributed.run around
    it: bench.py starts its two ranks itself; on the 1-GPU box they share device 0 and exchange
    over the mailbox transport. ONE line, n_gpus 2, one global batch through ydc_dispatch_sharded,
    the gathered placement equal to the oracle's — and BASELINE.json configs[3] (cfg4, strong)
    timed beside the weak figure."""
    env = {k: None for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    j = run_bench("--gpus", "2", "--steps", "3", "--warmup", "2", env=env, timeout=900)
    validate(j, steps=3, cpu_baseline=False)
    assert j["n_gpus"] == 2 and j["sharded"] is True and j["parity_vs_oracle"] is True
    assert j["transport"] in ("ipc", "ipc-device", "rccl"), j["transport_detail"]
    assert "200000 pending requests x 4000 servants" in j["config"]["workload"]
    s = j["strong_cfg4"]
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and s["sharded"] is True
    assert "4000000 pending requests x 16000 servants" in s["workload"]
    assert s["parity_vs_oracle"] is True and s["value"] > 0


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus N under a launcher that started another number of ranks: no line at all rather
    than a line whose n_gpus is not what was asked for (no GPU needed: refused before any)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], cwd=ROOT,
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 2 and "refusing" in out.stderr and "{" not in out.stdout


@pytest.mark.gpu
def test_bench_cfg5_streaming_line():
    j = run_bench("--config", "cfg5", "--steps", "40", "--warmup", "5")
    validate(j, steps=40, streaming=True)
    assert j["parity_ticks"] >= 20 and "hipGraph" in j["config"]["workload"]
    assert j["parity_vs_reference_fixture"] is True and j["fixture_ticks"] == 45
    e = j["tick_enqueued_eagerly"]  # (the same step without the graph replay, beside it)
    assert e["parity_vs_reference_fixture"] is True and e["ms_per_step"] > 0 and e["ticks"] >= 60


@pytest.mark.gpu
def test_bench_group_of_one_rank_line():
    """The N > 1 code path of bench.py with one rank and the RCCL-free transport: the line says
    whether the batch went through ydc_dispatch_sharded and over which transport."""
    j = run_bench("--steps", "3", "--warmup", "2", "--transport", "ipc", "--no-cpu-baseline",
                  env={"YDC_BENCH_FORCE_DIST": "1"})
    validate(j, steps=3, cpu_baseline=False)
    assert j["sharded"] is True and j["transport"] == "ipc" and j["parity_vs_oracle"] is True


def test_validator_accepts_the_committed_lines():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cfg2.json")))
    assert files, "no committed bench line"
    validate(json.load(open(files[-1])))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cfg5.json")))
    validate(json.load(open(files[-1])), streaming=True)


def test_bench_baseline_json_agrees():
    b = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "assignments/sec" in b["metric"]
    assert b.get("published") == {}  # nothing published for this path: vs_baseline stays null


def test_driver_line_of_a_full_record_stays_short():
    """bench.driver_line() on the largest record this repo ever produced (round 5's 23.8 KB line,
    which the driver could not parse): contract keys intact, roofline + cpu_baseline carried,
    one short record per configuration, well under the limit. No GPU needed."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_line.json")))
    assert len(json.dumps(full)) > 20000
    line = bench.driver_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT // 2, len(text)
    validate_line(line, cpu_baseline=True)
    for k in REQUIRED:
        assert line[k] == full[k]
    assert set(line["configs"]) == {"cfg3", "cfg4", "cfg5", "cfg2_150_digests"}
    assert line["td_surface"]["concurrent_calls_per_s"]["16"] > 0
