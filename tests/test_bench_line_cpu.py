"""bench.py's output contract (no GPU): the LAST stdout line is a compact JSON object the driver can parse out of an 8 KB
tail -- headline + roofline + cpu_baseline + a scalar digest of the extras -- and `python bench.py --gpus N` started bare
launches its own N ranks (round 4's line had grown to 36.6 KB and the driver recorded `parsed: null`; `--gpus 2` asserted)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r04final_bench.json")      # a real full result (36.6 KB as one line)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _full():
    return json.load(open(CANNED))


def test_compact_line_size_and_roundtrip():
    full = _full()
    assert len(json.dumps(full)) > 30000                     # the canned result is the one that broke the driver's parser
    line = bench.compact_line(full)
    assert "\n" not in line and len(line.encode()) < bench.COMPACT_LIMIT_BYTES <= 6000
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
        assert d[k] == full[k] or k in ("config", "roofline")
    for k in ROOFLINE:
        assert d["roofline"][k] == full["roofline"][k]
    for k in CPU:
        assert d["cpu_baseline"][k] == full["cpu_baseline"][k]
    assert d["config"]["workload"] == full["config"]["workload"] and "model" not in d["config"]
    # the digest: one scalar group per extra figure, every figure present
    ex = d["extra"]
    assert set(full["extra"]) <= set(ex) | {"reference_workloads"} and "reference_workloads" in ex
    assert ex["lotd_2p24_points"]["whole_step_frac"] == full["extra"]["lotd_2p24_points"]["whole_step_frac"]
    assert ex["reference_workloads"]["rows"] == 52 and ex["reference_workloads"]["faster"] == 52
    assert all(not isinstance(v, list) or len(v) <= 4 for g in ex.values() if isinstance(g, dict) for v in g.values())


def test_compact_line_never_outgrows_the_limit():
    full = _full()
    # a digest that explodes (an extra with thousands of keys in a group the digest copies) must cost digests, never the headline
    full["extra"]["march_composite"]["kernel_us_per_iter"] = {f"k{i}": float(i) for i in range(2000)}
    full["config"]["parallelism"] = "x" * 5000
    line = bench.compact_line(full)
    assert len(line.encode()) <= bench.COMPACT_LIMIT_BYTES
    d = json.loads(line)
    assert d["value"] == full["value"] and d["roofline"]["frac"] == full["roofline"]["frac"] and "cpu_baseline" in d
    assert "march_composite" not in d["extra"]


def test_extra_errors_survive_in_the_digest():
    full = _full()
    full["extra"]["c4_mixed_lotd"] = {"error": "RuntimeError('boom')" * 20}
    d = json.loads(bench.compact_line(full))
    assert d["extra"]["c4_mixed_lotd"]["error"].startswith("RuntimeError")


def test_emit_prints_full_tables_first_and_the_compact_line_last(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _full()
    bench.emit(full)
    lines = capsys.readouterr().out.splitlines()
    assert lines[-2].startswith("BENCH_FULL ") and json.loads(lines[-2][len("BENCH_FULL "):]) == full
    assert json.loads(lines[-1])["value"] == full["value"] and len(lines[-1]) < 6000
    assert json.load(open(tmp_path / "bench_extra.json")) == full


@pytest.mark.timeout(300)
def test_bare_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks start, meet on a gloo group and rank 0 prints the line
    (argument plumbing only -- no GPU here; the kernels of the N > 1 path are covered by test_dist_cpu.py / test_dist_gpu.py)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--plumbing-check"], capture_output=True, text=True, timeout=280, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] is None
    assert len(d["config"]["per_rank_ms_per_step"]) == 2 and "world size 2" in d["config"]["parallelism"]


def test_gpus_mismatch_is_an_error_not_an_assert():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-check"], capture_output=True,
                       text=True, timeout=120, env=env, cwd=str(ROOT))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "AssertionError" not in r.stderr
