"""bench.py's printed line (CPU): the driver keeps the last ~10 kB of stdout and parses the LAST JSON line.  Round 5's
line nested every extra configuration inside the headline (21 kB) and did not parse.  The contract now: one short line
per extra configuration first, the headline LAST and well under 6 kB.  Canned input: the complete objects of a real
round-5 run (profiles/r05_bench_c3.log: headline + five extra configurations)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _canned():
    lines = [ln for ln in open(os.path.join(ROOT, "profiles", "r05_bench_c3.log")) if ln.startswith("{")]
    full = json.loads(lines[-1])
    extras = full.pop("extra_configs")
    return full, extras


def test_headline_is_short_complete_and_last_in_a_10kB_tail():
    b = _bench()
    full, extras = _canned()
    # (a sixth extra configuration, as the default run has since round 6)
    extras = dict(extras, C1m=extras["C1"])
    out = []
    for name, sub in extras.items():
        ln = json.dumps(b.slim_line(sub, extra=True))
        assert len(ln) < 1000, (name, len(ln))
        assert json.loads(ln)["config"]["name"] == name or name == "C1m"
        out.append(ln)
    head = json.dumps(b.slim_line(full))
    assert len(head) < 6000, len(head)
    out.append(head)
    stdout = "\n".join(out) + "\n"
    tail = stdout[-10000:]
    last = [ln for ln in tail.splitlines() if ln.strip()][-1]
    line = json.loads(last)  # the last line of a 10 kB tail is the whole headline
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "host_boundary"):
        assert key in line, key
    assert line["config"]["name"] == "C3" and "100000000 x d=128" in line["config"]["workload"]
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "ms_per_launch", "stage_ms_per_step"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert "note" not in json.dumps(line)  # prose lives in DESIGN.md, not in the line
    # every extra line is in the tail too, each parseable on its own
    parsed = [json.loads(ln) for ln in tail.splitlines() if ln.startswith("{")]
    assert len(parsed) >= len(extras)


def test_slim_line_keeps_the_multi_gpu_block_small():
    b = _bench()
    full, _ = _canned()
    ranks = [{"rank": r, "device": r, "collective_ms_per_step": 0.5, "collectives_per_step": 3, "filter_ms": 1.0,
              "stage_ms_per_step": {"coarse": 0.1, "filter": 1.0, "note": "x" * 500}, "scan_bytes_per_step": 1e11}
             for r in range(8)]
    full["multi_gpu"] = {"backend": "nccl", "world": 8, "coarse": "sharded by queries", "collectives_per_step": 3,
                         "ranks": ranks}
    ln = json.dumps(b.slim_line(full))
    assert len(ln) < 6000
    m = json.loads(ln)["multi_gpu"]
    assert m["world"] == 8 and len(m["ranks"]) == 8
    assert len(json.dumps(m)) < 1024 + 8 * 40
