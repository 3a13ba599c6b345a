"""The bench line committed under profiles/ (produced by `python bench.py` on an MI355X) carries every field the
driver's contract names; bench.py itself parses and refuses to run without a GPU (no CPU path)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def test_committed_bench_line_has_the_contract_fields():
    d = _last_json(os.path.join(ROOT, "profiles", "r02_bench_default.json"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"].replace("×", "x") or d["metric"] == base["metric"]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["value"] > 10 * c["value"]          # north_star: >= 10x the CPU path at one GPU


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""     # no JSON line, no CPU fallback


def _run_bench(argv, env_extra, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                          timeout=timeout, env=env)


def test_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two fresh workers itself: rehearsed
    over gloo with the empty-step workload, which makes no GPU call."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "launch-check"],
                   {"SVOH_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                          # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_in_collective"] == 2 and d["backend"] == "gloo"
    assert d["units_total"] == 3 * 1000 + 3 * 1001            # SUM over both ranks' units
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"


def test_gpus_flag_must_agree_with_world_size():
    """Under a launcher (WORLD_SIZE set) a different --gpus is an error, not a silently different n_gpus."""
    r = _run_bench(["--gpus", "4", "--workload", "launch-check"],
                   {"SVOH_BENCH_BACKEND": "gloo", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""
    r = _run_bench(["--gpus", "1", "--steps", "2", "--workload", "launch-check"], {"SVOH_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_every_workload_choice_has_a_bench_function():
    """`--workload stereo` was once an accepted choice that nothing dispatched."""
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    choices = re.search(r'"--workload".*?choices=\[(.*?)\]', src, re.S).group(1)
    choices = set(re.findall(r'"([a-z0-9-]+)"', choices))
    table = re.search(r'out = \{(.*?)\}\[args.workload\]', src, re.S).group(1)
    dispatched = set(re.findall(r'"([a-z0-9-]+)":', table)) | {"align"}
    assert choices == dispatched, (choices ^ dispatched)
