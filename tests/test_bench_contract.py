"""CPU suite: bench.py's driver contract without a GPU (`--dry-run`: every step is a short sleep).

`python bench.py --gpus N` must create its own N ranks when no launcher did (the driver calls it both ways), rank 0
prints exactly one JSON line with n_gpus = N, and the step loop works against the wall budget: at least one step, at most
--steps, never past JB_BENCH_BUDGET_S."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "JB_BENCH_T0", "JB_BENCH_BUDGET_S"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_single_process_line_and_budget():
    out = _run(["--dry-run", "--steps", "3", "--warmup", "5"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["steps_requested"] == 3 and out["warmup_requested"] == 5
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    # a budget that is already spent: exactly one step is timed, the line still appears
    out = _run(["--dry-run", "--steps", "20", "--warmup", "5"], env=dict(JB_BENCH_BUDGET_S="0"))
    assert out["steps"] == 1 and out["steps_requested"] == 20
    # ... and the warm-up pass is never dropped, whatever the budget (round 4: every builder-side line had skipped it under a short
    # budget, i.e. timed the first job of a fresh process, and hid what the driver's later steps paid)
    assert out["warmup"] == 1 and out["warmup_requested"] == 5
    assert _run(["--dry-run", "--steps", "1"])["warmup"] == 0
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    main_src = ast.get_source_segment(src, next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "main"))
    assert "if a.warmup > 0:" in main_src and "budget_left() > 600" not in main_src


def test_gpus_2_spawns_its_own_ranks():
    """No outer torchrun: bench.py re-launches itself as 2 ranks (gloo here), both run the same number of steps, rank 0
    reports n_gpus = 2 and the max-over-ranks time (rank 1 sleeps twice as long per step)."""
    out = _run(["--gpus", "2", "--dry-run", "--steps", "4"])
    assert out["n_gpus"] == 2 and out["steps"] == 4
    assert out["dist"]["world_size"] == 2 and out["dist"]["backend"] == "gloo"
    assert out["ms_per_step"] >= 38.0            # rank 1: 40 ms per step
    assert out["config"]["parallelism"] == "sample-sharded x2"


def test_gpus_8_shards_config4_as_16_samples_per_rank():
    """BASELINE config 4 on one node: --gpus 8 -> 128 samples, 16 per rank (the sampler's shard_range), every rank reports its
    device, its samples and its own step seconds; the line carries the spread between the fastest and the slowest rank."""
    out = _run(["--gpus", "8", "--dry-run", "--steps", "2"], timeout=600)
    assert out["n_gpus"] == 8 and out["config"]["n_samples"] == 128 and out["config"]["samples_per_gpu"] == 16
    ranks = out["dist"]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8)) and all(r["n_samples"] == 16 for r in ranks)
    assert all(len(r["step_seconds"]) == 2 for r in ranks)
    lo, hi = out["dist"]["rank_step_seconds_min_max"]
    assert lo < hi and hi >= 0.15                # rank 7 sleeps 8 x 20 ms per step
    assert out["dist"]["backend"] == "gloo" and out["dist"]["launcher"] == "torch.distributed.run"


def test_outer_launcher_is_respected():
    """Launched the driver's way (torch.distributed.run outside): no second level of spawning."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "JB_BENCH_T0", "JB_BENCH_BUDGET_S"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "2"], capture_output=True, text=True, env=e, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
