"""`python bench.py --gpus N` (the driver's plain form, no torch.distributed.run) must start its N ranks itself, before
any GPU call, and rank 0 must print ONE self-describing JSON line.

CPU part (`-m "not gpu"`): `--launch-check` starts the ranks, joins them over gloo and all-reduces a one per rank -- no
kernel runs and the line carries no number.  GPU part: the real bench with 2 ranks sharing the one card, collectives
staged through gloo (RIHIP_DIST_BACKEND=gloo), tiny global batch, parsed like the driver parses it."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=str(ROOT))
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_plain_gpus2_form_spawns_two_ranks_and_prints_one_line():
    p, lines = _run(["--gpus", "2", "--launch-check"], timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["launch_check"] is True and line["n_gpus"] == 2 and line["dist_ranks"] == 2
    assert "10M users x 1M items" in line["config"]["workload"]      # cfg3 tables for N < 8


def test_launch_check_names_cfg4_tables_at_eight_ranks():
    p, lines = _run(["--gpus", "8", "--launch-check"], timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["dist_ranks"] == 8 and "100M users x 10M items" in line["config"]["workload"]


def test_launcher_env_wins_and_single_rank_needs_no_group():
    p, lines = _run(["--gpus", "1", "--launch-check"])
    assert p.returncode == 0 and json.loads(lines[0])["dist_ranks"] == 1


def test_a_failing_rank_stops_the_job():
    # unknown flag: every rank's argparse exits 2; the launcher must return non-zero instead of hanging
    p, _ = _run(["--gpus", "2", "--launch-check", "--no-such-flag"], timeout=120)
    assert p.returncode != 0


@pytest.mark.gpu
def test_bench_gpus2_gloo_rehearsal_on_one_card():
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--global-batch", "4096", "--no-secondary",
                     "--no-cpu-baseline"], env_extra={"RIHIP_DIST_BACKEND": "gloo"}, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["metric"] == "bpr_pairs_per_sec" and line["n_gpus"] == 2 and line["dist_ranks"] == 2
    assert line["dist_backend"] == "gloo" and line["rccl_ranks"] is None
    assert line["value"] > 0 and line["config"]["per_gpu_batch"] == 2048 and line["config"]["config"] == "cfg3"
    assert line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_single_gpu_line_has_the_contract_keys():
    p, lines = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "rccl_ranks"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["steps"] == 2
