"""bench.py's own multi-rank path, rehearsed without a GPU: `--dry-run-cpu` goes through the SAME code as
the 8-GPU driver run - spawn_ranks -> RANK/WORLD_SIZE/MASTER_* environment -> init_process_group (gloo instead of RCCL) ->
shard.sharded_sample (broadcast of conditioning, per-rank shard, all-gather of mels) -> MAX-reduce of the timing ->
one JSON line on rank 0 - with the package's explicit torch backend on a tiny denoiser.  Also covered: a dying rank
takes its siblings down within seconds (no waiting for a collective time-out), and the torchrun-style launch the driver
uses (ranks started by a launcher, bench.py reads the environment)."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, "no JSON line in:\n" + text[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("world", [2, 8])
def test_dry_run_rank_path(world):
    p = subprocess.run([sys.executable, BENCH, "--dry-run-cpu", "--gpus", str(world), "--steps", "2", "--warmup", "1"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["dry_run_cpu"] is True
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["rccl_world_size"] == world
    assert d["config"]["global_batch"] == world * 2            # weak scaling: 2 utterances per rank in the dry run
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    pr = d["per_rank_ms_per_step"]
    assert len(pr["ranks"]) == world and pr["min"] <= pr["max"]
    assert abs(d["ms_per_step"] - pr["max"]) < 1e-6           # the job's time is the slowest rank's
    # value = units ALL ranks processed / that time
    assert d["value"] == pytest.approx(world * 2 * 24 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)
    assert "cpu_baseline" not in d and "roofline" not in d     # a rehearsal, not a measurement


def test_rank_failure_stops_the_job_quickly():
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--dry-run-cpu", "--gpus", "4", "--steps", "1", "--warmup", "1",
                        "--dist-timeout", "120"],
                       env=_clean_env(DVITS_BENCH_FAIL_RANK="2"), capture_output=True, text=True, timeout=300)
    dt = time.time() - t0
    assert p.returncode != 0
    assert "stopping the other" in p.stderr
    assert dt < 60, "a dead rank must not leave the others waiting for the collective time-out (took %.0f s)" % dt
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_launcher_style_ranks():
    """The driver's form: N processes started by a launcher with RANK / WORLD_SIZE / MASTER_* set."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = _clean_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--dry-run-cpu", "--gpus", str(world), "--steps", "1", "--warmup", "1"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    d = _last_json(outs[0][0])
    assert d["n_gpus"] == world and d["config"]["global_batch"] == 2 * world
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # only rank 0 prints


def test_world_size_mismatch_is_refused():
    p = subprocess.run([sys.executable, BENCH, "--dry-run-cpu", "--gpus", "2"],
                       env=_clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr
