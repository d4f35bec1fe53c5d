"""bench.py's launch contract on the CPU: `python bench.py --gpus N` with no launcher around it spawns
the N ranks itself (child torchrun, rendezvous on 127.0.0.1), fails when WORLD_SIZE and --gpus disagree, verifies the
data-parallel gradients against the single-process gradients on the concatenated batch and reports per-rank times.
Runs with `--emulate-cpu` (gloo + the numpy emulator of the C ABI): launching / sharding / verification logic only --
the measured path on the MI355X is the same code over RCCL and the HIP library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--emulate-cpu", "--ngf", "8", "--size", "32", "--bs", "2", "--steps", "1", "--warmup", "1"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_bench_spawns_its_own_ranks_and_verifies_dp():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--verify-dp"] + SMALL,
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 4
    assert out["scaling"] == "weak" and out["value"] > 0 and len(out["ms_per_step_by_rank"]) == 2
    assert abs(out["ms_per_step"] - max(out["ms_per_step_by_rank"])) < 1e-2      # MAX over ranks
    dp = out["dp_verify"]
    assert dp["ok"] and dp["global_batch"] == 4 and dp["rel_l2_D"] < 1e-4 and dp["rel_l2_G"] < 1e-4
    assert out["config"]["workload"].startswith("custom")                          # not mislabelled as configs[1]


def test_bench_with_more_than_one_rank_checks_itself_by_default():
    """No flag: with two ranks the data-parallel gradient check runs on its own and the line carries it, the per-rank times of the same
    steps WITHOUT the collectives (this run's own one-GPU reference) and the per-rank exposed waits -- what a scaling record of the
    first real multi-GPU run has to show."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    dp = out["dp_verify"]
    assert dp["ok"] and dp["requested"].startswith("default") and dp["rel_l2_D"] < 1e-4 and dp["rel_l2_G"] < 1e-4
    assert len(out["ms_per_step_no_comm_by_rank"]) == 2 and all(v > 0 for v in out["ms_per_step_no_comm_by_rank"])
    assert len(out["comm_exposed_ms_per_step_by_rank"]) == 2 and "three gradient buckets" in out["comm"]
    # --no-verify-dp switches it off
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-verify-dp"] + SMALL,
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "dp_verify" not in json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])


def test_bench_refuses_a_world_size_that_is_not_the_request():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_mixed_resolution_with_four_ranks_on_different_buckets():
    """BASELINE.json configs[4]'s launch shape on the CPU: four gloo ranks, every rank-step draws one resolution bucket and the
    ranks START on different buckets (rank r on bucket r mod 3), so one step all-reduces gradients that come from different tile
    sizes (the parameters do not depend on the tile size).  --size 64 scales the 128/256/512 buckets to 32/64/128."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--mixed", "--emulate-cpu", "--ngf", "8",
                        "--size", "64", "--bs", "4", "--steps", "3", "--warmup", "3", "--blocks", "6", "--lambda-rs", "1"],
                       capture_output=True, text=True, env=_env(), timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["parallelism"] == "dp4" and out["config"]["global_batch"] == 16
    assert len(out["ms_per_step_by_rank"]) == 4 and out["value"] > 0 and out["collective_backend"] == "gloo"
    b = out["buckets_rank0"]
    assert [(x["tiles"], x["size"]) for x in b] == [(16, 32), (4, 64), (1, 128)]
    assert [x["steps"] for x in b] == [1, 1, 1], b          # three timed steps: rank 0 visited every bucket once
    assert "configs[4]" in out["config"]["workload"]


def test_bench_eight_ranks_mixed_resolution_verified():
    """The driver's 8-GPU launch shape rehearsed on the CPU (no 8-GPU node is available to the builder): eight gloo ranks, mixed-resolution
    buckets, the data-parallel gradients verified against the single-process gradients on the concatenated batch; every rank reports
    its step time and its exposed all-reduce time."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--mixed", "--emulate-cpu", "--ngf", "8",
                        "--size", "64", "--bs", "4", "--steps", "1", "--warmup", "1", "--blocks", "6"],
                       capture_output=True, text=True, env={**_env(), "OMP_NUM_THREADS": "1"}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8" and out["config"]["global_batch"] == 32
    assert len(out["ms_per_step_by_rank"]) == 8 and len(out["comm_exposed_ms_per_step_by_rank"]) == 8
    assert out["collective_backend"] == "gloo" and out["rccl_ranks"] == 0 and out["value"] > 0
    sch = out["bucket_schedule"]
    assert len(sch["first_bucket_by_rank"]) == 8 and sorted(set(sch["first_bucket_by_rank"])) == [0, 1, 2], sch        # the ranks start on different buckets
    assert sch["timed_steps_per_bucket_all_ranks"] in ([3, 3, 2], [3, 2, 3], [2, 3, 3]), sch                            # one timed step: eight draws over three buckets


def test_bench_mixed_resolution_load_balance_over_the_buckets():
    """configs[4]'s load balance: with a multiple of three timed steps every one of four ranks (started on different buckets) finishes the
    SAME number of steps per bucket, and every step mixes the three tile sizes over the ranks."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--mixed", "--emulate-cpu", "--ngf", "8",
                        "--size", "64", "--bs", "4", "--steps", "3", "--warmup", "3", "--blocks", "6"],
                       capture_output=True, text=True, env={**_env(), "OMP_NUM_THREADS": "1"}, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    sch = out["bucket_schedule"]
    assert len(sch["first_bucket_by_rank"]) == 4 and len(set(sch["first_bucket_by_rank"])) == 3, sch       # four ranks over three buckets: all three from step one
    assert sch["timed_steps_per_bucket_by_rank"] == [[1, 1, 1]] * 4, sch
    assert sch["timed_steps_per_bucket_all_ranks"] == [4, 4, 4], sch


def test_bench_eight_ranks_verify_dp():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--verify-dp", "--emulate-cpu", "--ngf", "8",
                        "--size", "32", "--bs", "1", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, env={**_env(), "OMP_NUM_THREADS": "1"}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    dp = out["dp_verify"]
    assert out["n_gpus"] == 8 and dp["ok"] and dp["global_batch"] == 8 and dp["rel_l2_D"] < 1e-4 and dp["rel_l2_G"] < 1e-4


def test_bench_single_process_line_has_the_scaling_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline"] + SMALL,
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and len(out["ms_per_step_by_rank"]) == 1 and out["comm_exposed_ms_per_step_by_rank"] == [0.0] and "rccl_ranks" in out
