"""bench.py's own N-rank launch (`python bench.py --gpus N` with no rendezvous in the environment): the parent starts N ranks through
torch.distributed.run before anything touches a GPU and relays rank 0's JSON line.  Rehearsed here on CPU with gloo and the
`--collective-only` step (the flat gradient buffer all-reduced; no kernels), so the launcher, the rendezvous on 127.0.0.1, the
barrier + max-over-ranks timing and the JSON relay are covered without a device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_gpus_flag_starts_that_many_ranks():
    r, out = run(["--gpus", "2", "--backend", "gloo", "--workload", "tiny", "--collective-only", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert out is not None and out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["collective_only"] and out["allreduce_ok"] and out["allreduce_bytes"] == 4 * 3143352
    assert out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak" and out["ms_per_step"] > 0


def test_single_rank_needs_no_launcher():
    r, out = run(["--gpus", "1", "--workload", "tiny", "--collective-only", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0 and out["n_gpus"] == 1 and out["allreduce_ok"]


def test_driver_launched_ranks_do_not_relaunch():
    """Under the driver's own `torch.distributed.run` the environment carries WORLD_SIZE: --gpus is then only a label."""
    r, out = run(["--gpus", "8", "--workload", "tiny", "--collective-only", "--steps", "1", "--warmup", "0"],
                 {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29417"})
    assert r.returncode == 0 and out["n_gpus"] == 1
