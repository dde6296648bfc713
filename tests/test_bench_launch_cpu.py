"""The launch path of `bench.py --gpus N`: with no WORLD_SIZE in the environment the script starts the N
ranks itself, before anything touches the GPU; under a launcher it runs as the rank it is told to be.  `--dry-launch` makes
every rank print its rendezvous environment and exit, so the path is testable on a host without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [(d["rank"], d["local_rank"], d["world"]) for d in lines] == [(0, 0, 2), (1, 1, 2)]
    assert all(d["self_launched"] and d["master"].startswith("127.0.0.1:") for d in lines)
    assert lines[0]["master"] == lines[1]["master"]


def test_launcher_environment_wins():
    # the driver's documented N > 1 form: torch.distributed.run sets WORLD_SIZE -- no second level of children
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=120,
                       env=_env(RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511"))
    assert r.returncode == 0, r.stderr
    d = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(d) == 1 and (d[0]["rank"], d[0]["world"], d[0]["self_launched"]) == (1, 2, False)


def test_gpus_1_does_not_spawn():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=120)
    d = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(d) == 1 and d[0]["world"] == 1 and not d[0]["self_launched"]


def test_failing_rank_is_reported():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--dry-launch"], env=_env(GPCORE_BENCH_DRY_FAIL_RANK="2"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 2


def test_parent_never_touches_the_gpu():
    # everything above the self-launch in main() must be free of torch / libgpcore imports: check the module's import-time set
    code = ("import sys; sys.argv=['bench.py','--gpus','2','--dry-launch']; import runpy\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code, e.code\n"
            "assert 'torch' not in sys.modules and 'gp_algos_amd._lib' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)[:5]\n" % BENCH)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
