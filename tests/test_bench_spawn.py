"""bench.py's launcher logic on CPU: `--gpus N` without a launcher starts N ranks itself; a launcher whose WORLD_SIZE
disagrees with --gpus is an error (a scaling run must never silently measure one GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def test_gpus_flag_spawns_that_many_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == [0, 1] and r["distinct_processes"] == 2
    # the N > 1 default is another workload than the N = 1 default: every line names the one-GPU rate of its own share
    assert r["default_workload"].startswith("C4") and "scaling_reference" in r["scaling_reference_doc"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count('out["scaling_reference"]') >= 2             # set on the bank (N >= 1) lines and on the default C2 line


def test_world_size_must_agree_with_gpus():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--spawn-check"],
                         env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "must agree" in out.stderr


def test_single_rank_spawn_check():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn-check"], env=_env(), capture_output=True, text=True, timeout=120)
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["n_gpus"] == 1 and r["ranks_seen"] == [0]


def test_require_rccl_refuses_the_gloo_exchange():
    """--require-rccl: a scaling line must come from the in-library RCCL all-reduce with all N ranks seen.  In gloo mode (ranks share
    devices, the exchange goes through torch.distributed) the flag makes the run exit non-zero and say why; without it the same
    command succeeds.  (The decision function is the one main_bank applies to the real line.)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check", "--dist-backend", "gloo"]
    ok = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    bad = subprocess.run(cmd + ["--require-rccl"], env=_env(), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "not the in-library RCCL all-reduce" in bad.stderr, bad.stderr[-2000:]
    sys.path.insert(0, ROOT)
    import bench
    assert bench.rccl_requirement_failure(bench.RCCL_COLLECTIVE, [0, 1, 2, 3], 4) is None
    assert "ranks seen" in bench.rccl_requirement_failure(bench.RCCL_COLLECTIVE, [0, 1, 3], 4)
    assert bench.rccl_requirement_failure("torch.distributed all_reduce (gloo) ...; in-library communicator failed: x", [0, 1], 2) is not None
    assert bench.rccl_requirement_failure("none (one shard)", [0], 1) is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'out["scaling_efficiency"]' in src and 'out["require_rccl"]' in src


def test_rccl_is_required_by_default_for_a_multi_gpu_line():
    """Round 6: `bench.py --gpus N` (the driver's command, no extra flag) requires the in-library RCCL exchange; --allow-gloo-exchange
    and an explicit --dist-backend gloo are the two ways out, and the printed line carries `collective` and `ranks_seen` at top level."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"]
    got = {}
    for name, extra in (("default", []), ("allow", ["--allow-gloo-exchange"]), ("gloo", ["--dist-backend", "gloo"])):
        out = subprocess.run(base + extra, env=_env(), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        got[name] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["require_rccl"]
    assert got == {"default": True, "allow": False, "gloo": False}
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"collective": collective,\n               "ranks_seen"' in src
