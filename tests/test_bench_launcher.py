"""bench.py --gpus N must start N ranks: the launcher, the barriers, the record gather and the JSON line are run
here with stand-in stages on the CPU over gloo (--standin is test infrastructure; the line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                          env=env, cwd=ROOT)


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out                                     # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_2_without_a_parent_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "4", "--standin"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["standin"] is True
    assert d["gathered_rows"] == 8 and d["gathered_ranks"] == [0, 1]          # every rank's records reached rank 0
    assert d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 4 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-2     # whole-job aggregate over both ranks


def test_driver_command_form_under_torch_distributed_run():
    """The driver's form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--frames", "3", "--standin"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["gathered_rows"] == 6


def test_single_rank_default_and_world_mismatch():
    d = _line(_run(["--steps", "2", "--warmup", "0", "--frames", "2", "--standin"]).stdout)
    assert d["n_gpus"] == 1 and d["gathered_rows"] == 2
    r = _run(["--gpus", "4", "--standin"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
