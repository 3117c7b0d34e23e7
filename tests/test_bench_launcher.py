"""bench.py started plainly with --gpus N > 1 launches its own N ranks (a child torch.distributed.run, never an exec) and
forwards rank 0's JSON line and the exit status.  CPU: the command line and the forwarding; GPU: two ranks on one MI355X."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_command_forwards_every_argument():
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "3", "--total-rays", "1048576", "--record", "stride:16", "--backend", "gloo"]
    cmd = bench.launcher_command(argv, 4, 29999, python="PY")
    assert cmd[:3] == ["PY", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                      # the ranks get the caller's arguments, in order, untouched
    assert bench.free_port() > 0


def test_run_ranks_forwards_rank0s_line_and_the_exit_status(monkeypatch, capsys, tmp_path):
    """run_ranks with the child replaced by a stand-in that behaves like N ranks: chatter on stdout and stderr, ONE JSON line
    from "rank 0", an exit status -- the parent's stdout must carry exactly that line and its return value the status."""
    child = tmp_path / "child.py"
    child.write_text('import sys\n'
                     'print("rank 1: hello")\n'
                     'print(\'{"metric": "m", "value": 1.5, "n_gpus": 2}\')\n'
                     'print("trailing chatter", file=sys.stderr)\n'
                     'sys.exit(int(sys.argv[1]))\n')
    for status in (0, 3):
        monkeypatch.setattr(bench, "launcher_command", lambda argv, gpus, port, python=None, s=status: [sys.executable, str(child), str(s)])
        rc = bench.run_ranks(["--gpus", "2"], 2)
        out = capsys.readouterr()
        assert rc == status
        assert out.out.strip().splitlines() == ['{"metric": "m", "value": 1.5, "n_gpus": 2}']
        assert "rank 1: hello" in out.err                      # everything else the ranks wrote goes to stderr
    # ranks that exit 0 without a line are a failure of the run, not a silent success
    child.write_text('print("no line")\n')
    monkeypatch.setattr(bench, "launcher_command", lambda argv, gpus, port, python=None: [sys.executable, str(child)])
    assert bench.run_ranks([], 2) != 0


def _env_without_rank_variables():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


@pytest.mark.timeout(300)
def test_plain_start_with_two_gpus_spawns_two_ranks_and_forwards_their_status():
    """No GPU here: both ranks stop at "no HIP device" -- which shows that the plain command reached two ranks (it used to
    exit with "launch with torch.distributed.run" before starting anything) and that their failure is the parent's status."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side test of the launcher (the GPU box runs the real thing below)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--cpu-seconds", "0"],
                       capture_output=True, text=True, env=_env_without_rank_variables(), timeout=280)
    assert r.returncode != 0
    assert r.stderr.count("no HIP device") >= 2, r.stderr[-2000:]
    assert "launch with torch.distributed.run" not in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_plain_start_with_two_gpus_runs_two_ranks_on_the_gpu_box():
    """`python3 bench.py --gpus 2 ...` as a plain subprocess: rc 0 and ONE JSON line with n_gpus 2 (both ranks share the box's
    one MI355X; gloo carries the collectives because RCCL wants one GPU per rank)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--all-on-device", "0",
                        "--steps", "2", "--cpu-seconds", "0", "--gather-rows", "64", "--record", "stride:16"],
                       capture_output=True, text=True, env=_env_without_rank_variables(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    # the default N > 1 line is the north-star run: the STRONG split of the 1 048 576-ray fan
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong"
    assert out["config"]["rays_total"] == 1048576 and out["config"]["rays_rank0"] == 524288
    g = out["config"]["dist"]["gather"]
    assert g["world"] == 2 and g["rays"] == 1048576 and "error" not in g
    assert g["trajectory"]["shape"] == [3, 6, 1048576]          # every 64th of 192 kept rows, both ranks' rays in ray order
    assert out["parity_check"]["ok"]
    # ... with the weak configuration (1 048 576 rays per GPU) as the secondary record of the same line
    w = out["weak"]
    assert "error" not in w, w
    assert w["scaling"] == "weak" and w["rays_total"] == 2 * 1048576 and w["value"] > 0 and w["steps"] == 2
    assert out["config"]["auto_exploration"]["kept"] in ("sliced", "plain", None)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rays_per_gpu_asks_for_the_weak_line():
    """--rays R: R rays per GPU (weak scaling) is the line's value, no secondary record."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--all-on-device", "0",
                        "--rays", "65536", "--steps", "2", "--cpu-seconds", "0", "--gather-rows", "256"],
                       capture_output=True, text=True, env=_env_without_rank_variables(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["rays_total"] == 2 * 65536 and "weak" not in out
    g = out["config"]["dist"]["gather"]
    assert g["trajectory"]["shape"] == [12, 6, 2 * 65536]          # every 256th of 3 072 rows, both ranks' rays in ray order
    assert out["parity_check"]["ok"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_plain_start_with_two_real_gpus_over_rccl():
    """`python3 bench.py --gpus 2` exactly as the driver types it, on a host with two GPUs: one rank per GPU, RCCL (nccl) for the
    timing reductions and the device-to-device read-back gather.  A one-GPU box skips it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--cpu-seconds", "0", "--record", "stride:16"],
                       capture_output=True, text=True, env=_env_without_rank_variables(), timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["rays_total"] == 1048576
    g = out["config"]["dist"]
    assert g["backend"] == "nccl" and g["gather"]["world"] == 2 and "error" not in g["gather"]
    assert out["parity_check"]["ok"] and "error" not in out["weak"]
