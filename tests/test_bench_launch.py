"""`python3 bench.py --gpus N` started plainly (the driver's command, no torchrun around it) must start its own N
ranks -- fresh child processes, before this process has made any GPU call -- relay rank 0's JSON line and exit code,
and print a JSON line of its own when the ranks cannot be started (BASELINE.json `metric`: "at 1/2/4/8")."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launcher_imports_neither_torch_nor_the_product():
    """the parent must be unable to touch the GPU: importing bench.py loads no torch and no product binding"""
    code = ("import sys, importlib.util as u; s = u.spec_from_file_location('b', %r); m = u.module_from_spec(s); "
            "s.loader.exec_module(m); print(sorted(k for k in sys.modules if k.split('.')[0] in ('torch', 'gpu_lib', 'bnn')))" % BENCH)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.strip()
    assert out == "[]", out


def test_rank_command_lines():
    b = _bench_module()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmds = b.rank_commands(8, argv, 29999, python="/usr/bin/python3", script="/x/bench.py")
    assert len(cmds) == 8
    for r, (cmd, env) in enumerate(cmds):
        assert cmd == ["/usr/bin/python3", "-u", "/x/bench.py"] + argv        # the same script, the same arguments
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "8"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29999"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                        # dmabuf IPC: RCCL needs it on this pool


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="checks the no-GPU answer; this machine has one")
def test_plain_multi_gpu_start_without_gpus_prints_a_json_error():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr
    d = json.loads(lines[0])
    assert "error" in d and d["n_gpus_visible"] == 0 and d["n_gpus"] == 2 and d["value"] is None


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="checks the no-GPU answer; this machine has one")
def test_a_rank_that_dies_is_reported_as_json(monkeypatch, capsys):
    """ranks that exit non-zero before printing the line: the launcher prints the error line and returns their code"""
    b = _bench_module()
    monkeypatch.setattr(b, "visible_gpus", lambda: 2)      # pretend two devices: the ranks then fail on their own (no GPU here)
    a = b.parse_args(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    rc = b.self_launch(a, ["--gpus", "2", "--steps", "1", "--warmup", "0"])
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert rc != 0 and len(out) == 1
    assert "error" in json.loads(out[0])


@pytest.mark.gpu
def test_plain_start_of_a_two_rank_rehearsal_runs_end_to_end():
    """`python3 bench.py --gpus 2 --rehearse-gloo` from the plain command on the one-GPU box: two fresh rank
    processes (gloo, both on cuda:0), one broadcast, shards, the line's self-check (CRC of the parameters on every
    rank, first 2048 classes of every rank against the CPU restatement)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-gloo", "--steps", "3", "--warmup", "1", "--batch", "16384"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["multi_gpu"]["ranks"] == 2
    assert d["multi_gpu"]["params_identical_on_all_ranks"] is True
    assert d["multi_gpu"]["first_2048_classes_of_every_rank_equal_oracle"] == [True, True]
    assert d["multi_gpu"]["collectives_on_the_data_path"] == 0


def test_ranks_that_never_finish_are_killed_and_reported(monkeypatch, capsys):
    """a rendezvous that hangs: after --launch-timeout the launcher kills exactly the processes it started and prints
    the JSON error line"""
    import time
    b = _bench_module()
    monkeypatch.setattr(b, "visible_gpus", lambda: 2)
    monkeypatch.setattr(b, "rank_commands", lambda n, argv, port, python=None, script=None:
                        [([sys.executable, "-c", "import time; time.sleep(120)"], {}) for _ in range(n)])
    a = b.parse_args(["--gpus", "2", "--launch-timeout", "2"])
    t0 = time.time()
    rc = b.self_launch(a, ["--gpus", "2"])
    assert rc == 124 and time.time() - t0 < 30
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(out) == 1 and "did not finish" in json.loads(out[0])["error"]
