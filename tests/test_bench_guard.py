"""bench.py's Guard: a phase that outlives its budget, an exception and a SIGTERM each end the process with ONE JSON line on rank 0
that carries "error" -- the first multi-GPU run happens under the driver, unobserved, and must not hang or die silently.
CPU only: the Guard is exercised without the workload (tests/test_gpu_rccl.py drives the real bench.py through it on a GPU)."""
import json
import os
import signal
import subprocess
import sys
import time

from conftest import ROOT

CHILD = r"""
import argparse, os, sys, time
sys.path.insert(0, os.environ["DPMTEST_ROOT"])
import bench
args = argparse.Namespace(steps=7, warmup=1)
g = bench.Guard(int(os.environ["RANK"]), 2, args)
mode = sys.argv[1]
if mode == "hang":
    g.phase("a collective that never completes", 0.7)
    time.sleep(60)
elif mode == "raise":
    g.phase("a step that raises", 30)
    try:
        raise RuntimeError("boom")
    except RuntimeError as e:
        g.fail(f"{type(e).__name__}: {e}")
elif mode == "term":
    g.phase("waiting for the launcher's SIGTERM", 30)
    print("READY", file=sys.stderr, flush=True)
    time.sleep(60)
elif mode == "ok":
    g.phase("a phase that completes", 5)
    g.finish()
    time.sleep(1.5)
    print("DONE")
"""


def _run(mode, rank=0, sigterm=False):
    env = dict(os.environ, DPMTEST_ROOT=ROOT, RANK=str(rank))
    p = subprocess.Popen([sys.executable, "-c", CHILD, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    if sigterm:
        assert "READY" in p.stderr.readline() + p.stderr.readline()
        p.send_signal(signal.SIGTERM)
    t0 = time.time()
    out, err = p.communicate(timeout=120)
    return p.returncode, out, err, time.time() - t0


def test_overrun_phase_ends_the_run_with_an_error_line():
    rc, out, err, secs = _run("hang")
    line = json.loads(out.strip().splitlines()[-1])
    assert rc == 124 and secs < 30
    assert line["value"] is None and line["n_gpus"] == 2 and line["steps"] == 7 and "exceeded its budget" in line["error"]
    assert line["phase"] == "a collective that never completes" and "[bench rank 0/2" in err


def test_exception_ends_the_run_with_an_error_line():
    rc, out, err, _ = _run("raise")
    line = json.loads(out.strip().splitlines()[-1])
    assert rc == 1 and "RuntimeError: boom" in line["error"] and line["metric"].startswith("LiDAR frames/s")


def test_other_ranks_report_on_stderr_only():
    rc, out, err, _ = _run("hang", rank=1)
    assert rc == 124 and out.strip() == "" and "[bench rank 1/2] ERROR" in err


def test_sigterm_from_the_launcher_leaves_a_record():
    rc, out, err, _ = _run("term", sigterm=True)
    assert rc == 143 and "SIGTERM" in json.loads(out.strip().splitlines()[-1])["error"]


def test_finished_guard_stays_quiet():
    rc, out, err, _ = _run("ok")
    assert rc == 0 and out.strip() == "DONE"


def test_parity_gate_accepts_the_reference_rows_and_refuses_perturbed_ones():
    """bench.py::parity_gate on rows built FROM the fixtures it checks against (the gate's own sensitivity, on the CPU): the
    reference's values pass; a pose 2e-4 m off, a key point one ulp off, a changed inlier count, an asymmetric information matrix
    each fail; another workload is reported as unchecked."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    gold = os.path.join(ROOT, "tests", "golden")
    poses, fps, enc = (np.load(os.path.join(gold, f)) for f in ("poses_full.npz", "fps.npz", "encoder_full.npz"))
    desc = torch.zeros(8, 131, 256)
    desc[0, 128:131] = torch.from_numpy(enc["synthetic0.coor"]) * 60.0
    for f in (0, 1):
        desc[f, :128] = torch.from_numpy(enc[f"synthetic{f}.fea"])
    table = torch.zeros(8, 56)
    table[:, 20:56] = torch.eye(6).reshape(36)
    for f in range(1, 6):
        k = f"pair{f - 1}_{f}"
        table[f, 0:9] = torch.from_numpy(poses[k + ".R"]).reshape(9)
        table[f, 9:12] = torch.from_numpy(poses[k + ".T"]).reshape(3)
        table[f, 12], table[f, 14] = float(poses[k + ".rmse"]), float(poses[k + ".n_conf"])
    g = bench.parity_gate(desc, table, 65536)
    assert g["checked"] and g["ok"] and g["pairs"] == 5 and g["fps_prefix_equal"] and g["max_dT_m"] == 0.0, g

    def broken(edit):
        d, t = desc.clone(), table.clone()
        edit(d, t)
        return bench.parity_gate(d, t, 65536)
    assert not broken(lambda d, t: t[3, 9].add_(2e-4))["ok"]                                           # a pose off by 0.2 mm
    assert not broken(lambda d, t: d[0, 128, 7].copy_(torch.nextafter(d[0, 128, 7], torch.tensor(9e9))))["ok"]   # a key point, one ulp
    assert not broken(lambda d, t: t[2, 14].add_(1))["ok"]                                             # inlier count
    assert not broken(lambda d, t: t[5, 21].add_(0.5))["ok"]                                           # information matrix asymmetric
    assert not broken(lambda d, t: d[1, 5, 5].add_(1e-3))["ok"]                                        # a descriptor feature
    assert bench.parity_gate(desc, table, 16384) == {"checked": False, "why": "fixtures cover the 65 536-point synthetic sequence, frames 0-5"}


def test_self_launch_command_line():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run: one process per GPU of this
    node, rendezvous on the loopback address, its own arguments passed through."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29777)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                   "--master-port", "29777", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    auto = bench.self_launch_command([], 2)
    assert 1024 < int(auto[auto.index("--master-port") + 1]) < 65536


def test_bench_without_a_launcher_starts_its_ranks():
    """On this GPU-less container the two self-launched ranks end at once with "needs a GPU" -- as rank 0's ONE error line with
    n_gpus = 2, which shows the re-execution happened and the guard's record survives it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        return   # on a GPU box the GPU suite covers the self-launch (test_gpu_multirank.py)
    assert out.returncode != 0
    assert "no launcher in the environment" in out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    # (rank 0 says "needs a GPU" itself, or -- when rank 1 got there first and the launcher's SIGTERM reached rank 0 while it was still
    # importing torch -- bench.py's start-up handler says so)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["value"] is None
    assert "needs a GPU" in lines[0]["error"] or "before the benchmark had started" in lines[0]["error"]


def test_sigterm_during_start_up_leaves_a_record():
    """bench.py installs a SIGTERM handler before it imports torch: a launcher that lost another rank during start-up still gets rank
    0's one error line."""
    bench_py = os.path.join(ROOT, "bench.py")
    # the part of bench.py that runs before `import torch`, as the main program
    code = ("import sys, time; head = open(%r).read().split('import torch  # noqa: E402')[0]; "
            "exec(compile(head, %r, 'exec'), {'__name__': '__main__', '__file__': %r}); "
            "print('READY', file=sys.stderr, flush=True); time.sleep(60)") % (bench_py, bench_py, bench_py)
    env = dict(os.environ, RANK="0", WORLD_SIZE="4")
    p = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    assert "READY" in p.stderr.readline()
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=60)
    assert p.returncode == 143
    line = json.loads(out.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 4 and "before the benchmark had started" in line["error"]
