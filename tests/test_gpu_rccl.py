"""GPU: first contact with RCCL on the hardware there is -- ONE GPU, so a process group of ONE rank with backend "nccl".

The sharded path's exchanges (shard.gather_to_root / gather_step_results / exchange_halo, comm.py's data channel) return
early for a lone rank; `force=True` / `RankCommunicateModule.loopback` make them issue the calls anyway, so communicator
creation, the collective on the caller's (side) stream with `record_stream`, owned receive buffers, grouped
send / receive to self and the stream hand-over back to the consumer all execute once before the driver's 8-GPU run does
it for the first time.  What this CANNOT show: xGMI transport, more than one peer, skew between ranks.

Runs in a child process under a hard timeout: a collective that never completes must not take the test session (or the
GPU lease) with it.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch
sys.path.insert(0, os.environ["DPMTEST_ROOT"])
import torch.distributed as dist
from deeppointmap_amd import shard
from deeppointmap_amd.comm import RankCommunicateModule

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
gen = torch.Generator().manual_seed(5)
desc = torch.randn(8, 131, 256, generator=gen).to(dev)
table = torch.randn(8, 56, generator=gen).to(dev)

# (1) the step's one collective, on a stream of its own as bench.py issues it (comm.wait_stream -> gather -> results used
#     on the caller's stream), three times over so that the communicator is reused, not only created
comm = torch.cuda.Stream(device=dev)
for it in range(3):
    d_in, t_in = desc + it, table - it           # produced on the caller's stream just before
    comm.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(comm):
        for t in (d_in, t_in):
            t.record_stream(comm)
        gd, gt = shard.gather_step_results(d_in.contiguous(), t_in, force=True)
        done = comm.record_event()
    torch.cuda.current_stream(dev).wait_event(done)
    assert gd.shape == d_in.shape and gt.shape == t_in.shape
    assert gd.data_ptr() != d_in.data_ptr()       # an owned receive buffer, not the input handed back
    assert torch.equal(gd, d_in) and torch.equal(gt, t_in), it
    del d_in, t_in                                 # the allocator may recycle them: record_stream covers the reader

# (2) plain gather of an odd-sized tensor
x = torch.arange(12345, device=dev, dtype=torch.float32)
assert torch.equal(shard.gather_to_root(x, force=True), x)

# (3) the block-boundary hand-over: grouped isend / irecv, here to the rank itself (rank + 1 mod 1)
last_desc, last_pcd = desc[-1], torch.randn(3, 65536, generator=gen).to(dev)
got_d, got_p = shard.exchange_halo(last_desc, last_pcd, force=True)
assert torch.equal(got_d, last_desc) and torch.equal(got_p, last_pcd)
got_d, got_p = shard.exchange_halo(last_desc, None, force=True)
assert torch.equal(got_d, last_desc) and got_p is None

# (4) comm.py: gloo control group + the default nccl group as data channel (the configuration agents / cloud use)
bus = RankCommunicateModule(control_group=dist.new_group(backend="gloo"), data_group=dist.group.WORLD, device=dev)
assert bus.data_is_nccl and bus.world == 1
bus.add_member(0)
for shape in ((131, 256), (3, 65536), (4, 4)):
    t = torch.randn(*shape, generator=gen).to(dev)
    back = bus.loopback(t)
    assert back.is_cuda and back.data_ptr() != t.data_ptr() and torch.equal(back, t)
host = torch.randn(6, 6, generator=gen)
back = bus.loopback(host)
assert not back.is_cuda and torch.equal(back, host)
bus.send_message(0, 0, "UPLOAD_SCAN", dict(key_points=desc[0]))      # a member talking to itself: the plain queue
cmd, msg = bus.fetch_message(0, block=False)
assert cmd == "UPLOAD_SCAN" and torch.equal(msg["key_points"], desc[0])
bus.close()

torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_call_paths_execute_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               HSA_ENABLE_IPC_MODE_LEGACY="0", DPMTEST_ROOT=ROOT)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])


def _bench(*extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if not k.startswith("DPM_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29300 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--force-collectives", "--frames", "8",
           "--points", "16384", "--steps", "3", "--warmup", "1", "--no-extras", "--cpu-frames", "0", *extra]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_bench_over_a_one_rank_rccl_group_records_its_ranks():
    """bench.py's N > 1 code path (process group with a timeout, rank census, the step's gather on the communication stream) on the
    one GPU there is: the line must say which ranks on which devices the process group saw."""
    import json
    out = _bench()
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["ranks"]["world_size"] == 1 and line["ranks"]["backend"] == "nccl" and len(line["ranks"]["members"]) == 1
    assert line["ranks"]["members"][0]["pci"] and "forced" in line["config"]["parallelism"]
    assert line["value"] > 0 and line["parity_gate"]["checked"] is False     # fixtures cover the 65 536-point workload only
    assert "timed region" in out.stderr and "[bench rank 0/1" in out.stderr  # breadcrumbs


def test_bench_ends_with_an_error_line_when_a_collective_never_completes():
    import json
    import time
    t0 = time.time()
    out = _bench("--inject-failure", "hang-in-gather", "--collective-timeout", "5")
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert out.returncode == 124 and time.time() - t0 < 300
    assert line["value"] is None and line["n_gpus"] == 1 and "exceeded its budget" in line["error"] and "timed region" in line["phase"]


def test_bench_ends_with_an_error_line_when_a_step_raises():
    import json
    out = _bench("--inject-failure", "raise-in-step")
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert out.returncode != 0 and line["value"] is None and "injected failure" in line["error"]
