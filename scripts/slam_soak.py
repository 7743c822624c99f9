#!/usr/bin/env python3
"""Soak run of deeppointmap_amd.system.SlamSystem: N synthetic scans through the multi-thread mode (or step) with loosened
thresholds so that key-frames, scan-to-map refinements, loop closures and optimiser runs all occur with procedural weights.
Reports scans/s, the back end's counters, device memory before / after, and checks that every pose is finite.
usage: slam_soak.py [frames=400] [points=16384] [mt|step]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from deeppointmap_amd import synthetic
from deeppointmap_amd.config import Cfg, default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.system import SlamSystem
from deeppointmap_amd.weights import init_procedural

F = int(sys.argv[1]) if len(sys.argv) > 1 else 400
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
mode = sys.argv[3] if len(sys.argv) > 3 else "mt"
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
args = default_args()
args.device = "cuda:0"
args.slam_system = Cfg(dict(edge_confidence_drop=0.0, edge_rmse_drop=1e9, key_frame_distance=3.0, loop_detection_rotation_min=0.0,
                            loop_detection_translation_min=0.0, loop_detection_transaction_gap=0.0,
                            loop_detection_prob_acpt_threshold=0.5, loop_detection_confidence_acpt_threshold=0.0))
slam = SlamSystem(args, init_procedural(Encoder(args)).to(dev), init_procedural(Decoder(args)).to(dev), device=dev)
base = synthetic.base_cloud(N)
scans = [synthetic.frame(f, N, base).unsqueeze(0) for f in range(min(F, 64))]
pad = torch.zeros(1, N, dtype=torch.bool)
data = lambda f: [scans[f % len(scans)], torch.eye(3).unsqueeze(0), torch.zeros(1, 3, 1), pad, None]
for f in range(4):
    slam.step(data(f))          # warm
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated(dev)
t0 = time.perf_counter()
if mode == "mt":
    slam.MT_Init()
    for f in range(4, F):
        slam.MT_Step(data(f))
    slam.MT_Done()
    slam.MT_Wait()
else:
    for f in range(4, F):
        slam.step(data(f))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
toks, poses = slam.trajectory()
assert torch.isfinite(poses).all(), "non-finite pose"
print(f"{mode}: {F - 4} scans of {N} points in {dt:.2f} s = {(F - 4) / dt:.1f} scans/s; graph {len(toks)} scans, stats {slam.backend.stats}")
print(f"device memory {m0 / 2**20:.0f} -> {torch.cuda.memory_allocated(dev) / 2**20:.0f} MiB (key-frame descriptors + full clouds stay resident)")
codes = {}
for c in slam.codes:
    codes[c.name] = codes.get(c.name, 0) + 1
print("exit codes", codes)
