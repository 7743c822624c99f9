"""debug: one registration call of tests/golden/slam_trace.npz, HIP path vs oracle, stage by stage.
usage: python scripts/debug/trace_call.py 69"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import T, load_golden, rot_angle
from oracle import dpm_oracle as O
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.params import decoder_shapes
from deeppointmap_amd.weights import init_procedural, procedural_state_dict
from test_slam_trace import rebuild
torch.set_grad_enabled(False)
cfg = default_args(); sd = procedural_state_dict(decoder_shapes(cfg))
g = load_golden("slam_trace.npz")
desc = {int(t): T(g["desc"][i]) for i, t in enumerate(g["desc_tokens"])}
k = int(sys.argv[1])
src, dst = rebuild(g, desc, k, "src"), rebuild(g, desc, k, "dst")
tr = {}
R, T_, conf, rmse = O.registration_forward(sd, cfg, src, dst, float(g[f"c{k}.num_sample"]), trace=tr)
dec = init_procedural(Decoder(cfg)).to("cuda:0")
tg = {}
Rg, Tg, cg, rg = dec.registration_forward(src, dst, num_sample=float(g[f"c{k}.num_sample"]), trace=tg)
print("pose: dT", float((Tg.cpu() - T_).norm()), "dR", rot_angle(Rg.cpu(), R), "n_conf", cg.numel(), conf.numel(), "rmse", rg, float(rmse))
print("x", float((tg["x"].cpu().view(-1, 256) - tr["x"].view(-1, 256)).abs().max()), "y", float((tg["y"].cpu().view(-1, 256) - tr["y"].view(-1, 256)).abs().max()))
si, di = tg["src_index"].cpu().flatten().long(), tg["dst_index"].cpu().flatten().long()
print("pairs equal:", bool((si == tr["src_index"]).all() and (di == tr["dst_index"]).all()), "n", si.numel())
if not bool((si == tr["src_index"]).all() and (di == tr["dst_index"]).all()):
    a = set(zip(si.tolist(), di.tolist())); b = set(zip(tr["src_index"].tolist(), tr["dst_index"].tolist()))
    print(" only gpu:", sorted(a - b)[:10], " only oracle:", sorted(b - a)[:10])
print("conf max diff", float((tg["conf"].cpu().flatten() - tr["conf"]).abs().max()), "min conf", float(tr["conf"].min()), "second min", float(tr["conf"].sort().values[1]))
print("n_corr", tg.get("n_corr"), tr["src"].shape[1], "iterations", tg.get("iterations"), "oracle inliers", int(tr["inlier"].sum()))
off = tg["offsets"].cpu().view(-1, 3)
n2 = (off ** 2).sum(1)
print("offset norms^2 nearest to the 4.0 cut:", (n2 - 4.0).abs().sort().values[:4].tolist())
