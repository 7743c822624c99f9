"""The three-way table of tests/test_gpu_margin.py as markdown: HIP against the reference in fp64, next to the reference's own fp32
evaluations (tests/golden/margin.npz).  Honours the DPM_* host knobs (deeppointmap_amd/knobs.py) for A/B reading."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import margin_cases
from deeppointmap_amd import knobs
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.weights import init_procedural
print("knobs:", knobs.apply_env())
torch.set_grad_enabled(False)
dec = init_procedural(Decoder(default_args())).to("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "margin.npz"))


def ang(A, B):
    M = torch.as_tensor(A).double().T @ torch.as_tensor(B).double()
    return float(np.arctan2(float(torch.linalg.norm(M - M.T)) / (2 * 2 ** 0.5), float((torch.trace(M) - 1) / 2)))


print("| case | class | inliers HIP / ref32 / ref64 | HIP - ref64 (m) | ref32 - ref64 (m) | ref32: 8 threads - 1 thread (m) | HIP - ref32 (m) | HIP - ref64 (rad) | ref32 - ref64 (rad) |")
print("|---|---|---|---|---|---|---|---|---|")
rows = [(f"{s}/{n}", cls, f"s{s}_n{n}", inp) for (s, n), cls, inp in margin_cases.all_cases()]
for fa, fb in ((701, 702), (900, 901)):
    rows.append((f"frames {fa} -> {fb}", "full size", f"path{fa}_{fb}", (torch.from_numpy(g[f"path{fa}_{fb}.desc_src"]), torch.from_numpy(g[f"path{fa}_{fb}.desc_dst"]), None, None, 0.5)))
sums = [0.0, 0.0, 0]
for name, cls, k, (s, d, ms, md, ns) in rows:
    R, T, conf, rmse = dec.registration_forward(s, d, ms, md, num_sample=ns)
    R, T = R.cpu().double().numpy(), T.cpu().double().numpy()
    a, b = np.linalg.norm(T - g[k + ".ref64.T"]), np.linalg.norm(g[k + ".ref32.T"] - g[k + ".ref64.T"])
    if cls == "margin":
        sums[0] += a; sums[1] += b; sums[2] += 1
    print(f"| {name} | {cls} | {conf.numel()} / {int(g[k + '.ref32.n_conf'])} / {int(g[k + '.ref64.n_conf'])} | {a:.2e} | {b:.2e} | "
          f"{np.linalg.norm(g[k + '.ref32.T'] - g[k + '.ref32t1.T']):.2e} | {np.linalg.norm(T - g[k + '.ref32.T']):.2e} | {ang(R, g[k + '.ref64.R']):.2e} | {ang(g[k + '.ref32.R'], g[k + '.ref64.R']):.2e} |")
print(f"\nmargin class, mean over {sums[2]} cases: HIP - ref64 {sums[0] / sums[2]:.2e} m, ref32 - ref64 {sums[1] / sums[2]:.2e} m")
