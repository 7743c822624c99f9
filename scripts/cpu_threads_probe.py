import sys, time, torch
sys.path.insert(0, '.')
from oracle import dpm_oracle as O
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.params import encoder_shapes, decoder_shapes
from deeppointmap_amd.weights import procedural_state_dict
torch.set_grad_enabled(False)
cfg = default_args(); sde = procedural_state_dict(encoder_shapes(cfg)); sdd = procedural_state_dict(decoder_shapes(cfg))
pts, pad = synthetic.frames(2, 65536)
import os
print('cpu_count', os.cpu_count())
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    O.encoder_forward(sde, cfg, pts[:1], pad[:1], fast_fps=True)
    t = time.perf_counter(); c, f, _ = O.encoder_forward(sde, cfg, pts[:1], pad[:1], fast_fps=True); te = time.perf_counter() - t
    d = torch.cat([f[0], c[0] * 60], 0)
    t = time.perf_counter(); R, T, _, _ = O.registration_forward(sdd, cfg, d, d, 0.5); tr = time.perf_counter() - t
    t = time.perf_counter(); O.information_matrix(pts[0] * 60, pts[1] * 60, O.se3(R, T)); ti = time.perf_counter() - t
    print(f'threads {th}: encode {te:.2f}s register {tr:.3f}s infomat {ti:.2f}s', flush=True)
