"""debug: fuzz_encoder cases replayed against the oracle's output computed in the BUILD container (gpurun_in_case_oracle.pt, made
there with `python scripts/debug/enc_cases_vs_build_oracle.py make B N start L0 .. [-- B N start L0 ..]`; untracked, it travels with gpurun: the oracle's kNN distances come from the host's sgemm, whose last bit -- and with it one neighbour set in a few hundred
passes -- differs between the build container, where the oracle is pinned to the reference, and the GPU host)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import reduced_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
cfg = reduced_args()
PT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_in_case_oracle.pt")
if len(sys.argv) > 1 and sys.argv[1] == "make":   # build container, CPU only
    from deeppointmap_amd.params import encoder_shapes
    from deeppointmap_amd.weights import procedural_state_dict
    sd = procedural_state_dict(encoder_shapes(cfg))
    out, rest, i = {}, sys.argv[2:], 0
    while rest:
        case, rest = (rest[:rest.index("--")], rest[rest.index("--") + 1:]) if "--" in rest else (rest, [])
        B, N, start = (int(a) for a in case[:3])
        lens = [int(a) for a in case[3:3 + B]]
        pts, pad = synthetic.frames(B, N, start=start)
        for b, L in enumerate(lens):
            if L < N:
                pad[b, L:] = True
                pts[b, :, L:] = 0.0
        oc, of, _ = O.encoder_forward(sd, cfg, pts, pad, fast_fps=False)
        out[f"c{i}"], i = dict(args=(B, N, start, lens), oc=oc, of=of), i + 1
    torch.save(out, PT)
    sys.exit(0)
enc = init_procedural(Encoder(cfg)).to("cuda:0")
sd = {k: v.detach().cpu() for k, v in enc.flat().items()}
ref = torch.load(PT)
for name, r in ref.items():
    B, N, start, lens = r["args"]
    pts, pad = synthetic.frames(B, N, start=start)
    for b, L in enumerate(lens):
        if L < N:
            pad[b, L:] = True
            pts[b, :, L:] = 0.0
    coor, fea, _ = enc(pts, pad)
    oc, of, _ = O.encoder_forward(sd, cfg, pts, pad, fast_fps=False)
    print(f"{name} {r['args']}: HIP vs oracle HERE {float((fea.cpu() - of).abs().max()):.2e}; HIP vs oracle of the BUILD container "
          f"{float((fea.cpu() - r['of']).abs().max()):.2e} (key points equal {torch.equal(coor.cpu(), r['oc'])}); oracle here vs there {float((of - r['of']).abs().max()):.2e}")
