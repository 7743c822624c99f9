"""One-off: one Encoder / Decoder instance driven from several Python threads at once (the reference's SLAM threads share
the modules, system/core.py:54-57), each on its own inputs and -- second half -- its own HIP stream; every result must
equal the serial one bit for bit."""
import os, random, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import reduced_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.registration import calculate_information_matrix_from_pcd
from deeppointmap_amd.weights import init_procedural
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = reduced_args()
enc, dec = init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev)
T = 6
work = []
for t in range(T):
    pts, pad = synthetic.frames(2, random.Random(t).choice([4096, 6000, 8192]), start=5 * t)
    work.append((pts, pad))

def job(t, own_stream, out):
    pts, pad = work[t]
    st = torch.cuda.Stream(device=dev) if own_stream else torch.cuda.current_stream(dev)
    res = []
    with torch.cuda.stream(st):
        for rep in range(6):
            coor, fea, _ = enc(pts, pad)
            d = torch.cat([fea, coor * 60.0], 1)
            R, Tt, conf, rmse = dec.registration_forward(d[0], d[1], num_sample=0.5)
            p = dec.loop_detection_forward(d, d.flip(0))
            SE3 = torch.eye(4); SE3[:3, :3] = R.cpu(); SE3[:3, 3] = Tt.cpu().flatten()
            G = calculate_information_matrix_from_pcd(pts[0] * 60, pts[1] * 60, SE3, device=dev)
            res.append((d.cpu(), R.cpu(), Tt.cpu(), conf.cpu(), rmse, p.cpu(), G))
    out[t] = res

serial = {}
for t in range(T):
    job(t, False, serial)
bad = 0
for own_stream in (False, True):
    for round_ in range(3):
        par = {}
        ths = [threading.Thread(target=job, args=(t, own_stream, par)) for t in range(T)]
        for th in ths: th.start()
        for th in ths: th.join()
        for t in range(T):
            for a, b in zip(par[t], serial[t]):
                same = all(torch.equal(x, y) if isinstance(x, torch.Tensor) else x == y for x, y in zip(a, b))
                if not same:
                    bad += 1
                    print(f"THREAD MISMATCH own_stream {own_stream} thread {t}: " + ", ".join(str(torch.equal(x, y) if isinstance(x, torch.Tensor) else x == y) for x, y in zip(a, b)))
                    break
print(f"{T} threads x 3 rounds x 2 stream modes: {bad} mismatches")
