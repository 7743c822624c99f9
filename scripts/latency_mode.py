import sys, time, torch
sys.path.insert(0, '.')
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.registration import calculate_information_matrix_from_pcd, make_descriptors, PoseTool
from deeppointmap_amd.weights import init_procedural
cfg = default_args()
enc, dec = init_procedural(Encoder(cfg)).to('cuda'), init_procedural(Decoder(cfg)).to('cuda')
pts, pad = synthetic.frames(10, 65536)
def T(): torch.cuda.synchronize(); return time.perf_counter()
prev = None
for f in range(10):
    p, m = pts[f:f + 1], pad[f:f + 1]
    t0 = T(); pd_, md = p.cuda(), m.cuda(); t1 = T()
    coor, fea, _ = enc(pd_, md); t2 = T()
    desc = make_descriptors(coor, fea, 60.0)[0]; t3 = T()
    msg = f'frame {f}: h2d {1e3*(t1-t0):.2f} enc {1e3*(t2-t1):.2f} cat {1e3*(t3-t2):.2f}'
    if prev is not None:
        R, Tt, conf, rmse = dec.registration_forward(prev[0], desc, num_sample=0.5); t4 = T()
        se3 = PoseTool.SE3(R.cpu(), Tt.cpu()); t5 = T()
        info = calculate_information_matrix_from_pcd(prev[1], pd_[0] * 60, se3, device='cuda'); t6 = T()
        msg += f' reg {1e3*(t4-t3):.2f} se3 {1e3*(t5-t4):.2f} info {1e3*(t6-t5):.2f}'
    prev = (desc, pd_[0] * 60)
    print(msg)
