"""How many rows of the first-stage neighbour queries the quarter-wave search hands to the tie replay / the full search."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import ops, synthetic

F, N = int(os.environ.get("F", 64)), 65536
dev = torch.device("cuda:0")
pts, pad = synthetic.frames(F, N)
xyz, lengths = ops.prepare_points(pts.to(dev), pad.to(dev))
fidx, new_xyz, new_len = ops.fps(xyz, lengths, 4096)


def al(x):
    return (x + 255) & ~255


def counts(ws, B, n):
    p = al(ws.data_ptr())
    p = al(p + 32 * B)
    hdr_off = al(ws.data_ptr()) - ws.data_ptr()
    p = al(p + 4 * B * (128 * 128 + 1))
    p = al(p + 16 * B * n)
    off = p - ws.data_ptr()
    c = ws[off:off + 8].view(torch.int32).cpu()
    hdr = ws[hdr_off:hdr_off + 32].cpu()
    return int(c[0]), int(c[1]), hdr[:12].view(torch.float32).tolist(), hdr[12:16].view(torch.int32).item(), hdr[16:20].view(torch.float32).item(), hdr[20:24].view(torch.int32).item()


for name, (p, l, c, K, r) in (("SA0", (xyz, lengths, new_xyz, 32, 0.05)), ("LA0", (new_xyz, new_len, new_xyz, 32, 0.1))):
    ws = ops.knn_grid(p, l, r)
    ops.knn_hybrid(p, l, c, K, r, grid=ws)
    torch.cuda.synchronize()
    t, todo, h, g, err2, H = counts(ws, p.shape[0], p.shape[1])
    print(f"{name}: rows {c.shape[0] * c.shape[1]}  tie {t}  todo {todo}  grid lo=({h[0]:.3f},{h[1]:.3f}) cell={1 / h[2]:.4f} g={g} err2={err2:.2e} H={H}")
