"""The batched information matrix (target grids, then the nearest-neighbour search + moments) alone on the chip, HIP-event
timed, on the bench's synthetic scans: python scripts/infomat_time.py  (DPM_LIB selects a build; prints a digest of the result so that
two builds can be compared bit for bit)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops, _lib, synthetic
dev = torch.device("cuda")
F, N = 64, 65536
pts, _ = synthetic.frames(F, N)
pcd = (pts.to(dev) * synthetic.COOR_SCALE).contiguous()
src = torch.arange(F, dtype=torch.int32, device=dev)
dst = (src + 1) % F
torch.manual_seed(0)
def poses(kind):
    Rt = torch.zeros(F, 20, device=dev)
    if kind == "identity":
        Rt[:, 0] = Rt[:, 4] = Rt[:, 8] = 1.0
    else:  # a random rotation about z plus a few metres, like an unconverged registration
        a = torch.rand(F, device=dev) * 6.28
        Rt[:, 0], Rt[:, 1], Rt[:, 3], Rt[:, 4], Rt[:, 8] = a.cos(), -a.sin(), a.sin(), a.cos(), 1.0
        Rt[:, 9:12] = torch.randn(F, 3, device=dev) * torch.tensor([5.0, 5.0, 0.3], device=dev)
    return Rt
for kind in ("identity", "random"):
    Rt = poses(kind)
    out = torch.zeros(F, 36, device=dev)
    grids = ops.information_matrix_grids(pcd, dst)
    for _ in range(3):
        ops.information_matrix_batched(pcd, src, dst, Rt, out, grids=grids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.information_matrix_batched(pcd, src, dst, Rt, out, grids=grids)
    e1.record(); torch.cuda.synchronize()
    digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{os.path.basename(_lib.LIB_PATH)}: {kind} poses, {F} pairs of {N}-point scans: search {e0.elapsed_time(e1) / 20 * 1e3:.1f} us, "
          f"matched {float(out[:, 21].mean()):.0f} of {N} per pair, digest {digest}")
