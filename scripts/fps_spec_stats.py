import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from deeppointmap_amd import synthetic, _lib, ops
lib = _lib.load()
B, N, K = 1, 65536, 4096
pts = synthetic.frame(0).t().contiguous().unsqueeze(0).cuda()
lens = torch.full((B,), N, dtype=torch.int32, device='cuda')
idx = torch.empty(B, K, dtype=torch.int32, device='cuda'); new = torch.empty(B, K, 3, device='cuda'); nl = torch.empty(B, dtype=torch.int32, device='cuda')
ws = torch.zeros(lib.dpm_fps_workspace_bytes(B, N, K), dtype=torch.uint8, device='cuda')
for algo in (2, 3):
    for rep in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.dpm_fps_ex(pts.data_ptr(), lens.data_ptr(), B, N, K, idx.data_ptr(), new.data_ptr(), nl.data_ptr(), ws.data_ptr(), algo, torch.cuda.current_stream().cuda_stream)
        e1.record(); torch.cuda.synchronize()
    base = ((ws.data_ptr() + 255) & ~255) + 256
    off = base - ws.data_ptr()
    dbg = ws[off - 64: off - 32].view(torch.int32).cpu().tolist()
    print(f"algo {algo}: {e0.elapsed_time(e1):.3f} ms  dbg [rounds, j1, j2, j3, over, nc1, nc2, nc3] = {dbg}")
