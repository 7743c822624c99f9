"""debug: build fps.hip with -DDPM_FPS_STATS into a scratch .so and report active-bucket statistics."""
import ctypes, os, subprocess, sys, torch
sys.path.insert(0, '.')
from deeppointmap_amd import synthetic
src = 'deeppointmap_amd/csrc/fps.hip'
so = '/tmp/libfps_stats.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-DDPM_FPS_STATS', src, 'deeppointmap_amd/csrc/fps_tree.hip', '-o', so])
lib = ctypes.CDLL(so)
lib.dpm_fps_workspace_bytes.restype = ctypes.c_size_t
B, N, K = 1, 65536, 4096
pts = synthetic.frame(0).t().contiguous().unsqueeze(0).cuda()
lens = torch.full((B,), N, dtype=torch.int32, device='cuda')
idx = torch.empty(B, K, dtype=torch.int32, device='cuda'); new = torch.empty(B, K, 3, device='cuda'); nl = torch.empty(B, dtype=torch.int32, device='cuda')
ws = torch.zeros(lib.dpm_fps_workspace_bytes(B, N, K), dtype=torch.uint8, device='cuda')
P = ctypes.c_void_p
ALGO = int(os.environ.get('ALGO', '2'))
for rep in range(2):
    ws.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.dpm_fps_ex(P(pts.data_ptr()), P(lens.data_ptr()), B, N, K, P(idx.data_ptr()), P(new.data_ptr()), P(nl.data_ptr()), P(ws.data_ptr()), ALGO, P(torch.cuda.current_stream().cuda_stream))
    e1.record(); torch.cuda.synchronize()
    base = (ws.data_ptr() + 255) & ~255
    off = base - ws.data_ptr()
    hdr = ws[off:off + 256].view(torch.int64).cpu()
    hdr32 = ws[off:off + 256].view(torch.int32).cpu()
    ph = [int(hdr[32 - 28 + i]) for i in range(6)]
    names = ['box+ballot', 'bucket updates', 'wave argbest+lds write', 'barrier', 'final reduce', 'loop/flush']
    print('  phase cycles per round (mean over waves): ' + ', '.join(f'{n}={v / (K - 1):.0f}' for n, v in zip(names, ph)))
    print(f'rc={rc} time {e0.elapsed_time(e1):.3f} ms  total active buckets {int(hdr[31])}  per round {int(hdr[31]) / (K - 1):.2f}  max per wave-round {int(hdr32[60])}')
