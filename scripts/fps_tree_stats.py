"""debug: build the FPS sources with -DDPM_FPS_STATS into a scratch .so and report, for the one-wave tree kernel
(algo 4), the cycles per round spent in each phase and the pruning statistics."""
import ctypes, os, subprocess, sys, torch
sys.path.insert(0, '.')
from deeppointmap_amd import synthetic
so = '/tmp/libfps_stats.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
                       '-DDPM_FPS_STATS', 'deeppointmap_amd/csrc/fps.hip', 'deeppointmap_amd/csrc/fps_tree.hip', '-o', so])
lib = ctypes.CDLL(so)
lib.dpm_fps_workspace_bytes.restype = ctypes.c_size_t
B, N, K = int(os.environ.get('B', '1')), 65536, 4096
pts = synthetic.frames(B, N)[0].transpose(1, 2).contiguous().cuda()
lens = torch.full((B,), N, dtype=torch.int32, device='cuda')
idx = torch.empty(B, K, dtype=torch.int32, device='cuda'); new = torch.empty(B, K, 3, device='cuda'); nl = torch.empty(B, dtype=torch.int32, device='cuda')
ws = torch.zeros(lib.dpm_fps_workspace_bytes(B, N, K), dtype=torch.uint8, device='cuda')
P = ctypes.c_void_p
for rep in range(2):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.dpm_fps_ex(P(pts.data_ptr()), P(lens.data_ptr()), B, N, K, P(idx.data_ptr()), P(new.data_ptr()), P(nl.data_ptr()), P(ws.data_ptr()), 4, P(torch.cuda.current_stream().cuda_stream))
    e1.record(); torch.cuda.synchronize()
    off = ((ws.data_ptr() + 255) & ~255) - ws.data_ptr()
    hdr = ws[off:off + 256].view(torch.int64).cpu().tolist()
    R = K - 1
    names = ['select', 'node test', 'load issue', 'leaf evaluation', 'load wait (first leaf)', 'node patch', 'leaf test', 'loop/flush']
    print(f'rc={rc}  {e0.elapsed_time(e1):.3f} ms for {B} frame(s)')
    print('  cycles per round (frame 0): ' + ', '.join(f'{n}={hdr[i] / R:.0f}' for i, n in enumerate(names) if n != '-') + f'  total={sum(hdr[:8]) / R:.0f}')
    print(f'  per round: node passes {hdr[8] / R:.2f}, load groups {hdr[9] / R:.2f}, leaves loaded {hdr[10] / R:.2f}, leaves changed {hdr[12] / R:.2f}, slow selects {hdr[11] / R:.4f}')
