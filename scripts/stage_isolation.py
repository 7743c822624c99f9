"""Where the pipelined step goes: throughput of the feature + registration stages with the geometry stage removed
(one precomputed sampling result reused), and of the geometry stage alone (two passes in flight)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural

dev = torch.device("cuda:0")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
F = 64
pts, pad = synthetic.frames(F, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * 60).contiguous()
pairs = [((f - 1) % F, f) for f in range(F)]
pre = hot.encoder.presample(pts, pad)
torch.cuda.synchronize()
sb = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)


def fr_steps(n):
    prev = None
    for _ in range(n):
        desc = hot.extract(pts, pad, presampled=pre)
        ev = main.record_event()
        if prev is not None:
            d, e = prev
            with torch.cuda.stream(sb):
                sb.wait_event(e)
                hot.register(d, pcd, pairs, materialize=False)
        prev = (desc, ev)
    torch.cuda.synchronize()


fr_steps(3)
t = time.perf_counter(); fr_steps(20); dt = (time.perf_counter() - t) / 20
print(f"feature + registration stages only (two streams): {dt * 1e3:.2f} ms per step")

ga, gb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def g_steps(n):
    for i in range(n):
        with torch.cuda.stream(ga if i % 2 == 0 else gb):
            hot.encoder.presample(pts, pad)
    torch.cuda.synchronize()
g_steps(4)
t = time.perf_counter(); g_steps(20); dt = (time.perf_counter() - t) / 20
print(f"geometry stage only (two passes in flight): {dt * 1e3:.2f} ms per batch")

# ---- what exactly costs the combined pipeline its ~1.6 ms?  Keep 128 workgroups of 1024 SLEEPING threads resident
# (the wave-slot / LDS footprint of two geometry passes, without their memory traffic) while the feature +
# registration stages run.
import ctypes, subprocess, tempfile
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(1024) void occupy(long long cycles, int lds_words) {
    extern __shared__ int pad[];
    if (lds_words && threadIdx.x == 0) pad[0] = 1;
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(64);
}
extern "C" int launch_occupy(int wgs, int threads, long long cycles, int lds_bytes, void *stream) {
    hipLaunchKernelGGL(occupy, dim3(wgs), dim3(threads), lds_bytes, (hipStream_t)stream, cycles, lds_bytes / 4);
    return (int)hipGetLastError();
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "occ.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(d, "occ.hip"), "-o",
                       os.path.join(d, "libocc.so")])
occ = ctypes.CDLL(os.path.join(d, "libocc.so"))
occ.launch_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
so = torch.cuda.Stream(device=dev)
for wgs, threads, lds in ((128, 1024, 33408), (128, 512, 33408), (128, 256, 33408), (128, 1024, 0), (64, 1024, 33408)):
    torch.cuda.synchronize()
    # 100 MHz constant clock: 25 ms per launch, relaunched back to back on the side stream
    for _ in range(8):
        occ.launch_occupy(wgs, threads, 2_500_000, lds, so.cuda_stream)
    fr_steps(2)
    t = time.perf_counter(); fr_steps(12); dt = (time.perf_counter() - t) / 12
    torch.cuda.synchronize()
    print(f"feature + registration with {wgs} sleeping workgroups of {threads} threads, {lds} B LDS: {dt * 1e3:.2f} ms per step")
