"""What does the launch-bound tail of the encoder (levels 3-4 and both FeaturePropagations: 28 launches, 0.42 ms of kernel time
in sequence) cost a PIPELINED step?  The tail is computed 1, 2 and 3 times per batch (identical results, Encoder._price_tail) and the
pipelined step timed: the slope is the tail's marginal cost next to the other stages."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural
dev = torch.device("cuda")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
pts, pad = synthetic.frames(64, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * synthetic.COOR_SCALE).contiguous()


def run(steps):
    for _ in range(5):
        hot.submit(pts, pad, pcd)
    hot.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hot.submit(pts, pad, pcd)
    hot.flush(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    for extra in (0, 1, 2, 0):
        hot.encoder._price_tail = extra
        print(f"tail evaluated {1 + extra} x per batch: {run(60):.3f} ms per step", flush=True)

# ... and the same question for PURE latency on the feature stream (a one-thread spin kernel in front of the descriptor emission:
# no chip work, no memory traffic): does the step follow the feature stream's chain length?
from deeppointmap_amd import ops
hot.encoder._price_tail = 0
orig = ops.emit_descriptors
for us in (0, 200, 400, 0):
    def emit(*a, _us=us, **k):
        if _us:
            torch.cuda._sleep(int(_us * 1000))   # the spin kernel counts ~1 cycle per ns on this part (checked below)
        return orig(*a, **k)
    ops.emit_descriptors = emit
    import deeppointmap_amd.encoder as enc_mod
    enc_mod.ops.emit_descriptors = emit
    print(f"spin of {us} us on the feature stream: {run(60):.3f} ms per step", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(400 * 1000); torch.cuda.synchronize()
print(f"(a 400 us spin alone takes {(time.perf_counter() - t0) * 1e3:.3f} ms)")
