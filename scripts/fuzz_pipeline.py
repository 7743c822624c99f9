"""One-off stress of the stream pipeline: batches of varying size / padding are submitted back to back, their input
tensors dropped at once (so that the caching allocator recycles the memory while stages are still in flight), and every
result is compared BIT FOR BIT with an unpipelined step on the same input.  A missing stream dependency or
record_stream shows up as a mismatch."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args, reduced_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural
torch.set_grad_enabled(False)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
full = len(sys.argv) > 3 and sys.argv[3] == "full"
rng = random.Random(seed)
dev = torch.device("cuda:0")
cfg = default_args() if full else reduced_args()
enc, dec = init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev)
pipe, plain = HotPath(enc, dec), HotPath(enc, dec)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    chain = rng.random() < 0.3 and os.environ.get("NO_CHAIN") != "1"
    pipe = HotPath(enc, dec); pipe.chain = chain
    plain = HotPath(enc, dec); plain.chain = chain
    pipe.geometry_depth = rng.choice([1, 2, 2]) if os.environ.get("DEPTH2") != "1" else 2
    inputs, outs = [], []
    seq_scans = rng.random() < 0.8        # a chain carries scans for every batch or for none
    for _ in range(rng.randint(3, 9)):
        F = rng.randint(2, 6) if full else rng.randint(2, 12)
        N = rng.choice([65536, 32768]) if full else rng.choice([4096, 8192, rng.randint(3000, 9000)])
        pts, pad = synthetic.frames(F, N, start=rng.randint(0, 40))
        for f in range(F):
            if rng.random() < 0.3:
                L = rng.randint(N // 2, N)
                pad[f, L:] = True
                pts[f, :, L:] = 0
        with_scans = seq_scans if chain else rng.random() < 0.8
        inputs.append((pts, pad, with_scans))
        p, q = pts.to(dev), pad.to(dev)
        m = (p * 60).contiguous() if with_scans else None
        r = pipe.submit(p, q, m)
        if os.environ.get("KEEP") == "1":
            inputs[-1] = inputs[-1] + ((p, q, m),)
        del p, q, m                       # the caller lets go of its tensors immediately
        if os.environ.get("NO_JUNK") != "1":
            junk = torch.empty(rng.randint(1, 64) << 20, device=dev).normal_()  # ... and allocates something else
            del junk
        if r is not None:
            outs.append((r[0].clone(), r[1].clone()))
    outs += [(d.clone(), t.clone()) for d, t in pipe.flush()]
    torch.cuda.synchronize()
    assert len(outs) == len(inputs), (len(outs), len(inputs))
    for i, (inp, (d, t)) in enumerate(zip(inputs, outs)):
        pts, pad, with_scans = inp[:3]
        p, q = pts.to(dev), pad.to(dev)
        d0, _, t0_ = plain.step(p, q, (p * 60).contiguous() if with_scans else None, materialize=False)
        if not (torch.equal(d, d0) and torch.equal(t.nan_to_num(7.0), t0_.nan_to_num(7.0))):
            bad += 1
            print(f"PIPELINE MISMATCH seed {seed}: sequence {n} batch {i}: F {pts.shape[0]} N {pts.shape[2]} chain {chain} scans {with_scans}: "
                  f"descriptors equal {torch.equal(d, d0)}, table max diff {float((t - t0_).abs().nan_to_num(0).max()):.3e}")
        n += 1
print(f"seed {seed}: {n} batches, {bad} mismatches, {time.time() - t0:.0f} s")
