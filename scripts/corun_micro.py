"""Decision gate of round 4: do a matrix-pipe kernel of the registration stage and a vector-ALU kernel of the feature stage
run AT THE SAME TIME when they are launched on two HIP streams, or does the chip serialise them?

For every pair (A = matrix kernel, B = vector kernel) of the step's chip-filling kernels:
    alone      n_A launches of A / n_B launches of B on one stream each (n chosen so that both take ~T ms)
    serial     the same launches on ONE stream
    concurrent the same launches on TWO streams released by one event
and reports concurrent / (alone_A + alone_B)  (1.0 = the chip serialises, max/sum = perfect overlap), next to the shader
clock `rocm-smi` shows while each regime runs for ~1.5 s (the chip clocks to its power budget: MI355X_MICROARCH.md,
"DVFS give-back").

Occupancy shapes: with an experimental library (`python deeppointmap_amd/csrc/build.py --out deeppointmap_amd/libdpm_exp.so
-DDPM_EXPERIMENT`, `DPM_LIB=...`) the kernels take unused dynamic LDS from DPM_GEMM_LDS_PAD / DPM_ATT_LDS_PAD /
DPM_KNN_LDS_PAD / DPM_GATHER_LDS_PAD, which caps their workgroups per CU (256-thread workgroups: k per CU = k waves per
SIMD); `--sweep` re-runs the pairs under {matrix kernel 8 / 4 / 2 per CU} x {vector kernel 8 / 4 per CU}.

    python scripts/corun_micro.py [--sweep] [--clock] [--ms 12] > profiles-ready markdown on stdout
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deeppointmap_amd import _lib, ops, synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural

ap = argparse.ArgumentParser()
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--clock", action="store_true")
ap.add_argument("--ms", type=float, default=12.0, help="length of every timed regime per kernel")
args = ap.parse_args()

dev = torch.device("cuda:0")
F, N = 64, 65536
cfg = default_args()
enc = init_procedural(Encoder(cfg)).to(dev)
pts, pad = synthetic.frames(F, N)
pts, pad = pts.to(dev), pad.to(dev)
xyz, lengths = ops.prepare_points(pts, pad)
_, new_xyz, new_len = ops.fps(xyz, lengths, 4096)
grid0 = ops.knn_grid(xyz, lengths, 0.05)
idx0 = ops.knn_hybrid(xyz, lengths, new_xyz, 32, 0.05)
pcd_m = (pts * 60.0).contiguous()
src = torch.arange(F, dtype=torch.int32, device=dev).roll(1)
dst = torch.arange(F, dtype=torch.int32, device=dev)
grids = ops.information_matrix_grids(pcd_m, dst)
Rt = torch.zeros(F, 12, device=dev)
Rt[:, 0] = Rt[:, 4] = Rt[:, 8] = 1.0
Rt[:, 9] = 0.5
info = torch.empty(F, 36, device=dev)
gx = torch.randn(2 * F * 256, 256, device=dev)
gw, gb = torch.randn(768, 256, device=dev) / 16, torch.randn(768, device=dev)
go = torch.empty(2 * F * 256, 768, device=dev)
qkv = torch.randn(2 * F * 256, 768, device=dev)
ao = torch.empty(2 * F * 256, 256, device=dev)
sa = "downsampler.0.sa.mlp"
W0, b0 = enc.p("point_mlp0.weight"), enc.p("point_mlp0.bias")
Wsa, bsa = enc.p(sa + ".0.weight"), enc.p(sa + ".0.bias")
gsa, besa = enc.p(sa + ".1.ln.weight"), enc.p(sa + ".1.ln.bias")

KERNELS = {
    "gemm768": ("matrix", lambda: ops.linear(gx, gw, gb, out=go)),
    "attention": ("matrix", lambda: ops.attention(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], 2 * F, 256, 256, 8, out=ao, kv_shift=F)),
    "knn_sa0": ("vector", lambda: ops.knn_hybrid(xyz, lengths, new_xyz, 32, 0.05)),   # grid build (one workgroup per frame) + search + tie replay: a prebuilt grid serves ONE search
    "gather32": ("vector", lambda: ops.group_mlp_max_from_xyz(xyz, W0, b0, new_xyz, idx0, Wsa, bsa, gsa, besa, 0.05)),
    "nn1": ("vector", lambda: ops.information_matrix_batched(pcd_m, src, dst, Rt, info, grids=grids)),
}
PAIRS = [("gemm768", "knn_sa0"), ("attention", "gather32"), ("gemm768", "nn1"), ("gemm768", "gather32"), ("attention", "knn_sa0")]
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def run(fn, n, stream):
    with torch.cuda.stream(stream):
        for _ in range(n):
            fn()


def run2(fa, na, fb, nb):
    """A on s1 and B on s2, enqueued alternately so that neither stream starts a regime ahead of the other"""
    for i in range(max(na, nb)):
        if i < na:
            with torch.cuda.stream(s1):
                fa()
        if i < nb:
            with torch.cuda.stream(s2):
                fb()


def timed(body):
    """wall time (ms, HIP events on the default stream around a fork / join of s1 and s2)"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream(dev)
    e0.record(cur)
    s1.wait_event(e0), s2.wait_event(e0)
    body()
    cur.wait_stream(s1), cur.wait_stream(s2)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def alone_us(fn):
    for _ in range(3):
        fn()
    t = timed(lambda: run(fn, 10, s1)) / 10
    n = max(3, int(round(args.ms / t)))
    return timed(lambda: run(fn, n, s1)) / n * 1e3, n


def smi_clock():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        card = next(iter(j.values()))
        sclk = next((v for k, v in card.items() if "sclk" in k.lower()), "?")
        pwr = next((v for k, v in card.items() if "power" in k.lower() and "socket" in k.lower()), None) or \
            next((v for k, v in card.items() if "power" in k.lower()), "?")
        return f"{sclk} / {pwr} W"
    except Exception as e:  # noqa: BLE001
        return f"? ({type(e).__name__})"


def clock_during(body, seconds=1.5):
    """keep `body` (a callable that enqueues ~args.ms of work) running for `seconds`, sample rocm-smi in the middle"""
    got = []
    th = threading.Thread(target=lambda: (time.sleep(seconds * 0.4), got.append(smi_clock())))
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        body()
        torch.cuda.synchronize()
    th.join()
    return got[0] if got else "?"


def measure(tag=""):
    al = {k: alone_us(fn) for k, (_, fn) in KERNELS.items()}
    print(f"\n### {tag or 'shipped occupancy'}\n")
    print("| kernel | kind | alone us | launches per regime |\n|---|---|---|---|")
    for k, (kind, _) in KERNELS.items():
        print(f"| {k} | {kind} | {al[k][0]:.1f} | {al[k][1]} |")
    print("\n| pair (matrix x vector) | alone A ms | alone B ms | sum | serial (1 stream) | concurrent (2 streams) | concurrent / sum | max / sum (perfect overlap) |")
    print("|---|---|---|---|---|---|---|---|")
    rows = []
    for a, b in PAIRS:
        fa, fb = KERNELS[a][1], KERNELS[b][1]
        na, nb = al[a][1], al[b][1]
        ta = timed(lambda: run(fa, na, s1))
        tb = timed(lambda: run(fb, nb, s2))

        def interleaved():
            with torch.cuda.stream(s1):
                for i in range(max(na, nb)):
                    if i < na:
                        fa()
                    if i < nb:
                        fb()
        ser = timed(interleaved)
        con = min(timed(lambda: run2(fa, na, fb, nb)) for _ in range(3))
        rows.append((a, b, ta, tb, ser, con))
        print(f"| {a} x {b} | {ta:.2f} | {tb:.2f} | {ta + tb:.2f} | {ser:.2f} | {con:.2f} | **{con / (ta + tb):.3f}** | {max(ta, tb) / (ta + tb):.3f} |")
    if args.clock:
        print("\n| regime (1.5 s each) | sclk / socket power while it runs |\n|---|---|")
        print(f"| idle | {smi_clock()} |")
        for a, b in PAIRS[:3]:
            fa, fb = KERNELS[a][1], KERNELS[b][1]
            na, nb = al[a][1], al[b][1]
            print(f"| {a} alone | {clock_during(lambda: run(fa, na, s1))} |")
            print(f"| {b} alone | {clock_during(lambda: run(fb, nb, s2))} |")
            print(f"| {a} x {b} concurrent | {clock_during(lambda: run2(fa, na, fb, nb))} |")
    sys.stdout.flush()
    return rows


print(f"# co-run micro-benchmark ({'experimental' if _lib.experimental() else 'shipped'} library {os.path.basename(_lib.LIB_PATH)})")
measure()
if args.sweep:
    if not _lib.experimental():
        sys.exit("--sweep needs DPM_LIB=<library built with -DDPM_EXPERIMENT>")
    # 256-thread workgroups, k per CU <=> k waves per SIMD.  pad so that floor(160 KiB / (static + pad)) = k
    def pad_for(k, static):
        return max(0, (160 * 1024) // k - static - 256) if k < 8 else 0
    for km in (8, 4, 2):
        for kv in (8, 4):
            os.environ["DPM_GEMM_LDS_PAD"] = str(pad_for(km, 17408))
            os.environ["DPM_ATT_LDS_PAD"] = str(pad_for(km, 17920))
            os.environ["DPM_KNN_LDS_PAD"] = str(pad_for(kv, 0))
            os.environ["DPM_GATHER_LDS_PAD"] = str(pad_for(kv, 0))
            args.clock = False
            measure(f"matrix kernels <= {km} workgroups per CU, vector kernels (knn, gather) <= {kv}")
