"""The pipelined step taken apart by TIMING on the real pipeline (HotPath.submit; nothing profiled, nothing re-implemented):
stages are switched off by handing back a cached result of theirs (wrong data flow, right timing of what remains), so every row
is the throughput of the remaining stages in the shipped stream layout.  Then the feature + registration stages next to resident
do-nothing workgroups with the sampling kernels' footprint (scripts/micro/occupy.hip): what the sampling stage costs the others by
being RESIDENT (wave slots, LDS) as opposed to by what it does.  Timing waits for the pipeline's own streams only.
Output: a markdown table (profiles/r06_step_model.md).  Needs deeppointmap_amd/csrc/build/libocc.so
(hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/occupy.hip)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd import pipeline as pipeline_mod
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
F = 64
N_STEPS = int(os.environ.get("STEPS", "40"))
pts, pad = synthetic.frames(F, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * 60).contiguous()
main = torch.cuda.current_stream(dev)

# ---- originals, and one cached result of each stage (taken from a normal run)
orig = dict(presample=hot.encoder.presample, first=hot.encoder.sample_first_level, grids=pipeline_mod.ops.information_matrix_grids,
            extract=hot.extract, register=hot.register)
cache = {}


def deep_clone(x):
    """a private copy of a stage result: the feature stage consumes parts of the sampling result (tie / todo queues of the search
    grids), so a cached one is handed out as a copy (~70 MB of device copies per batch, ~20 us)"""
    if isinstance(x, torch.Tensor):
        return x.clone()
    if isinstance(x, dict):
        return {k: deep_clone(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(deep_clone(v) for v in x)
    return x


def caching(name, fn):
    def f(*a, **k):
        out = fn(*a, **k)
        if name not in cache:
            cache[name] = deep_clone(out) if name == "presample" else out   # before the feature stage touches it
        return out
    return f


hot.encoder.presample = caching("presample", orig["presample"])
pipeline_mod.ops.information_matrix_grids = caching("grids", orig["grids"])
hot.extract = caching("extract", orig["extract"])
hot.register = caching("register", orig["register"])
for _ in range(6):
    hot.submit(pts, pad, pcd)
hot.flush()
torch.cuda.synchronize()


def configure(G=True, Fs=True, R=True):
    hot.encoder.presample = orig["presample"] if G else (lambda *a, **k: deep_clone(cache["presample"]))
    hot.encoder.sample_first_level = orig["first"] if G else (lambda pl, pd: [None] * len(pl))
    pipeline_mod.ops.information_matrix_grids = orig["grids"] if G else (lambda *a, **k: cache["grids"])
    hot.extract = orig["extract"] if Fs else (lambda points, padding, presampled=None: cache["extract"])
    hot.register = orig["register"] if R else (lambda *a, **k: cache["register"])


def wait_pipeline():
    main.synchronize()
    for s in hot._side["geo"] + [hot._side["reg"]]:
        s.synchronize()


rows = []
# ---- probes: four operators of the feature / registration stages timed with HIP events on their own stream INSIDE every
# configuration: do the kernels themselves get slower next to the sampling stage, or do gaps open between them?
import deeppointmap_amd.ops as ops_mod
probe_events = {}


def probe(name, pick=lambda *a, **k: True):
    fn = getattr(ops_mod, name)

    def f(*a, **k):
        if not probing[0] or not pick(*a, **k):
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        probe_events.setdefault(name, []).append((e0, e1))
        return out
    setattr(ops_mod, name, f)


probing = [False]
probe("group_mlp_max_from_xyz")                                                    # F: the first-level gather (1 per batch)
probe("knn_hybrid", lambda points, lengths, centers, K, *a, **k: points.shape[1] == 65536)   # F: the first-level neighbour search
probe("linear_kvplanes", lambda x, W, *a, **k: W.shape[0] == 768 and x.shape[0] == 2 * F * 256)  # R: the 256 -> 768 projections
probe("information_matrix_batched")                                                # R: the nearest-neighbour search + moments
probe("fps", lambda xyz, *a, **k: xyz.shape[1] == 65536)                             # G: packing + the 4095 rounds of one launch
PROBES = ("group_mlp_max_from_xyz", "knn_hybrid", "linear_kvplanes", "information_matrix_batched", "fps")


def timed(label, n=N_STEPS, warm=6):
    for _ in range(warm):
        hot.submit(pts, pad, pcd)
    hot.flush()
    wait_pipeline()
    probe_events.clear()
    probing[0] = True
    t = time.perf_counter()
    th = 0.0
    for _ in range(n):
        h = time.perf_counter()
        hot.submit(pts, pad, pcd)
        th += time.perf_counter() - h
    hot.flush()
    wait_pipeline()
    dt = (time.perf_counter() - t) / n * 1e3
    probing[0] = False
    pr = {k: (sum(a.elapsed_time(b) for a, b in v) / len(v) * 1e3 if v else float("nan")) for k, v in ((p, probe_events.get(p, [])) for p in PROBES)}
    rows.append((label, dt, th / n * 1e3, pr))
    print(f"{label}: {dt:.3f} ms per step (host: {th / n * 1e3:.3f} ms per submit); probes us: " + ", ".join(f"{k} {v:.0f}" for k, v in pr.items()), flush=True)
    return dt


configure()
timed("G | F | R: the whole pipeline")
configure(R=False); timed("G | F (registration switched off)")
configure(Fs=False); timed("G | R (features switched off)")
configure(G=False); timed("F | R (sampling + grids switched off)")
# the sampling stage split in two: its first-level sampling (Sort-Tile-Recursive packing + the 4095 rounds, 128 latency-bound
# workgroups in flight) and everything else it launches (staging, lower levels, search grids of the neighbour queries and of the
# information matrix: ~0.5 ms of kernel time per batch, partly chip-filling)
import deeppointmap_amd.encoder as enc_mod
real_fps = enc_mod.ops.fps
first_level = {}


def fps_cached(xyz, lengths, K, algo=0):
    key = (tuple(xyz.shape), K)
    if key not in first_level:
        first_level[key] = real_fps(xyz, lengths, K, algo)
    return tuple(t.clone() for t in first_level[key]) if xyz.shape[1] == 65536 else real_fps(xyz, lengths, K, algo)


configure()
enc_mod.ops.fps = fps_cached
timed("G without its first-level sampling | F | R")
enc_mod.ops.fps = real_fps
configure(G=False)
pre0 = cache["presample"]


def presample_sampling_only(*a, **k):
    real_fps(pre0["xyz"], pre0["lengths"], cfg.encoder.npoint[0])
    return deep_clone(cache["presample"])


hot.encoder.presample = presample_sampling_only
timed("first-level sampling only | F | R")
# the same launches at the same places of the host's enqueue order, but on two OTHER streams: the feature stage no longer waits for them
side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
side_n = [0]


def presample_sampling_unordered(*a, **k):
    st = side[side_n[0] % 2]
    side_n[0] += 1
    with torch.cuda.stream(st):
        real_fps(pre0["xyz"], pre0["lengths"], cfg.encoder.npoint[0])
    return deep_clone(cache["presample"])


hot.encoder.presample = presample_sampling_unordered
timed("first-level sampling enqueued at the same places but on streams of its own (nothing waits for it) | F | R")
for s_ in side:
    s_.synchronize()
hot.encoder.presample = presample_sampling_only
# the sampling launch REPLACED by 64 sleeping workgroups of its shape that stay for 6.5 ms, at its place in the pipeline (the feature
# stage waits for them exactly as it waits for the sampling): is the price paid for what the rounds DO, or for the 6.5 ms in the chain?
occ_early = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeppointmap_amd", "csrc", "build", "libocc.so"))
occ_early.launch_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]


def presample_sleepers(*a, **k):
    occ_early.launch_occupy(64, 1024, 650_000, 33408, torch.cuda.current_stream(dev).cuda_stream)
    return deep_clone(cache["presample"])


hot.encoder.presample = presample_sleepers
timed("64 SLEEPING workgroups for 6.5 ms in the sampling launch's place | F | R")
hot.encoder.presample = presample_sampling_only
for nf in (32, 16, 8):   # fewer sampling workgroups in flight: how the price scales with their number
    xs, ls = pre0["xyz"][:nf].contiguous(), pre0["lengths"][:nf].contiguous()

    def presample_some(*a, _xs=xs, _ls=ls, **k):
        real_fps(_xs, _ls, cfg.encoder.npoint[0])
        return deep_clone(cache["presample"])

    hot.encoder.presample = presample_some
    timed(f"first-level sampling of {nf} of the 64 frames only | F | R")
hot.encoder.presample = presample_sampling_only
from deeppointmap_amd import _lib
if _lib.experimental():   # DPM_LIB = a -DDPM_EXPERIMENT build: the packing kernels without the 4095 rounds
    os.environ["DPM_ABLATE_FPS_ROUNDS"] = "1"
    timed("first-level packing only (no rounds) | F | R")
    del os.environ["DPM_ABLATE_FPS_ROUNDS"]
    # ... and the rounds with parts of them switched off (wrong picks -- the feature stage runs on the cached, correct sampling
    # result), every round held to 1.6 us so that all variants keep their workgroups resident equally long
    os.environ["DPM_FPS_PACE"] = "160"
    for bits, what in ((0, "complete rounds"), (1, "rounds without bucket updates (no global loads / stores, no per-bucket arg-max)"),
                       (2, "rounds without the exchange (no LDS, no barrier)"), (3, "rounds with the box test only")):
        os.environ["DPM_FPS_ABLATE"] = str(bits)
        timed(f"first-level sampling only, paced at 1.6 us per round, {what} | F | R")
    del os.environ["DPM_FPS_PACE"], os.environ["DPM_FPS_ABLATE"]
configure(G=False, R=False); timed("F alone")
configure(G=False, Fs=False); timed("R alone")
configure(Fs=False, R=False); timed("G alone (two passes in flight)")

occ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deeppointmap_amd", "csrc", "build", "libocc.so"))
occ.launch_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
so = torch.cuda.Stream(device=dev)
configure(G=False)
for wgs, threads, lds in ((128, 1024, 33408), (128, 1024, 4736), (128, 512, 33408), (128, 512, 4736), (128, 256, 4736), (64, 1024, 33408), (256, 512, 4736)):
    torch.cuda.synchronize()
    n = 24
    # resident for the whole measurement (~(6 + 24) steps of ~4 ms): ONE launch of 200 ms on a side stream, not waited for by the timing
    occ.launch_occupy(wgs, threads, 20_000_000, lds, so.cuda_stream)
    timed(f"F | R next to {wgs} idle workgroups of {threads} threads holding {lds} B of LDS each", n=n)
# ... and next to workgroups of the sampling kernel's shape that DO a chosen part of a round every 1.5 us (scripts/micro/occupy.hip):
# which activity is it that costs the other stages time?
occ.launch_active.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
nblk = 128 * 1024                                        # 128 frames x 1024 buckets of 1 KB (+ 256 B of `closest` each)
state = torch.zeros(nblk * (1024 + 256) // 4, device=dev)
sink = torch.zeros(4, device=dev)
for mode, what in ((1, "~40 dependent vector instructions per wave"), (2, "one bucket-sized load + closest load / store per wave"),
                   (4, "the exchange: LDS write, barrier, two dependent LDS reads"), (7, "all three")):
    torch.cuda.synchronize()
    occ.launch_active(128, 1024, 20_000_000, 33408, mode, 150, state.data_ptr(), nblk, sink.data_ptr(), so.cuda_stream)
    timed(f"F | R next to 128 workgroups of 1024 threads doing, every 1.5 us: {what}", n=24)
# ... and next to the REAL first-level sampling, launched back to back on two side streams that nothing waits for: is the price of
# the sampling stage paid for sharing the chip with it, or for depending on it?
torch.cuda.synchronize()
sx, sy = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for i in range(2 * 22):   # 22 launches of ~6 ms per stream: longer than the measurement
    with torch.cuda.stream(sx if i % 2 == 0 else sy):
        real_fps(pre0["xyz"], pre0["lengths"], cfg.encoder.npoint[0])
timed("F | R next to the real first-level sampling (two launches in flight) that nothing waits for", n=24)
torch.cuda.synchronize()
if _lib.experimental():   # the same with parts of a round switched off (wrong picks, nobody reads them), unpaced
    for bits, what in ((1, "without bucket updates"), (2, "without the exchange"), (3, "box test only")):
        os.environ["DPM_FPS_ABLATE"] = str(bits)
        for i in range(2 * 22):
            with torch.cuda.stream(sx if i % 2 == 0 else sy):
                real_fps(pre0["xyz"], pre0["lengths"], cfg.encoder.npoint[0])
        timed(f"F | R next to the sampling that nothing waits for, rounds {what}", n=24)
        torch.cuda.synchronize()
    del os.environ["DPM_FPS_ABLATE"]
configure()
timed("G | F | R once more")
# the sampling stage of batch i is enqueued behind `geo.wait_stream(caller's stream)` (inputs the caller may have produced there):
# with the feature stage of batch i - 3 at the head of that stream, G(i) cannot start before F(i - 3) has finished.  Without it:
for g_ in hot._side["geo"]:
    g_.wait_stream = lambda s: None
timed("G | F | R, the sampling streams NOT waiting for the caller's stream (inputs resident)")
configure(R=False); timed("G | F, the same")
configure(Fs=False); timed("G | R, the same")
configure()
print()
print("| configuration | ms per 64-frame step | host ms per submit | first-level gather us | first-level neighbour search us | 256 -> 768 projection us | information matrices us | first-level sampling launch us |")
print("|---|---|---|---|---|---|---|---|")
for label, dt, th, pr in rows:
    print(f"| {label} | {dt:.3f} | {th:.3f} | " + " | ".join(f"{pr[p]:.0f}" for p in PROBES) + " |")
