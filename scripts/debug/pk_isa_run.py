"""debug (GPU box): the co-residency stress on the code objects scripts/debug/pk_isa_variants.py made.

Every deeppointmap_amd/csrc/build/pkiso/<name>.hsaco holds group_gather_ln_max_kernel<32,1,true> in one ISA variant.  The kernel
is launched through hipModuleLaunchKernel on a stream confined to half of the compute units while the library's bf16x3 GEMM runs
on another stream confined to the SAME compute units (tests/corun_stress.py found: disjoint compute units never fail, shared ones
fail in half of the launches), `iters` times; every output is compared with the variant's own output on an idle chip.
Prints one JSON line per variant: failing launches, and what the differing elements look like.
usage: pk_isa_run.py [iters] [noise: bf16x3 | fp32 | none] [variant names ...]
"""
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import corun_stress  # noqa: E402
from deeppointmap_amd import ops, synthetic  # noqa: E402
from deeppointmap_amd.config import default_args  # noqa: E402
from deeppointmap_amd.encoder import Encoder  # noqa: E402
from deeppointmap_amd.weights import init_procedural  # noqa: E402

KERNEL = b"_ZN12_GLOBAL__N_126group_gather_ln_max_kernelILi32ELi1ELb1EEEvPKfS2_S2_S2_S2_PKiS2_iS2_S2_iiixifPf"
PKISO = os.path.join(ROOT, "deeppointmap_amd", "csrc", "build", "pkiso")
hip = ctypes.CDLL(corun_stress.hip_runtime_path())


def chk(err, what):
    if err:
        raise RuntimeError(f"{what}: hip error {err}")


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    noise = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
    names = sys.argv[3:] or [l.split("\t")[0] for l in open(os.path.join(PKISO, "MANIFEST.tsv")) if l.strip()]
    split = os.environ.get("STRESS_CU_SPLIT", "same")
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    enc = init_procedural(Encoder(default_args())).to(dev)
    B, N, S, K, C = 4, 16384, 2048, 32, 32
    pts, pad = synthetic.frames(B, N, start=40)
    xyz, lengths = ops.prepare_points(pts.to(dev).contiguous(), pad.to(dev).contiguous())
    _, cen, _ = ops.fps(xyz, lengths, S)
    idx = ops.knn_hybrid(xyz, lengths, cen, K, 0.05)
    m = "downsampler.0.sa.mlp"
    W = enc.p(m + ".0.weight").reshape(C, 19)
    W0, b0 = enc.p("point_mlp0.weight").reshape(16, 3), enc.p("point_mlp0.bias")
    A = (W[:, :16].double() @ W0.double()).float().contiguous()
    cvec = (W[:, :16].double() @ b0.double() + enc.p(m + ".0.bias").double()).float().contiguous()
    Wr = W[:, 16:].contiguous()
    gm, bt = enc.p(m + ".1.ln.weight").contiguous(), enc.p(m + ".1.ln.bias").contiguous()
    total, cpw = B * S, 2
    grid = (total + 4 * cpw - 1) // (4 * cpw)
    lo = [0xFFFFFFFF] * 4 + [0] * 4
    hi = [0] * 4 + [0xFFFFFFFF] * 4
    vstream = corun_stress._masked_stream(dev, lo) if split else torch.cuda.Stream(device=dev)
    nstream = corun_stress._masked_stream(dev, hi if split == "halves" else lo) if split else torch.cuda.Stream(device=dev)
    nfn = corun_stress._noise(dev, noise)
    torch.cuda.synchronize()

    def launcher(path):
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        chk(hip.hipModuleLoad(ctypes.byref(mod), path.encode()), "hipModuleLoad " + path)
        chk(hip.hipModuleGetFunction(ctypes.byref(fn), mod, KERNEL), "hipModuleGetFunction")

        def call(out, stream):
            vals = [ctypes.c_void_p(0), ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(cvec.data_ptr()), ctypes.c_void_p(xyz.data_ptr()),
                    ctypes.c_void_p(cen.data_ptr()), ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(Wr.data_ptr()), ctypes.c_int(3),
                    ctypes.c_void_p(gm.data_ptr()), ctypes.c_void_p(bt.data_ptr()), ctypes.c_int(N), ctypes.c_int(S), ctypes.c_int(K),
                    ctypes.c_longlong(total), ctypes.c_int(cpw), ctypes.c_float(1.0 / 0.05), ctypes.c_void_p(out.data_ptr())]
            params = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in vals])
            chk(hip.hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(stream.cuda_stream), params, None), "launch")
        return call

    for name in names:
        call = launcher(os.path.join(PKISO, name + ".hsaco"))
        with torch.cuda.stream(vstream):
            ref = torch.empty(B, S, C, device=dev)
            call(ref, vstream)
            vstream.synchronize()
            idle_bad = 0
            for _ in range(20):
                o = torch.empty(B, S, C, device=dev)
                call(o, vstream)
                idle_bad += int(not torch.equal(o, ref))
        stop, launches = threading.Event(), [0]

        def noise_loop():
            with torch.cuda.stream(nstream):
                while not stop.is_set():
                    for _ in range(30):
                        nfn()
                    launches[0] += 30
                    nstream.synchronize()
        th = None
        if nfn is not None:
            th = threading.Thread(target=noise_loop)
            th.start()
            time.sleep(0.05)
        bad, n_elem, n_rows, larger, cols, lanes = 0, 0, 0, 0, {}, {}
        t0 = time.time()
        with torch.cuda.stream(vstream):
            for _ in range(iters):
                o = torch.empty(B, S, C, device=dev)
                call(o, vstream)
                if not torch.equal(o, ref):
                    bad += 1
                    d = (o != ref).reshape(-1, C)
                    n_elem += int(d.sum())
                    n_rows += int(d.any(1).sum())
                    larger += int((o.reshape(-1, C)[d] > ref.reshape(-1, C)[d]).sum())
                    for c in d.any(0).nonzero().flatten().tolist():
                        cols[c] = cols.get(c, 0) + 1
        dt = time.time() - t0
        stop.set()
        if th is not None:
            th.join()
        torch.cuda.synchronize()
        print(json.dumps({"variant": name, "noise": noise, "cu_split": split, "iters": iters, "bad_launches": bad, "bad_on_idle_chip": idle_bad,
                          "differing_elements": n_elem, "differing_rows": n_rows, "elements_larger_than_ref": larger,
                          "columns_hit": dict(sorted(cols.items())), "noise_launches": launches[0], "seconds": round(dt, 2)}), flush=True)


if __name__ == "__main__":
    main()
