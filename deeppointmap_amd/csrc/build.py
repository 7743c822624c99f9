#!/usr/bin/env python3
"""Build libdpm_hip.so (gfx950) in-tree with hipcc.  `python deeppointmap_amd/csrc/build.py`.

hipcc cross-compiles without a GPU.  Everything is compiled with -ffp-contract=off: the distance
arithmetic of FPS / kNN / interpolation must round exactly like the reference's; kernels use
explicit fmaf() wherever fusion is wanted.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libdpm_hip.so")
ARCH = "gfx950"
# -packed-fp32-ops off: no v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 anywhere.  Round 4: with the packed forms compiled in
# (hipcc 7.2 packs adjacent fp32 multiply-adds on its own; the encoder's first-level gather kernel held 52 of them) that kernel
# returned a few wrong maxima in a few rows in up to 40 % of its launches WHILE ANOTHER WAVE ON THE CHIP EXECUTED bf16 MATRIX
# INSTRUCTIONS -- the bf16x3 GEMM on another stream, in another process on the same GPU, even a register-only MFMA
# micro-benchmark in another process -- and never otherwise (fp32 MFMA neighbours: 0 of 2 400 passes).  Same source without
# the packed forms: 0 of 1 800 passes, 0 of 15 runs of the three-process test that had failed one time in five
# (scripts/debug/enc_stress.py, enc_stress_streams.py, sa0_forensics.py).  NOT understood further: a register-only
# v_pk_fma_f32 loop next to MFMA waves computes correctly (scripts/micro/pk_vs_mfma.hip), longer wait states in front of the
# kernel's hand-written DPP steps change nothing -- it is the compiled sequence around the packed forms, not the instruction
# alone.  They buy 1.15-1.2x on the multiply-adds they cover (scripts/micro/pk_fma_rate.hip) and nothing in the pipelined bench
# (4.74 ms per step without them, 4.79 with, fp32 GEMMs both times).
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
SOURCES = {
    "fps.hip": [],
    "fps_tree.hip": [],
    "knn.hip": [],
    "encoder_ops.hip": [],
    "gemm.hip": [],
    "gemm_b3.hip": [],
    "group_mlp.hip": [],
    "decoder_ops.hip": [],
    "match.hip": [],
    "infomat.hip": [],
    "preprocess.hip": [],
    "voxel_sample.hip": [],
    "posegraph.hip": [],
}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, flags=(), out=None):
    """flags / out: an experimental build with extra compiler flags into a library of its own (objects under build/<name>/;
    `DPM_LIB=<out>` selects it at run time: A/B measurements of one tree in one GPU session)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build") if out is None else os.path.join(HERE, "build", os.path.basename(out) + ".d")
    OUT = globals()["OUT"] if out is None else out
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "dpm_hip.h"))
    jobs = []
    for src, extra in SOURCES.items():
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *COMMON, *extra, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT, *objs])
    return OUT


def build_abi_smoke(verbose=False):
    """tests/c/abi_gpu_smoke.cpp -> csrc/build/abi_gpu_smoke: the torch-free HIP host program that drives the C ABI (run by
    tests/test_gpu_ops.py).  Built HERE, next to the library, so that the GPU box only runs it: a first hipcc invocation on a
    fresh box takes minutes while the toolchain pages in."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    root = os.path.dirname(os.path.dirname(HERE))
    src = os.path.join(root, "tests", "c", "abi_gpu_smoke.cpp")
    exe = os.path.join(HERE, "build", "abi_gpu_smoke")
    if not os.path.exists(src):
        return None
    if _stale(exe, [src, OUT, os.path.join(root, "include", "dpm_hip.h")]):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O2", "-ffp-contract=off", "-I" + os.path.join(root, "include"), src, "-o", exe,
               "-L" + os.path.dirname(OUT), "-ldpm_hip", "-Wl,-rpath,$ORIGIN/../.."]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    if "--out" in sys.argv:   # python build.py --out <lib.so> [extra compiler flags ...]
        i = sys.argv.index("--out")
        print(build(verbose=True, flags=tuple(a for a in sys.argv[i + 2:]), out=os.path.abspath(sys.argv[i + 1])))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_abi_smoke(verbose=True))
