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
# -packed-fp32-ops off: no v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 anywhere (csrc/isa_lint.py checks the linked library).
# hipcc 7.2 packs adjacent fp32 multiply-adds on its own, and on MI355X a v_pk_fma_f32 whose op_sel routes the HIGH dword of a
# 64-bit source to the LOW lane (`op_sel:[0,1,0]`: the compiler's broadcast of the second of two adjacent registers) returns wrong
# results while other waves of the SAME compute unit execute bf16 matrix instructions.  Round 4 met it as a few wrong maxima of the
# encoder's first-level gather in up to 40 % of its launches next to the bf16x3 GEMM; round 5 bisected it (profiles/r05_pk_opsel.md):
#   * compiler-generated code only -- a build of that kernel without any inline asm fails the same way; wait states around the
#     packed instructions (s_nop 3 / 15 before or after each) and full s_waitcnt everywhere change nothing;
#   * of the kernel's 26 packed instructions the four `op_sel:[0,1,0]` ones are the cause: replaced by scalar pairs 0 of 1000
#     launches differ, with them (everything else scalar or not) ~900 of 1000 (both streams confined to the same compute units;
#     on disjoint compute units 0 of 3000; fp32 MFMA neighbours: 0);
#   * scripts/micro/pk_opsel_vs_mfma.hip shows the event in isolation (v_pk_mul_f32 op_sel:[0,1]: sixteen lanes of a wave read the
#     routed operand as 0), scripts/debug/pk_isa_variants.py + pk_isa_run.py are the ISA bisection, tests/test_gpu_corun_stress.py
#     is the regression guard (the shipped library must be bit-stable next to the bf16x3 GEMM; a packed build fails it).
# The packed forms bought 1.15-1.2x on the multiply-adds they cover (scripts/micro/pk_fma_rate.hip) and nothing in the pipelined
# bench (4.74 ms per step without them, 4.79 with, fp32 GEMMs both times).
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


def build(force=False, verbose=False, flags=(), out=None, only=None):
    """flags / out: an experimental build with extra compiler flags into a library of its own (objects under build/<name>/;
    `DPM_LIB=<out>` selects it at run time: A/B measurements of one tree in one GPU session).  only: the sources the extra
    flags apply to (the rest of that library is compiled like the shipped one)."""
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
            jobs.append([hipcc, *COMMON, *extra, *(flags if only is None or src in only else ()), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        # link beside the target and rename only after the lint: a library that failed the lint must not be what the next
        # build() finds up to date
        staged = OUT + ".unlinted"
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", staged, *objs])
        sys.path.insert(0, HERE)
        import isa_lint
        try:
            if out is None:
                isa_lint.check(staged)    # the shipped library: no packed fp32 instructions (see COMMON)
            elif verbose:
                print("packed fp32 instructions:", isa_lint.packed_fp32(staged))
        except BaseException:
            os.remove(staged)
            raise
        os.replace(staged, OUT)
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
    if "--out" in sys.argv:   # python build.py --out <lib.so> [--only a.hip,b.hip] [extra compiler flags ...]
        i = sys.argv.index("--out")
        rest, only = sys.argv[i + 2:], None
        if rest[:1] == ["--only"]:
            only, rest = set(rest[1].split(",")), rest[2:]
            assert only <= set(SOURCES), only
        print(build(verbose=True, flags=tuple(rest), out=os.path.abspath(sys.argv[i + 1]), only=only))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_abi_smoke(verbose=True))
