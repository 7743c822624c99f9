"""debug (build container, no GPU): ISA-level variants of ONE kernel for the packed-fp32 corruption hunt.

group_mlp.hip is compiled WITH packed fp32 instructions and WITHOUT any inline asm (-DDPM_DPP_BUILTIN -DDPM_VMAX_BUILTIN:
compiler-generated code only -- that build still fails tests/corun_stress.py), its device assembly is edited inside
group_gather_ln_max_kernel<32,1,true> only, and every variant is assembled into a code object of its own under
deeppointmap_amd/csrc/build/pkiso/<name>.hsaco.  scripts/debug/pk_isa_run.py (GPU box) loads them with hipModuleLoad and runs
the co-residency stress on each.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN12_GLOBAL__N_126group_gather_ln_max_kernelILi32ELi1ELb1EEEvPKfS2_S2_S2_S2_PKiS2_iS2_S2_iiixifPf"
OUT = os.path.join(ROOT, "deeppointmap_amd", "csrc", "build", "pkiso")
PK = ["-Xclang", "-target-feature", "-Xclang", "+packed-fp32-ops"]

PAIR = re.compile(r"v\[(\d+):(\d+)\]")
LIST = lambda s: [int(x) for x in s.strip("[]").split(",")]


def compile_s(extra, name):
    s = os.path.join(OUT, name + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", *extra,
                           "-S", "--cuda-device-only", "-o", s, os.path.join(ROOT, "deeppointmap_amd", "csrc", "group_mlp.hip")],
                          stderr=subprocess.DEVNULL)
    return open(s).read().split("\n")


def kernel_span(lines):
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    return a, b


def split_pk(line):
    """one v_pk_{fma,add,mul}_f32 on VGPR pairs -> the two scalar instructions, or None when the form is not handled"""
    m = re.match(r"\s*(v_pk_fma_f32|v_pk_add_f32|v_pk_mul_f32)\s+(.*)", line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    mods = dict(re.findall(r"(op_sel_hi|op_sel|neg_lo|neg_hi):(\[[\d,]+\])", rest))
    regs = PAIR.findall(rest.split(" op_sel")[0].split(" neg_")[0])
    nsrc = 3 if op == "v_pk_fma_f32" else 2
    if len(regs) != nsrc + 1 or len(rest.split(" op_sel")[0].split(" neg_")[0].split(",")) != nsrc + 1:
        return None
    regs = [(int(a), int(b)) for a, b in regs]
    d, srcs = regs[0], regs[1:]
    sel = LIST(mods.get("op_sel", "[" + ",".join(["0"] * nsrc) + "]"))
    selh = LIST(mods.get("op_sel_hi", "[" + ",".join(["1"] * nsrc) + "]"))
    nlo = LIST(mods.get("neg_lo", "[" + ",".join(["0"] * nsrc) + "]"))
    nhi = LIST(mods.get("neg_hi", "[" + ",".join(["0"] * nsrc) + "]"))
    sop = {"v_pk_fma_f32": "v_fma_f32", "v_pk_add_f32": "v_add_f32", "v_pk_mul_f32": "v_mul_f32"}[op]
    lo_src = [srcs[i][sel[i]] for i in range(nsrc)]
    hi_src = [srcs[i][selh[i]] for i in range(nsrc)]
    fmt = lambda dst, ss, neg: f"\t{sop} v{dst}, " + ", ".join(("-" if n else "") + f"v{r}" for r, n in zip(ss, neg))
    lo, hi = fmt(d[0], lo_src, nlo), fmt(d[1], hi_src, nhi)
    if d[0] not in hi_src:
        return [lo, hi]
    if d[1] not in lo_src:
        return [hi, lo]
    return None


def variants(base):
    a, b = kernel_span(base)
    body = base[a:b + 1]
    ispk = lambda l: re.match(r"\s*v_pk_", l) is not None

    def rebuild(new_body):
        return base[:a] + new_body + base[b + 1:]

    def each(fn):
        out = []
        for l in body:
            out.extend(fn(l))
        return rebuild(out)

    def split_where(pred):
        n = [0]

        def fn(l):
            if ispk(l) and pred(l):
                s = split_pk(l)
                if s is not None:
                    n[0] += 1
                    return s
            return [l]
        r = each(fn)
        return r, n[0]

    v = {}
    v["e0_unmodified"] = (base, "packed build, compiler-generated code only")
    v["e1_nop3_after_pk"] = (each(lambda l: [l, "\ts_nop 3"] if ispk(l) else [l]), "s_nop 3 after every v_pk_*")
    v["e2_nop3_before_pk"] = (each(lambda l: ["\ts_nop 3", l] if ispk(l) else [l]), "s_nop 3 before every v_pk_*")
    v["e1b_nop15_after_pk"] = (each(lambda l: [l, "\ts_nop 15"] if ispk(l) else [l]), "s_nop 15 after every v_pk_*")
    for name, pred, what in (("e3_split_all", lambda l: True, "every v_pk_* as two scalar instructions"),
                             ("e4_split_opsel", lambda l: "op_sel:" in l, "v_pk_* with op_sel:[..] (high-half broadcast) as scalar pairs"),
                             ("e5_split_opselhi", lambda l: "op_sel_hi:" in l, "v_pk_* with op_sel_hi:[..] (low-half broadcast) as scalar pairs"),
                             ("e6_split_plain", lambda l: "op_sel" not in l, "v_pk_* without broadcast (v_pk_add_f32) as scalar pairs")):
        r, n = split_where(pred)
        v[name] = (r, f"{what}: {n} of {sum(map(ispk, body))} packed instructions replaced")
    # the high-to-low route taken by another instruction: v_pk_mov_b32 op_sel:[1,0] (a swap of the pair's halves) into a free pair,
    # then the never-failing low-half broadcast form; and the same with a plain v_mov_b32 (no routed operand anywhere)
    def reroute(mov):
        def fn(l):
            m = re.match(r"\s*v_pk_fma_f32 (v\[\d+:\d+\]), (v\[\d+:\d+\]), v\[(\d+):(\d+)\], (v\[\d+:\d+\]) op_sel:\[0,1,0\]\s*$", l)
            if not m:
                return [l]
            d, a, blo, bhi, c = m.groups()
            first = f"\tv_pk_mov_b32 v[94:95], v[{blo}:{bhi}], v[{blo}:{bhi}] op_sel:[1,0]" if mov == "pk" else f"\tv_mov_b32_e32 v94, v{bhi}"
            return [first, "\ts_nop 0", f"\tv_pk_fma_f32 {d}, {a}, v[94:95], {c} op_sel_hi:[1,0,1]"]
        out = [x.replace(".amdhsa_next_free_vgpr 94", ".amdhsa_next_free_vgpr 96") for x in each(fn)]
        return [x.replace(KERNEL + ".num_vgpr, 94", KERNEL + ".num_vgpr, 96") for x in out]
    v["e8_pkmov_swap_then_low_bcast"] = (reroute("pk"), "op_sel:[0,1,0] forms as v_pk_mov_b32 op_sel:[1,0] (swap into v[94:95]) + v_pk_fma_f32 op_sel_hi:[1,0,1]")
    v["e9_mov_then_low_bcast"] = (reroute("mov"), "op_sel:[0,1,0] forms as v_mov_b32 v94, <high half> + v_pk_fma_f32 op_sel_hi:[1,0,1]")
    # waits: every s_waitcnt of the kernel as a full wait
    v["e7_waitcnt_zero"] = (each(lambda l: ["\ts_waitcnt vmcnt(0) lgkmcnt(0)"] if re.match(r"\s*s_waitcnt", l) else [l]),
                            "every s_waitcnt as vmcnt(0) lgkmcnt(0)")
    return v


def main():
    os.makedirs(OUT, exist_ok=True)
    base = compile_s(PK + ["-DDPM_DPP_BUILTIN", "-DDPM_VMAX_BUILTIN"], "base_pk_noasm")
    nopk = compile_s(["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DDPM_DPP_BUILTIN", "-DDPM_VMAX_BUILTIN"], "base_nopk_noasm")
    vs = variants(base)
    vs["n0_nopk"] = (nopk, "the shipped flags (no packed fp32 instructions), no inline asm")
    extra = sys.argv[1:]
    manifest = []
    for name, (lines, what) in vs.items():
        if extra and name not in extra:
            continue
        s = os.path.join(OUT, name + ".s")
        open(s, "w").write("\n".join(lines))
        o = os.path.join(OUT, name + ".o")
        subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
        subprocess.check_call([LLVM + "/ld.lld", "-shared", o, "-o", os.path.join(OUT, name + ".hsaco")])
        os.remove(o)
        manifest.append(f"{name}\t{what}")
        print(name, "--", what)
    open(os.path.join(OUT, "MANIFEST.tsv"), "w").write("\n".join(manifest) + "\n")


if __name__ == "__main__":
    main()
