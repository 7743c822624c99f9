"""ISA lint of the built library: no packed fp32 vector instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32
are what hipcc forms from adjacent fp32 operations unless told not to).

Why (profiles/r05_pk_opsel.md): on MI355X a v_pk_fma_f32 whose op_sel routes the HIGH dword of a 64-bit
source to the LOW lane (`op_sel:[0,1,0]`: how the compiler broadcasts the second of two adjacent registers) returns wrong results
while other waves of the same compute unit execute bf16 matrix instructions -- compiler-generated code, no inline asm involved, no
wait state helps; found by ISA bisection of the encoder's first-level gather kernel.  The library is therefore compiled with
`-target-feature -packed-fp32-ops` (csrc/build.py) and this check keeps it that way: csrc/build.py runs it after linking, and
tests/test_isa_lint.py on the CPU.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile



def _find_objdump() -> str:
    """llvm-objdump of the toolchain that built the library: next to $HIPCC's clang, under /opt/rocm, or on PATH"""
    hipcc = os.path.realpath(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    rocm = os.path.dirname(os.path.dirname(hipcc))
    for cand in (os.path.join(rocm, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(os.path.dirname(hipcc), "llvm-objdump"),
                 "/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("llvm-objdump not found (looked beside $HIPCC, under /opt/rocm/lib/llvm/bin and on PATH): the library cannot be linted")


PACKED = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b(.*)")


def device_disassembly(lib_path: str):
    """yields the disassembly lines of every gfx950 code object bundled in the shared library"""
    OBJDUMP = _find_objdump()
    tmp = tempfile.mkdtemp(prefix="dpm_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(f for f in os.listdir(tmp) if "hipv4-amdgcn" in f)
        if not objs:
            raise RuntimeError(f"{lib_path}: no gfx950 code object found")
        for f in objs:
            out = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            yield from out.splitlines()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def packed_fp32(lib_path: str) -> dict:
    """-> {'v_pk_fma_f32': n, ..., 'op_sel high-to-low': n}: packed fp32 arithmetic in the library's device code"""
    counts = {}
    for line in device_disassembly(lib_path):
        m = PACKED.search(line)
        if not m:
            continue
        counts[m.group(1)] = counts.get(m.group(1), 0) + 1
        sel = re.search(r"op_sel:\[([\d,]+)\]", m.group(2))
        if sel and "1" in sel.group(1):
            counts["op_sel high-to-low"] = counts.get("op_sel high-to-low", 0) + 1
    return counts


# the erratum is about 32-bit operands routed across the halves of a 64-bit register pair: instructions on fp32 / 32-bit packed data.
# 16-bit forms (v_pk_*_f16 / _bf16 / _i16 / _u16, v_mad_mix*, v_cvt_*) use op_sel to pick halves of ONE dword -- another mechanism,
# reported by `routed_operands(..., all_forms=True)` but not a build failure.
ROUTED = re.compile(r"\b(v_\w+)\b[^/]*\bop_sel:\[([\d,]+)\]")
ROUTED_32 = re.compile(r"^v_pk_\w+_(?:f32|b32|u32|i32)$")
ROUTED_OK = {"v_pk_mov_b32"}   # measured harmless in the failing kernel's place (profiles/r05_pk_opsel.md, variant e8)


def routed_operands(lib_path: str, all_forms: bool = False) -> dict:
    """-> {instruction: n} for every packed 32-bit vector instruction with a high-half op_sel outside ROUTED_OK: the operand routing
    the erratum was found on (all_forms: every vector instruction with one, 16-bit forms included).  Nothing in the library has
    one; a new one should be looked at before it ships."""
    counts = {}
    for line in device_disassembly(lib_path):
        m = ROUTED.search(line)
        if m and "1" in m.group(2) and m.group(1) not in ROUTED_OK and (all_forms or ROUTED_32.match(m.group(1))):
            counts[m.group(1)] = counts.get(m.group(1), 0) + 1
    return counts


def check(lib_path: str) -> None:
    routed = routed_operands(lib_path)
    if routed:
        raise RuntimeError(f"{lib_path}: vector instructions with a high-half op_sel {routed}: not measured next to bf16 matrix "
                           "instructions yet (csrc/isa_lint.py, profiles/r05_pk_opsel.md)")
    found = packed_fp32(lib_path)
    if found:
        raise RuntimeError(f"{lib_path} contains packed fp32 instructions {found}: they return wrong results next to bf16 matrix "
                           "instructions on MI355X (csrc/isa_lint.py); compile with -Xclang -target-feature -Xclang -packed-fp32-ops")


if __name__ == "__main__":
    import sys
    print(packed_fp32(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libdpm_hip.so")))
