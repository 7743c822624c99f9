"""Regression guard for the round-4 corruption (csrc/build.py has the history): kernels with hand-written DPP steps must
return the same bits while bf16 matrix instructions of another stream share the chip.  tests/corun_stress.py is the body.

Sensitivity of the guard (recorded in profiles/r05_corun_stress.md): a library built from the same sources WITH packed fp32
instructions (`python deeppointmap_amd/csrc/build.py --out <lib> -Xclang -target-feature -Xclang +packed-fp32-ops`,
`DPM_LIB=<lib> python tests/corun_stress.py`) fails it; the shipped library must not.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(noise, iters, extra_env=None, only=()):
    env = {k: v for k, v in os.environ.items() if not k.startswith("DPM_")}   # the shipped library, no knobs
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "corun_stress.py"), str(iters), noise, *only], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_dpp_kernels_next_to_bf16_matrix_instructions_are_bit_stable():
    res = _run("bf16x3", 300)
    bad = {k: v for k, v in res["differing_calls"].items() if not k.startswith("_") and v}
    assert res["differing_calls"]["_noise_launches"] >= 300, f"the noise stream hardly ran: {res}"
    assert not bad, f"results changed while bf16 MFMAs shared the chip (of {res['iters']} calls each): {bad}"


def test_dpp_kernels_next_to_fp32_matrix_instructions_are_bit_stable():
    res = _run("fp32", 100)
    bad = {k: v for k, v in res["differing_calls"].items() if not k.startswith("_") and v}
    assert not bad, f"results changed while fp32 MFMAs shared the chip (of {res['iters']} calls each): {bad}"


def test_gathers_sharing_compute_units_with_bf16_matrix_instructions_are_bit_stable():
    """both streams confined to the SAME half of the compute units (hipExtStreamCreateWithCUMask): the situation in which a
    packed build of the first-level gather differs in every second launch (profiles/r05_pk_opsel.md); on disjoint compute units
    it never does."""
    res = _run("bf16x3", 1000, {"STRESS_CU_SPLIT": "same"}, only=("gather",))
    if "skipped" in res:
        pytest.skip(res["skipped"])
    bad = {k: v for k, v in res["differing_calls"].items() if not k.startswith("_") and v}
    assert res["cu_split"] == "same" and res["differing_calls"]["_noise_launches"] >= 300, res
    assert not bad, f"results changed while bf16 MFMAs shared the compute units (of {res['iters']} calls each): {bad}"
