"""reference import path system.modules.utils: the hot-path names come from the MI355X implementation; the
rest of the reference module (EXIT_CODE, Communicate_Module, colour helpers) is control-plane code that stays
with the reference and is re-exported from its file when a reference checkout follows on sys.path."""
import importlib.util
import os
import sys

from deeppointmap_amd.registration import (PoseTool, calculate_information_matrix_from_pcd,  # noqa: F401
                                           simvec_to_num)


def _reference_rest():
    here = os.path.abspath(__file__)
    for p in sys.path:
        cand = os.path.join(p, "system", "modules", "utils.py")
        if os.path.isfile(cand) and os.path.abspath(cand) != here:
            return cand
    return None


_ref = _reference_rest()
if _ref is not None:
    try:
        _spec = importlib.util.spec_from_file_location("_reference_system_modules_utils", _ref)
        _mod = importlib.util.module_from_spec(_spec)
        _spec.loader.exec_module(_mod)  # needs the reference's own dependencies (open3d, matplotlib)
        for _k in ("EXIT_CODE", "Communicate_Module", "agent_color", "agent_color_darker", "coorsys_color"):
            if hasattr(_mod, _k):
                globals()[_k] = getattr(_mod, _k)
    except Exception:  # the control-plane names are optional for the hot path
        pass
