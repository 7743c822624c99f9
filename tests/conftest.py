import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

torch.set_grad_enabled(False)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a GPU and the built library: skip them (rather than error) where either is missing"""
    so = os.path.join(ROOT, "deeppointmap_amd", "libdpm_hip.so")
    why = None
    if not torch.cuda.is_available():
        why = "no GPU visible"
    elif not os.path.exists(so):
        why = "deeppointmap_amd/libdpm_hip.so is not built (python __graft_entry__.py)"
    if why:
        skip = pytest.mark.skip(reason=why)
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="session")
def cfg_full():
    from deeppointmap_amd.config import default_args
    return default_args()


@pytest.fixture(scope="session")
def cfg_reduced():
    from deeppointmap_amd.config import reduced_args
    return reduced_args()


@pytest.fixture(scope="session")
def sd_enc(cfg_full):
    from deeppointmap_amd.params import encoder_shapes
    from deeppointmap_amd.weights import procedural_state_dict
    return procedural_state_dict(encoder_shapes(cfg_full))


@pytest.fixture(scope="session")
def sd_dec(cfg_full):
    from deeppointmap_amd.params import decoder_shapes
    from deeppointmap_amd.weights import procedural_state_dict
    return procedural_state_dict(decoder_shapes(cfg_full))


def rot_angle(Ra, Rb):
    """geodesic angle (rad) between two rotation(-like) 3x3 matrices."""
    M = (torch.as_tensor(Ra).double().T @ torch.as_tensor(Rb).double())
    # atan2(sin, cos): arccos alone loses everything below sqrt(eps_f32) ~ 3e-4 rad near identity
    sin = float(torch.linalg.norm(M - M.T)) / (2 * 2 ** 0.5)
    return float(np.arctan2(sin, float((torch.trace(M) - 1) / 2)))


def idx_rows_equal_as_sets(a, b):
    """(S,K) index arrays: per-row set equality -> bool (S,)."""
    a = np.sort(np.asarray(a), axis=1)
    b = np.sort(np.asarray(b), axis=1)
    return (a == b).all(axis=1)
