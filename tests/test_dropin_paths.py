"""CPU: the reference's import paths resolve to the MI355X classes when deeppointmap_amd/dropin leads sys.path."""
import os
import subprocess
import sys

from conftest import ROOT


def test_reference_import_paths_resolve_to_our_classes():
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(1, %r)\n"
        "from network.encoder.encoder import Encoder\n"
        "from network.decoder.decoder import Decoder\n"
        "from system.modules.utils import calculate_information_matrix_from_pcd, PoseTool, simvec_to_num\n"
        "import deeppointmap_amd.encoder as e, deeppointmap_amd.decoder as d, deeppointmap_amd.registration as r\n"
        "assert Encoder is e.Encoder and Decoder is d.Decoder\n"
        "assert calculate_information_matrix_from_pcd is r.calculate_information_matrix_from_pcd\n"
        "from deeppointmap_amd.config import default_args\n"
        "enc, dec = Encoder(default_args()), Decoder(default_args())\n"
        "assert len(enc.state_dict()) == 110 and len(dec.state_dict()) == 82\n"
        "import torch; assert float(PoseTool.SE3(torch.eye(3), torch.ones(3,1))[0,3]) == 1.0\n"
        "print('ok')\n") % (os.path.join(ROOT, "deeppointmap_amd", "dropin"), ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
