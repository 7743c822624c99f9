"""MT extractor contract (reference system/core.py:134-185) and batched scan pre-processing on the HIP path."""
import queue
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, T, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_mt_extractor_batches_in_order(cfg_reduced):
    from deeppointmap_amd import synthetic
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.extractor import MTExtractor
    from deeppointmap_amd.weights import init_procedural
    enc = init_procedural(Encoder(cfg_reduced)).to(DEV)
    ex = MTExtractor(enc, coor_scale=60.0)
    pts, pad = synthetic.frames(70, 4096)
    pad[5, 3000:] = True           # a padded scan inside a batch
    pts[5, :, 3000:] = 0.0
    # the synchronous contract: ExtractionThread.process on a concatenated batch
    want = ex.process(pts[:40].contiguous(), pad[:40].contiguous()).cpu()   # > 32 is allowed here; the queue path caps it
    coor, fea, _ = enc(pts[:3], pad[:3])
    assert torch.equal(want[:3], torch.cat([fea, coor * 60.0], dim=1).cpu())
    # the thread body over a queue: 70 scans, an intermediate exit code, the final one
    EXIT_SOFT, EXIT_FINAL = "soft-exit", "exit"
    qi, qo = queue.Queue(), queue.Queue()
    for i in range(70):
        qi.put((0.1 * i, pts[i:i + 1], np.eye(3), np.zeros(3), pad[i:i + 1], None))
        if i == 33:
            qi.put(EXIT_SOFT)
    qi.put(EXIT_FINAL)
    sizes = []
    drain0 = ex.drain

    def drain(q, is_exit):
        scans, ctrl = drain0(q, is_exit)
        sizes.append(len(scans))
        return scans, ctrl
    ex.drain = drain
    ex.run(qi, qo, make_scan=lambda item, d: ("scan", item[0], d), is_exit=lambda x: isinstance(x, str),
           is_final=lambda x: x == EXIT_FINAL)
    out = []
    while not qo.empty():
        out.append(qo.get())
    scans = [o for o in out if not isinstance(o, str)]
    assert max(sizes) <= 32 and sum(sizes) == 70
    assert [round(s[1] * 10) for s in scans] == list(range(70))          # arrival order kept
    assert [o for o in out if isinstance(o, str)] == [EXIT_SOFT, EXIT_FINAL]
    got = torch.stack([s[2] for s in scans[:40]])
    assert got.shape == (40, 131, want.shape[2]) and torch.equal(got, want)  # batch composition does not change a frame's result


def test_preprocess_scans_batch_equals_single():
    from deeppointmap_amd.preprocess import preprocess_scan, preprocess_scans
    sys.path.insert(0, GOLDEN)
    from raw_scan import raw_scan
    g = load_golden("preprocess.npz")
    base = raw_scan()
    gen = torch.Generator().manual_seed(2)
    scans = [base, base[:90000] + 0.05 * torch.randn(90000, 3, generator=gen), T(g["kitti0_m.in"]), base[::3].contiguous(), base * 0.5]
    pts, pad, lens = preprocess_scans(scans, padding_to=-1, streams=3)
    assert pts.shape[0] == 5 and pts.shape[2] == max(lens) and pad.shape == (5, max(lens))
    for b, s in enumerate(scans):
        p1, m1 = preprocess_scan(s)
        n = p1.shape[2]
        assert lens[b] == n and torch.equal(pts[b, :, :n], p1[0]) and not bool(pad[b, :n].any()) and bool(pad[b, n:].all())
        assert float(pts[b, :, n:].abs().max()) == 0.0 if n < pts.shape[2] else True
    assert np.array_equal(pts[0, :, :lens[0]].t().cpu().numpy(), g["raw120k.out"])   # = the reference's own transforms
    p2, m2, l2 = preprocess_scans(scans[:2], padding_to=100000)
    assert p2.shape == (2, 3, 100000) and l2 == lens[:2]
    with pytest.raises(ValueError):
        preprocess_scans(scans[:1], padding_to=100)
