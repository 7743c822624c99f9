"""GPU: the whole hot path at BASELINE.json's full size -- encode two 65 536-point scans, register them,
build the information matrix -- against the pose the REFERENCE computed on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import T, assert_features_close, load_golden, rot_angle
from oracle import dpm_oracle as O
from deeppointmap_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hot(cfg_full):
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.pipeline import HotPath
    from deeppointmap_amd.weights import init_procedural
    return HotPath(init_procedural(Encoder(cfg_full)).to(DEV), init_procedural(Decoder(cfg_full)).to(DEV))


def test_full_size_pose_parity_vs_reference(hot):
    g = load_golden("decoder.npz")
    pts, pad = synthetic.frames(2, 65536)
    pcd_m = (pts * 60).to(DEV)
    desc, edges, table = hot.step(pts.to(DEV), pad.to(DEV), pcd_m)
    # descriptors of both frames equal the reference's (the registration fixture stores them)
    assert np.array_equal(desc[0, 128:].cpu().numpy(), g["synthetic01.src_desc"][128:])     # key points in metres: exact
    assert_features_close(desc[0, :128].cpu().numpy(), g["synthetic01.src_desc"][:128], "pipeline frame 0 descriptors")
    assert_features_close(desc[1, :128].cpu().numpy(), g["synthetic01.dst_desc"][:128], "pipeline frame 1 descriptors")
    e = edges[1]  # frame 0 -> frame 1
    dT = float((e.T.cpu() - T(g["synthetic01.T"])).norm())
    dR = rot_angle(e.R.cpu(), g["synthetic01.R"])
    assert dT < 1e-4 and dR < 1e-4, (dT, dR)   # north_star tolerance: 1e-4 m / 1e-4 rad
    assert e.conf.shape[0] == g["synthetic01.conf"].shape[0]
    assert abs(e.rmse - float(g["synthetic01.rmse"])) < 1e-3
    # edge table row = [registration header | information matrix], written by the kernels themselves
    row = table[1].cpu()
    assert torch.equal(row[:9].view(3, 3), e.R.cpu()) and torch.equal(row[9:12].view(3, 1), e.T.cpu())
    assert abs(float(row[12]) - e.rmse) < 1e-7 and int(row[14]) == e.conf.shape[0]
    assert abs(float(row[16]) - float(e.conf[:30].mean())) < 1e-9
    G = O.information_matrix(pts[0] * 60, pts[1] * 60, O.se3(e.R.cpu(), e.T.cpu()))
    np.testing.assert_allclose(row[20:].view(6, 6).numpy(), G.numpy(), rtol=2e-4, atol=2e-4 * float(G.abs().max()))


def test_size_independent_properties(hot):
    # permuting the points of a scan permutes nothing downstream except FPS's start point (index 0):
    # keep point 0 fixed, shuffle the rest -> identical keypoint SET and descriptors up to ordering
    pts, pad = synthetic.frames(1, 65536)
    perm = torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(65535, generator=torch.Generator().manual_seed(0))])
    d0 = hot.extract(pts.to(DEV), pad.to(DEV))[0].cpu()
    d1 = hot.extract(pts[:, :, perm].contiguous().to(DEV), pad.to(DEV))[0].cpu()
    k0 = d0[128:].t().contiguous().numpy()
    k1 = d1[128:].t().contiguous().numpy()
    assert np.array_equal(k0, k1), "FPS picks depend on geometry only (ties aside), not on storage order"
    # features: which of two EQUIDISTANT neighbours is grouped at the K-th slot follows the reference's
    # heap-select and therefore the storage order, so a handful of rows may differ in one neighbour
    diff = np.abs(d0[:128].numpy() - d1[:128].numpy())
    assert diff.max() < 1e-2 and (diff > 3e-4).mean() < 0.15
    # registering a scan against itself gives the identity
    edges, _ = hot.register(torch.stack([d0, d0]).to(DEV), None, [(0, 1)])
    assert float((edges[0].R.cpu() - torch.eye(3)).abs().max()) < 1e-4
    assert float(edges[0].T.cpu().abs().max()) < 1e-3


@pytest.mark.parametrize("feature_split", [0, 1, 3])
def test_streaming_pipeline_equals_plain_steps(hot, feature_split):
    # submit/flush (side-stream presampling overlapped with the previous batch) must give bit-identical results -- also with the
    # feature stage cut in two pipeline stages on two streams (Encoder.forward's stop_level / resume)
    from deeppointmap_amd.pipeline import HotPath
    hot = HotPath(hot.encoder, hot.decoder)
    hot.feature_split = feature_split
    batches = []
    for i in range(5):
        pts, pad = synthetic.frames(2, 16384, start=2 * i)
        batches.append((pts.to(DEV), pad.to(DEV), (pts * 60).to(DEV)))
    plain = []
    for p, m, q in batches:
        desc, _, table = hot.step(p, m, q, materialize=False)
        plain.append((desc.clone(), table.clone()))
    outs = []
    for p, m, q in batches:
        r = hot.submit(p, m, q)
        if r is not None:
            outs.append(r)
    outs.extend(hot.flush())
    assert hot.flush() == [] and len(outs) == 5
    torch.cuda.synchronize()
    for (d0, t0), (d1, t1) in zip(plain, outs):
        assert torch.equal(d0, d1) and torch.equal(t0, t1)
    # resident inputs (complete before submit): the geometry streams need not wait for the caller's stream (round 6)
    torch.cuda.synchronize()
    hot.inputs_on_caller_stream = False
    outs = []
    for p, m, q in batches:
        r = hot.submit(p, m, q)
        if r is not None:
            outs.append(r)
    outs.extend(hot.flush())
    torch.cuda.synchronize()
    for (d0, t0), (d1, t1) in zip(plain, outs):
        assert torch.equal(d0, d1) and torch.equal(t0, t1)


def test_streaming_pipeline_survives_callers_that_drop_their_inputs(hot):
    """submit() returns while three streams still read the batch: a caller that frees its tensors and allocates new ones
    right away must not get its memory recycled under a running stage (the inputs are put on record with every stream
    that reads them).  Varying batch shapes, ring and chain mode; every result bit-equal to an unpipelined step.
    (scripts/fuzz_pipeline.py found 4 % wrong information matrices before the record_stream calls were there.)"""
    import random
    from deeppointmap_amd.pipeline import HotPath
    rng = random.Random(5)
    for chain in (False, True):
        pipe, plain = HotPath(hot.encoder, hot.decoder), HotPath(hot.encoder, hot.decoder)
        pipe.chain = plain.chain = chain
        inputs, outs = [], []
        for i in range(10):
            F, N = rng.randint(2, 4), rng.choice([16384, 20000, 24000])
            pts, pad = synthetic.frames(F, N, start=3 * i)
            if i % 3 == 1:
                pad[0, N - 3000:] = True
                pts[0, :, N - 3000:] = 0
            inputs.append((pts, pad))
            p, q = pts.to(DEV), pad.to(DEV)
            m = (p * 60).contiguous()
            r = pipe.submit(p, q, m)
            del p, q, m
            junk = torch.empty(rng.randint(8, 96) << 20, device=DEV).normal_()   # takes over whatever was just freed
            del junk
            if r is not None:
                outs.append((r[0].clone(), r[1].clone()))
        outs += [(d.clone(), t.clone()) for d, t in pipe.flush()]
        torch.cuda.synchronize()
        assert len(outs) == len(inputs)
        for (pts, pad), (d, t) in zip(inputs, outs):
            p = pts.to(DEV)
            d0, _, t0 = plain.step(p, pad.to(DEV), (p * 60).contiguous(), materialize=False)
            assert torch.equal(d, d0) and torch.equal(t, t0), chain
    # a chain whose predecessor came without its scan cannot build the first edge's information matrix: say so
    pipe = HotPath(hot.encoder, hot.decoder)
    pipe.chain = True
    pts, pad = synthetic.frames(2, 16384)
    pipe.step(pts.to(DEV), pad.to(DEV), None, materialize=False)
    with pytest.raises(ValueError, match="chain mode"):
        pipe.step(pts.to(DEV), pad.to(DEV), (pts * 60).to(DEV), materialize=False)


def test_five_consecutive_full_size_pairs_vs_reference(hot):
    """BASELINE.json config 2 in miniature: six 65 536-point scans, five odometry edges, every pose within the
    north_star tolerance of what the REFERENCE computes end to end (tests/golden/poses_full.npz)."""
    g = load_golden("poses_full.npz")
    pts, pad = synthetic.frames(6, 65536)
    desc = hot.extract(pts.to(DEV), pad.to(DEV))
    edges, table = hot.register(desc, (pts * 60).to(DEV), [(f - 1, f) for f in range(1, 6)])
    worst_t = worst_r = 0.0
    for e in edges:
        k = f"pair{e.src}_{e.dst}"
        worst_t = max(worst_t, float((e.T.cpu() - T(g[k + ".T"])).norm()))
        worst_r = max(worst_r, rot_angle(e.R.cpu(), g[k + ".R"]))
        assert e.conf.shape[0] == int(g[k + ".n_conf"]), k
        assert abs(e.rmse - float(g[k + ".rmse"])) < 1e-3, k
        assert abs(float(e.conf[:30].mean()) - float(g[k + ".conf30"])) < 1e-6 * 10, k
    assert worst_t < 1e-4 and worst_r < 1e-4, (worst_t, worst_r)


def test_num_sample_variants_vs_reference(hot):
    g = load_golden("poses_full.npz")
    pts, pad = synthetic.frames(2, 65536)
    desc = hot.extract(pts.to(DEV), pad.to(DEV))
    for tag, ns in (("int100", 100), ("float300", 300.0), ("float0.25", 0.25)):
        R, Tt, conf, rmse = hot.decoder.registration_forward(desc[0], desc[1], num_sample=ns)
        assert float((Tt.cpu() - T(g[f"ns_{tag}.T"])).norm()) < 1e-4 and rot_angle(R.cpu(), g[f"ns_{tag}.R"]) < 1e-4, tag
        assert conf.shape[0] == int(g[f"ns_{tag}.n_conf"]), tag


def test_kitti_sample_pair_end_to_end(hot):
    ge, gd = load_golden("encoder_full.npz"), load_golden("decoder.npz")
    pts = torch.stack([T(ge["kitti0.points"]), T(ge["kitti1.points"])])
    desc = hot.extract(pts.to(DEV), torch.zeros(2, pts.shape[2], dtype=torch.bool, device=DEV))
    R, Tt, conf, rmse = hot.decoder.registration_forward(desc[0], desc[1], num_sample=0.5)
    assert float((Tt.cpu() - T(gd["kitti01.T"])).norm()) < 1e-4 and rot_angle(R.cpu(), gd["kitti01.R"]) < 1e-4


def test_decoder_is_reentrant_across_threads(hot):
    """system/core.py:55-57 drives ONE Decoder from three threads; results must not depend on interleaving."""
    import threading
    gd = load_golden("decoder.npz")
    names = ["synthetic01", "kitti01", "map1024_vs_256"]
    want = {n: hot.decoder.registration_forward(T(gd[n + ".src_desc"]), T(gd[n + ".dst_desc"]), num_sample=0.5) for n in names}
    got, errs = {}, []

    def work(n):
        try:
            for _ in range(5):
                got[n] = hot.decoder.registration_forward(T(gd[n + ".src_desc"]), T(gd[n + ".dst_desc"]), num_sample=0.5)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(n,)) for n in names]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for n in names:
        assert torch.equal(got[n][0], want[n][0]) and torch.equal(got[n][1], want[n][1]) and got[n][3] == want[n][3]


def test_registration_graphs_captured_ahead_serve_every_thread(cfg_full):
    """Decoder.capture_registration_graphs: what a caller does before it starts worker threads (SlamSystem.MT_Init; the reference's
    multi-thread mode drives one Decoder from three threads, system/core.py:54-57) -- captures cannot happen once a second thread
    exists.  Three threads then hammer ONE 256 x 256 graph with different inputs: every result bit-equal to the eager path, and the
    replayed call cheaper than the eager one."""
    import threading
    import time
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.weights import init_procedural
    if threading.active_count() > 1:
        pytest.skip("needs a single-threaded process to capture")
    dec = init_procedural(Decoder(cfg_full)).to(DEV)
    gen = torch.Generator().manual_seed(5)

    def pair(M, N):
        mk = lambda n: torch.cat([torch.rand(128, n, generator=gen), (torch.rand(3, n, generator=gen) * 2 - 1) * 40]).to(DEV)
        return mk(M), mk(N)

    inputs = {t: [pair(256, 256) for _ in range(6)] for t in range(3)}
    inputs[1].append(pair(512, 256))                            # a second captured shape in the mix
    dec.graph_min_hits = 0                                      # eager reference values, nothing captured behind the scenes
    want = {t: [dec.registration_forward(s, d, num_sample=0.5) for s, d in inputs[t]] for t in range(3)}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dec.registration_forward(*inputs[0][0], num_sample=0.5)
    eager_ms = (time.perf_counter() - t0) / 20 * 1e3
    dec.graph_min_hits = 2
    assert dec.capture_registration_graphs([(256, 256, 0.5), (512, 256, 0.5), (1, 1, 0.5)]) == 2   # k = 0 for 1 x 1: nothing to capture
    assert dec.capture_registration_graphs([(256, 256, 0.5)]) == 0                                 # already there
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dec.registration_forward(*inputs[0][0], num_sample=0.5)
    replay_ms = (time.perf_counter() - t0) / 20 * 1e3
    got, errs = {}, []

    def work(t):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
                for _ in range(4):
                    got[t] = [dec.registration_forward(s, d, num_sample=0.5) for s, d in inputs[t]]
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    mt_ms = (time.perf_counter() - t0) / (4 * sum(len(v) for v in inputs.values())) * 1e3
    assert not errs, errs
    for t in range(3):
        for (R, T_, c, rmse), (Rw, Tw, cw, rw) in zip(got[t], want[t]):
            assert torch.equal(R, Rw) and torch.equal(T_, Tw) and torch.equal(c, cw) and rmse == rw
    print(f"one-pair registration 256 x 256: eager {eager_ms:.3f} ms, replayed {replay_ms:.3f} ms, three threads {mt_ms:.3f} ms per call")
    assert replay_ms < eager_ms
    # three instances per shape: the threads replay side by side instead of queueing behind one graph; same bits
    assert dec.capture_registration_graphs([(256, 256, 0.5)], copies=3) == 2       # instance 0 exists
    got.clear()
    ts = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    mt3_ms = (time.perf_counter() - t0) / (4 * sum(len(v) for v in inputs.values())) * 1e3
    assert not errs, errs
    for t in range(3):
        for (R, T_, c, rmse), (Rw, Tw, cw, rw) in zip(got[t], want[t]):
            assert torch.equal(R, Rw) and torch.equal(T_, Tw) and torch.equal(c, cw) and rmse == rw
    print(f"   ... with three graph instances per shape: {mt3_ms:.3f} ms per call")
    with pytest.raises(RuntimeError):   # the capture call itself refuses once a thread exists
        ev = threading.Event()
        th = threading.Thread(target=ev.wait)
        th.start()
        try:
            dec.capture_registration_graphs([(128, 128, 0.5)])
        finally:
            ev.set(), th.join()
