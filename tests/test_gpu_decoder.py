"""GPU: decoder / registration kernels and the Decoder module against reference fixtures and the oracle."""
import numpy as np
import pytest
import torch

from conftest import T, assert_features_close, load_golden, rot_angle
from oracle import dpm_oracle as O
from deeppointmap_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from deeppointmap_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def dec(cfg_full):
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.weights import init_procedural
    return init_procedural(Decoder(cfg_full)).to(DEV)


def test_position_embedding_vs_reference(dec, ops):
    g = load_golden("decoder.npz")
    xyz = T(g["posemb.xyz"]).to(DEV)
    out = ops.posemb(xyz, dec._dimt(torch.device(DEV)), 256).cpu().numpy()
    np.testing.assert_allclose(out, g["posemb.out"], rtol=0, atol=2e-6)
    assert np.all(out[:, 252:] == 0)


def test_attention_core_vs_torch(ops):
    gen = torch.Generator().manual_seed(3)
    # the last two shapes are large enough for the 32-queries-per-wave instantiation (M % 128 == 0, >= 1024 blocks)
    for B, M, N in [(1, 256, 256), (2, 100, 333), (1, 1024, 256), (3, 17, 5), (64, 256, 256), (70, 128, 200)]:
        q, k, v = (torch.randn(B * n, 256, generator=gen) for n in (M, N, N))
        out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), B, M, N, 8).cpu()
        qh = q.view(B, M, 8, 32).transpose(1, 2).double()
        kh = k.view(B, N, 8, 32).transpose(1, 2).double()
        vh = v.view(B, N, 8, 32).transpose(1, 2).double()
        want = (torch.softmax(qh @ kh.transpose(-1, -2) / 32 ** 0.5, -1) @ vh).transpose(1, 2).reshape(B * M, 256).float()
        torch.testing.assert_close(out, want, rtol=1e-4, atol=2e-5)
    # strided operands: q/k/v as column slices of one (rows, 768) buffer (how the module calls it)
    qkv = torch.randn(256, 768, generator=gen).to(DEV)
    a = ops.attention(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], 1, 256, 256, 8)
    b = ops.attention(qkv[:, :256].contiguous(), qkv[:, 256:512].contiguous(), qkv[:, 512:].contiguous(), 1, 256, 256, 8)
    assert torch.equal(a, b)
    # kv_shift: sequence b reads the keys / values of sequence (b + shift) mod B -- both directions of a cross
    # attention in one launch must equal the two explicit launches bit for bit
    B, M = 6, 128
    qkv = torch.randn(B * M, 768, generator=gen).to(DEV)
    one = ops.attention(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], B, M, M, 8, kv_shift=B // 2)
    R = B // 2 * M
    two = torch.cat([ops.attention(qkv[:R, :256], qkv[R:, 256:512], qkv[R:, 512:], B // 2, M, M, 8),
                     ops.attention(qkv[R:, :256], qkv[:R, 256:512], qkv[:R, 512:], B // 2, M, M, 8)])
    assert torch.equal(one, two)
    # seq_index: the batch elements are drawn from a smaller set of stored sequences (dpm_attention_indexed) -- equal to the
    # shifted form over the gathered rows, bit for bit, in both wave layouts (16 and 32 queries per wave)
    for U, M, B in ((5, 128, 8), (33, 256, 64)):
        store = torch.randn(U * M, 768, generator=gen).to(DEV)
        seq = torch.randint(0, U, (B,), generator=gen, dtype=torch.int32).to(DEV)
        rows = (seq.long()[:, None] * M + torch.arange(M, device=DEV)[None, :]).reshape(-1)
        gathered = store[rows].contiguous()
        want = ops.attention(gathered[:, :256], gathered[:, 256:512], gathered[:, 512:], B, M, M, 8, kv_shift=B // 2)
        got = ops.attention(store[:, :256], store[:, 256:512], store[:, 512:], B, M, M, 8, kv_shift=B // 2, seq_index=seq)
        assert torch.equal(got, want), (U, M, B)


def test_attention_key_split_vs_torch(ops):
    """few queries against many keys (scan tokens attending a map tile) run key-split (dpm_attention_split): against fp64
    torch, against the plain kernel, ragged last ranges, strided operands and shifted keys; the split count depends on the
    shape of one sequence only, so a batch equals its per-sequence calls bit for bit"""
    gen = torch.Generator().manual_seed(31)
    for B, M, N in [(1, 256, 4096), (2, 100, 1500), (1, 17, 1024), (3, 256, 2049), (1, 1024, 4096)]:
        ns = ops.attention_key_splits(B, M, N, 8, 32)
        assert ns > 1, (B, M, N)
        q, k, v = (torch.randn(B * n, 256, generator=gen) for n in (M, N, N))
        out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), B, M, N, 8).cpu()
        qh = q.view(B, M, 8, 32).transpose(1, 2).double()
        kh = k.view(B, N, 8, 32).transpose(1, 2).double()
        vh = v.view(B, N, 8, 32).transpose(1, 2).double()
        want = (torch.softmax(qh @ kh.transpose(-1, -2) / 32 ** 0.5, -1) @ vh).transpose(1, 2).reshape(B * M, 256).float()
        torch.testing.assert_close(out, want, rtol=1e-4, atol=2e-5)
        if B > 1:   # batch == per-sequence calls
            one = torch.cat([ops.attention(q[b * M:(b + 1) * M].to(DEV), k[b * N:(b + 1) * N].to(DEV), v[b * N:(b + 1) * N].to(DEV),
                                           1, M, N, 8).cpu() for b in range(B)])
            assert torch.equal(out, one)
    assert ops.attention_key_splits(64, 256, 256, 8, 32) == 1 and ops.attention_key_splits(1, 4096, 4096, 8, 32) == 1
    # strided q / k / v (column slices of a fused projection) and kv_shift
    B, M, N = 2, 128, 2048
    qkv = torch.randn(B * N, 768, generator=gen).to(DEV)
    qq = qkv[:B * M, :256]
    a = ops.attention(qq, qkv[:, 256:512], qkv[:, 512:], B, M, N, 8, kv_shift=1)
    want = torch.cat([ops.attention(qq[:M].contiguous(), qkv[N:, 256:512].contiguous(), qkv[N:, 512:].contiguous(), 1, M, N, 8),
                      ops.attention(qq[M:].contiguous(), qkv[:N, 256:512].contiguous(), qkv[:N, 512:].contiguous(), 1, M, N, 8)])
    assert torch.equal(a, want)


def test_attention_other_head_widths_vs_torch(ops):
    """decoder.model_channel other than 256 (head widths 8, 16, 64, 128 at the reference's 8 heads): the generic kernel,
    with key masks and shifted keys; a width without a kernel is refused, not mis-computed."""
    gen = torch.Generator().manual_seed(8)
    for d, B, M, N in [(16, 2, 100, 77), (8, 1, 33, 200), (64, 3, 64, 64), (128, 2, 50, 130)]:
        E = 8 * d
        q, k, v = (torch.randn(B * n, E, generator=gen) for n in (M, N, N))
        mask = torch.rand(B, N, generator=gen) < 0.3
        mask[:, 0] = False
        for km, shift in ((None, 0), (mask, 0), (mask, 1 if B > 1 else 0)):
            out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), B, M, N, 8, kv_shift=shift,
                                key_mask=None if km is None else km.view(torch.uint8).to(DEV)).cpu()
            qh = q.view(B, M, 8, d).transpose(1, 2).double()
            kh = k.view(B, N, 8, d).transpose(1, 2).double().roll(-shift, 0)
            vh = v.view(B, N, 8, d).transpose(1, 2).double().roll(-shift, 0)
            sc = qh @ kh.transpose(-1, -2) / d ** 0.5
            if km is not None:
                sc = sc.masked_fill(km.roll(-shift, 0)[:, None, None, :], float("-inf"))
            want = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B * M, E).float()
            torch.testing.assert_close(out, want, rtol=1e-4, atol=2e-5)
    with pytest.raises(ValueError):
        ops.attention(torch.zeros(4, 96, device=DEV), torch.zeros(4, 96, device=DEV), torch.zeros(4, 96, device=DEV), 1, 4, 4, 8)


def test_dual_softmax_topk_vs_torch(ops):
    gen = torch.Generator().manual_seed(7)
    for M, N, k in [(256, 256, 128), (1024, 256, 640), (300, 77, 1), (64, 64, 4096), (2048, 1024, 1536), (4096, 256, 2176)]:
        a = torch.nn.functional.normalize(torch.randn(M, 64, generator=gen), dim=1)
        b = torch.nn.functional.normalize(torch.randn(N, 64, generator=gen), dim=1)
        S = a @ b.t()
        P = torch.softmax(S / 0.1, 1) * torch.softmax(S / 0.1, 0)
        wv, wi = torch.topk(P.reshape(-1), k)
        Sd = S.clone().to(DEV)
        val, idx = ops.dual_softmax_topk(Sd, 0.1, k)
        torch.testing.assert_close(Sd.cpu(), P, rtol=2e-5, atol=1e-12)
        torch.testing.assert_close(val.cpu(), wv, rtol=2e-5, atol=1e-12)
        assert bool((val[:-1] >= val[1:]).all()), "top-k must come out sorted descending"
        # the GPU's own matrix, re-selected exactly on the host, must give the same index set
        gv, gi = torch.topk(Sd.cpu().reshape(-1), k)
        assert set(idx.cpu().tolist()) == set(gi.tolist())


def test_fused_match_vs_torch_and_vs_the_unfused_operators(ops):
    """dpm_match_topk (similarity -> dual softmax -> top-k in two launches, the M x N matrix never in memory) against torch
    in double and against the five-launch path it replaces: ragged strips (M % 64 != 0), narrow tiles (N < 256, N % 64 != 0),
    one strip, 64 strips (a map tile against a scan), k up to the list capacity, batches."""
    gen = torch.Generator().manual_seed(17)
    for B, M, N, C, k in [(1, 256, 256, 256, 128), (3, 256, 256, 256, 128), (1, 300, 77, 64, 1), (2, 64, 64, 32, 2048),
                          (1, 37, 200, 96, 500), (1, 1024, 256, 256, 640), (1, 4096, 256, 256, 1088), (2, 130, 256, 64, 193)]:
        a = torch.nn.functional.normalize(torch.randn(B, M, C, generator=gen), dim=2)
        b = torch.nn.functional.normalize(torch.randn(B, N, C, generator=gen), dim=2)
        # the operator takes every one of these shapes; the decoder sends it those whose candidate merge is short
        assert ops.match_supported(M, N, C, k) == (-(-M // 64) * min(k, 64 * N) <= ops.MATCH_MAX_MERGE)
        val, idx = ops.match_topk(a.to(DEV), b.to(DEV), 0.1, k)
        S = ops.similarity_batched(a.to(DEV), b.to(DEV))
        v0, i0 = ops.dual_softmax_topk(S, 0.1, k)   # S now holds the unfused path's P
        for p in range(B):
            Sd = a[p].double() @ b[p].double().t()
            P = torch.softmax(Sd / 0.1, 1) * torch.softmax(Sd / 0.1, 0)
            wv, wi = torch.topk(P.reshape(-1), k)
            torch.testing.assert_close(val[p].cpu().double(), wv, rtol=3e-5, atol=1e-12)
            assert bool((val[p, :-1] >= val[p, 1:]).all()), "top-k must come out sorted descending"
            got = idx[p].cpu().long()
            assert len(set(got.tolist())) == k and int(got.min()) >= 0 and int(got.max()) < M * N
            # every selected entry carries the value of its own position (the indices are not merely plausible) ...
            torch.testing.assert_close(val[p].cpu().double(), P.reshape(-1)[got], rtol=3e-5, atol=1e-12)
            # ... and nothing outside the selection beats the smallest selected value by more than rounding
            rest = P.reshape(-1).clone()
            rest[got] = 0
            assert float(rest.max()) <= float(val[p, -1]) * (1 + 3e-5) + 1e-12
        # against the operators it replaces: same values to the last bits of the column sums, same pairs wherever the
        # k-th and (k+1)-th values are further apart than that
        torch.testing.assert_close(val, v0, rtol=2e-6, atol=1e-30)
        Pm = S.view(B, -1)
        for p in range(B):
            srt = torch.sort(Pm[p], descending=True).values
            if k < M * N and float(srt[k - 1] - srt[k]) > 1e-5 * float(srt[k - 1]):
                assert set(idx[p].tolist()) == set(i0[p].tolist()), (B, M, N, k)


def test_fused_match_ties_padding_and_batch_independence(ops):
    """exact ties at the k-th value go to the smaller flat indices (the rule of dpm_dual_softmax_topk); a pair's result
    does not depend on the batch it travels in; rows of zeros (tau so small that most products underflow) still give k
    valid, distinct pairs."""
    gen = torch.Generator().manual_seed(3)
    # duplicated descriptors: rows 0..63 of a repeat rows 64..127, so P has exactly equal entries in pairs of rows
    base = torch.nn.functional.normalize(torch.randn(64, 32, generator=gen), dim=1)
    a = torch.cat([base, base, base]).unsqueeze(0).to(DEV)          # (1,192,32)
    b = torch.nn.functional.normalize(torch.randn(1, 100, 32, generator=gen), dim=2).to(DEV)
    for k in (1, 7, 300, 2048):
        val, idx = ops.match_topk(a, b, 0.1, k)
        S = ops.similarity_batched(a, b)
        v0, i0 = ops.dual_softmax_topk(S, 0.1, k)
        assert torch.equal(idx, i0), k       # identical rows -> identical column folds -> the same tie order to the index
        torch.testing.assert_close(val, v0, rtol=2e-6, atol=0)
    # batch independence, bit for bit
    A = torch.nn.functional.normalize(torch.randn(5, 256, 256, generator=gen), dim=2).to(DEV)
    Bm = torch.nn.functional.normalize(torch.randn(5, 256, 256, generator=gen), dim=2).to(DEV)
    v, i = ops.match_topk(A, Bm, 0.1, 128)
    for p in (0, 3):
        v1, i1 = ops.match_topk(A[p:p + 1].contiguous(), Bm[p:p + 1].contiguous(), 0.1, 128)
        assert torch.equal(v[p], v1[0]) and torch.equal(i[p], i1[0])
    # underflow: tau = 1e-3 leaves a few non-zero products per matrix, the rest are exact zeros tied at the threshold
    v, i = ops.match_topk(A[:2].contiguous(), Bm[:2].contiguous(), 1e-3, 1000)
    S = ops.similarity_batched(A[:2].contiguous(), Bm[:2].contiguous())
    v0, i0 = ops.dual_softmax_topk(S, 1e-3, 1000)
    assert torch.equal(v == 0, v0 == 0)
    for p in range(2):
        assert len(set(i[p].tolist())) == 1000
        nz = int((v0[p] > 0).sum())
        assert set(i[p, :nz].tolist()) == set(i0[p, :nz].tolist())
        assert torch.equal(i[p, nz:], i0[p, nz:])    # among the zeros: the smallest flat indices, ascending


def test_topk_ties_and_exact_values(ops):
    # many equal values: result must still be a valid top-k (multiset of values equal), deterministic
    P = (torch.randint(0, 6, (128, 128)).float() / 8).to(DEV)
    k = 500
    Q = P.clone()
    v1, i1 = ops.dual_softmax_topk(Q, 1.0, k)
    v2, i2 = ops.dual_softmax_topk(P.clone(), 1.0, k)
    assert torch.equal(v1, v2) and torch.equal(i1, i2)  # deterministic under ties
    wv = torch.topk(Q.cpu().reshape(-1), k)[0]
    assert torch.equal(v1.cpu(), wv)  # same multiset of values as an exact host top-k of the same matrix
    assert len(set(i1.cpu().tolist())) == k


@pytest.mark.parametrize("name", ["svd_clean200", "svd_outliers300", "svd_few40", "svd_reflect120", "svd_lowconf100"])
def test_kabsch_loop_vs_reference(name, ops):
    g = load_golden("decoder.npz")
    w, src, dst = T(g[name + ".w"]), T(g[name + ".src"]), T(g[name + ".dst"])
    res = ops.corr_kabsch(None, src.t().contiguous().to(DEV), dst.t().contiguous().to(DEV), None, None,
                          w.to(DEV), 2.0).cpu()
    assert res.dim() == 1
    R, Tt, rmse, n, n_in = res[:9].view(3, 3), res[9:12].view(3, 1), float(res[12]), int(res[13]), int(res[14])
    assert n == w.numel() and n_in == int(g[name + ".mask"].sum())
    np.testing.assert_allclose(R.numpy(), g[name + ".R"], atol=2e-6)
    np.testing.assert_allclose(Tt.numpy(), g[name + ".T"], atol=2e-5)
    assert abs(rmse - float(g[name + ".rmse"])) < 2e-5
    np.testing.assert_allclose(res[20:20 + n_in].numpy(), w.numpy()[g[name + ".mask"]], rtol=0, atol=0)
    assert abs(float(res[16]) - float(w[T(g[name + ".mask"])][:30].mean())) < 1e-6
    if name == "svd_reflect120":
        assert np.linalg.det(R.double().numpy()) < 0  # R = V U^T is left uncorrected (decoder.py:243)


@pytest.mark.parametrize("n", [70, 253, 1000, 4095, 4096, 5000])
def test_kabsch_initial_inliers_with_tied_weights(n, ops):
    """decoder.py:233-235 seeds the inliers with torch.topk(w, 64), and w holds every confidence twice: equal weights
    straddle the 64th place about every other call.  Which of them torch.topk returns is std::nth_element's (n < 4096)
    or std::partial_sort's (n >= 4096) business; the kernel replays it.  The oracle calls torch.topk itself."""
    g = torch.Generator().manual_seed(n)
    src = torch.randn(3, n, generator=g) * 10
    a = 0.2
    R = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
    dst = R @ src + torch.tensor([[0.5], [1.0], [-0.25]]) + 0.3 * torch.randn(3, n, generator=g)
    for levels in (6, 40):  # few distinct values: ties everywhere; more values: the occasional straddling pair
        w = (torch.randint(0, levels, (n // 2 + 1,), generator=g).float() / (2.5 * levels)).sort(descending=True).values
        w = torch.cat([w, w])[torch.randperm(2 * w.numel(), generator=g)[:n].sort().values]  # two descending runs, holes
        Ro, To, inl, rmse = O.solve_svd(w, src, dst)
        res = ops.corr_kabsch(None, src.t().contiguous().to(DEV), dst.t().contiguous().to(DEV), None, None, w.to(DEV), 2.0).cpu()
        n_in = int(res[14])
        assert n_in == int(inl.sum()), (n, levels)
        assert torch.equal(res[20:20 + n_in], w[inl]), (n, levels)
        assert float((res[9:12].view(3, 1) - To).norm()) < 2e-5 and rot_angle(res[:9].view(3, 3), Ro) < 1e-5


@pytest.mark.parametrize("name", ["synthetic01", "kitti01", "map1024_vs_256"])
def test_registration_forward_vs_reference(name, dec):
    g = load_golden("decoder.npz")
    s, d = T(g[name + ".src_desc"]), T(g[name + ".dst_desc"])
    tr = {}
    R, Tt, conf, rmse = dec.registration_forward(s, d, num_sample=0.5, trace=tr)  # CPU inputs, like ScanPack
    assert R.is_cuda and tuple(R.shape) == (3, 3) and tuple(Tt.shape) == (3, 1) and isinstance(rmse, float)
    assert_features_close(tr["x"].cpu().numpy(), g[name + ".src_corr"][:-3].T, f"decoder {name} correlated src features")
    assert_features_close(tr["y"].cpu().numpy(), g[name + ".dst_corr"][:-3].T, f"decoder {name} correlated dst features")
    np.testing.assert_allclose(tr["conf"].cpu().numpy().reshape(-1), g[name + ".pair_conf"], rtol=3e-3, atol=0)
    assert tr["n_corr"] == g[name + ".corr_w"].shape[0]
    # the tolerance north_star states: 1e-4 m / 1e-4 rad
    dT = float((Tt.cpu() - T(g[name + ".T"])).norm())
    dR = rot_angle(R.cpu(), g[name + ".R"])
    assert dT < 1e-4 and dR < 1e-4, (dT, dR)
    assert conf.shape[0] == g[name + ".conf"].shape[0]
    np.testing.assert_allclose(conf.cpu().numpy(), g[name + ".conf"], rtol=3e-3, atol=0)
    assert abs(rmse - float(g[name + ".rmse"])) < 1e-3
    # batched call shape contract (decoder.py:122-126)
    Rb, Tb, cb, rb = dec.registration_forward(s.unsqueeze(0), d.unsqueeze(0), num_sample=0.5)
    assert tuple(Rb.shape) == (1, 3, 3) and tuple(Tb.shape) == (1, 3, 1) and cb.dim() == 2 and isinstance(rb, list)
    assert torch.equal(Rb[0], R)


def test_registration_argument_errors(dec):
    d = torch.zeros(131, 16)
    with pytest.raises(ValueError):
        dec.registration_forward(d, d, num_sample=0.0)
    with pytest.raises(ValueError):
        dec.registration_forward(d, d, num_sample="half")
    with pytest.raises(AssertionError):
        dec.forward(d, d)  # training-only in the reference (decoder.py:37)
    with pytest.raises(ValueError):  # masks are (B, tokens) bool (tests/test_round2_fixtures.py holds their arithmetic)
        dec.registration_forward(d, d, src_padding_mask=torch.zeros(1, 15, dtype=torch.bool))
    with pytest.raises(ValueError):
        dec.registration_forward(d, d, src_padding_mask=torch.zeros(1, 16, dtype=torch.uint8))


def test_loop_detection_vs_reference(dec):
    g = load_golden("decoder.npz")
    p = dec.loop_detection_forward(T(g["loop.src"]), T(g["loop.dst"]))
    assert tuple(p.shape) == (3,)
    np.testing.assert_allclose(p.cpu().numpy(), g["loop.prob"], atol=2e-5, rtol=0)


def test_information_matrix_vs_oracle():
    from deeppointmap_amd.registration import calculate_information_matrix_from_pcd
    for n in (4096, 65536):
        a, b = synthetic.frame(0, n) * 60, synthetic.frame(1, n) * 60
        SE3 = synthetic.relative_pose(0, 1).float()
        SE3[:3, 3] += torch.tensor([0.03, -0.02, 0.01])
        want = O.information_matrix(a, b, SE3)
        got = calculate_information_matrix_from_pcd(a, b, SE3, device=DEV)
        assert got.device.type == "cpu" and got.dtype == torch.float32 and tuple(got.shape) == (6, 6)
        assert float(want[3, 3]) > 0.3 * n  # most points do find a neighbour within 1 m
        assert float(got[3, 3]) == float(want[3, 3])  # the matched set has exactly the oracle's size
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-4, atol=2e-4 * float(want.abs().max()))
    # hand case: nothing within the radius -> zero matrix; identity on a shifted copy -> counts
    a = torch.tensor([[0.0, 10, 20], [0, 0, 0], [1, 2, 3]])
    got = calculate_information_matrix_from_pcd(a, a + 50.0, torch.eye(4), device=DEV)
    assert float(got.abs().sum()) == 0.0
    got = calculate_information_matrix_from_pcd(a, a, torch.eye(4), device=DEV)
    assert float(got[3, 3]) == 3.0 and float(got[0, 4]) == -6.0 and float(got[1, 5]) == -30.0


def test_information_matrix_vs_reference_code():
    """the HIP path against the reference's own function (tests/golden/infomat.npz; knn_points(K=1) answered by exhaustive
    search when the fixture was made): matched-set size exactly, entries within fp32 accumulation noise of the
    reference's fp32 sums"""
    from deeppointmap_amd.registration import calculate_information_matrix_from_pcd
    from test_oracle_golden import _infomat_cases
    g, cases = _infomat_cases()
    for name, (a, b) in cases.items():
        want = g[name + ".info"]
        got = calculate_information_matrix_from_pcd(a, b, T(g[name + ".SE3"]), device=DEV).numpy()
        assert got[3, 3] == want[3, 3], name
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4 * np.abs(want).max(), err_msg=name)


def test_batched_registration_equals_per_pair_calls(dec):
    g = load_golden("decoder.npz")
    names = ["synthetic01", "kitti01"]
    S = torch.stack([T(g[n + ".src_desc"]) for n in names] + [T(g["kitti01.dst_desc"])])
    D = torch.stack([T(g[n + ".dst_desc"]) for n in names] + [T(g["synthetic01.src_desc"])])
    table = torch.zeros(3, 56, device=DEV)
    res = dec.registration_forward_batch(S, D, 0.5, header_out=table[:, :20])
    assert tuple(res.shape) == (3, 20 + 2 * 128)
    for b in range(3):
        R, Tt, conf, rmse = dec.registration_forward(S[b], D[b], num_sample=0.5)
        n_in = int(res[b, 14])
        assert torch.equal(res[b, :9].view(3, 3), R) and torch.equal(res[b, 9:12].view(3, 1), Tt)
        assert n_in == conf.numel() and torch.equal(res[b, 20:20 + n_in], conf)
        assert float(res[b, 12]) == rmse
        assert torch.equal(table[b, :20], res[b, :20])


def test_stacked_sides_equal_the_per_side_loop(dec):
    """M != N (scan-to-map): the row-wise layers run once over the stacked source + target rows; bit-identical to the
    per-side loop, with and without padding masks, single pairs and batches"""
    gen = torch.Generator().manual_seed(12)
    for B, M, N, masked in [(1, 1024, 256, False), (1, 4096, 256, False), (2, 300, 77, True), (3, 256, 512, True)]:
        s = torch.cat([torch.rand(B, 128, M, generator=gen), 60 * torch.randn(B, 3, M, generator=gen)], 1).to(DEV)
        d = torch.cat([torch.rand(B, 128, N, generator=gen), 60 * torch.randn(B, 3, N, generator=gen)], 1).to(DEV)
        ms = md = None
        if masked:
            ms, md = torch.zeros(B, M, dtype=torch.bool), torch.zeros(B, N, dtype=torch.bool)
            ms[:, M - 20:], md[:, N - 9:] = True, True
        assert dec.stack_sides
        try:
            a = dec._descriptor_attention_forward(s, d, ms, md)
            dec.stack_sides = False
            b = dec._descriptor_attention_forward(s, d, ms, md)
        finally:
            dec.stack_sides = True
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]), (B, M, N, masked)


def test_pairs_entry_equals_batched_entry(dec):
    """registration_forward_pairs shares the per-frame decoder prefix between the pairs a frame takes part in; rows
    are computed by row-wise kernels, so the result must equal the gathered-batch entry point bit for bit."""
    g = load_golden("decoder.npz")
    frames = torch.stack([T(g["synthetic01.src_desc"]), T(g["synthetic01.dst_desc"]), T(g["kitti01.src_desc"]),
                          T(g["kitti01.dst_desc"])]).to(DEV)
    src = torch.tensor([0, 1, 2, 3, 0], dtype=torch.int32, device=DEV)
    dst = torch.tensor([1, 2, 3, 0, 2], dtype=torch.int32, device=DEV)
    t1, t2 = torch.zeros(5, 56, device=DEV), torch.zeros(5, 56, device=DEV)
    r1 = dec.registration_forward_pairs(frames, src, dst, 0.5, header_out=t1[:, :20])
    r2 = dec.registration_forward_batch(frames[src.long()], frames[dst.long()], 0.5, header_out=t2[:, :20])
    assert torch.equal(t1, t2) and torch.equal(r1[:, :20], r2[:, :20])
    for b in range(5):  # entries past the inlier count are unspecified
        n_in = int(r1[b, 14])
        assert n_in > 0 and torch.equal(r1[b, 20:20 + n_in], r2[b, 20:20 + n_in])


def test_pairs_entry_at_other_decoder_widths(cfg_full):
    """A decoder whose heads are not 32 wide (model_channel 128 / 512: heads of 16 / 64) has no indexed attention kernel:
    the pair-list entry must fall back to projecting per pair side (it raised `unsupported shape` before) and still equal the
    gathered-batch entry bit for bit; the 128-wide decoder is also held to the oracle."""
    import copy
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.weights import init_procedural
    gen = torch.Generator().manual_seed(23)
    for width in (128, 512):
        cfg = copy.deepcopy(cfg_full)
        cfg.decoder.model_channel = width
        d = init_procedural(Decoder(cfg)).to(DEV)
        assert d.dedup_frames
        frames = torch.rand(5, 131, 96, generator=gen)
        frames[:, 128:] = (torch.rand(5, 3, 96, generator=gen) * 2 - 1) * 30
        frames = frames.to(DEV)
        src = torch.tensor([0, 1, 2, 3, 0], dtype=torch.int32, device=DEV)
        dst = torch.tensor([1, 2, 3, 4, 2], dtype=torch.int32, device=DEV)
        r1 = d.registration_forward_pairs(frames, src, dst, 0.5)
        r2 = d.registration_forward_batch(frames[src.long()], frames[dst.long()], 0.5)
        assert torch.equal(r1[:, :16].contiguous().view(torch.int32), r2[:, :16].contiguous().view(torch.int32))
        if width == 128:
            tr = {}
            d.registration_forward(frames[0], frames[1], num_sample=0.5, trace=tr)
            sd = {k: v.detach().cpu() for k, v in d.flat().items()}
            xo, _, yo, _ = O.descriptor_attention(sd, cfg, frames[0].cpu()[None], frames[1].cpu()[None])
            torch.testing.assert_close(tr["x"].cpu().view(-1), xo.reshape(-1), rtol=1e-3, atol=2e-4)
            torch.testing.assert_close(tr["y"].cpu().view(-1), yo.reshape(-1), rtol=1e-3, atol=2e-4)


def test_pairs_entry_projects_each_frame_once(dec):
    """64 consecutive-frame pairs over 65 frames (the bench's shape: 32-query waves in the attention kernel): the first
    cross-attention block's q | k | v projection runs once per FRAME and the attention kernel picks each pair's sequences
    through the index list (dpm_attention_indexed); bit-equal to projecting every pair side (dedup_frames = False)."""
    gen = torch.Generator().manual_seed(11)
    frames = torch.randn(65, 131, 256, generator=gen)
    frames[:, 128:] = torch.rand(65, 3, 256, generator=gen) * 2 - 1
    frames = frames.to(DEV)
    src = torch.arange(0, 64, dtype=torch.int32, device=DEV)
    dst = src + 1
    assert dec.dedup_frames
    try:
        r1 = dec.registration_forward_pairs(frames, src, dst, 0.5)
        dec.dedup_frames = False
        r2 = dec.registration_forward_pairs(frames, src, dst, 0.5)
    finally:
        dec.dedup_frames = True
    assert torch.equal(r1[:, :16].contiguous().view(torch.int32), r2[:, :16].contiguous().view(torch.int32))  # NaN-safe
    for b in range(64):
        n_in = int(r1[b, 14])
        assert torch.equal(r1[b, 20:20 + n_in], r2[b, 20:20 + n_in])


def test_batched_information_matrix_equals_single(ops):
    """3 pairs (plain block mapping) and 8 pairs (XCD-aware mapping, one of them with the source pushed half
    out of the target's bounding box) against the single-pair entry point."""
    from deeppointmap_amd.registration import calculate_information_matrix_from_pcd
    pts = torch.stack([synthetic.frame(f, 8192) * 60 for f in range(4)]).to(DEV)
    for pairs in ([(0, 1), (1, 2), (2, 0)], [(0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (1, 3), (2, 2), (3, 1)]):
        E = len(pairs)
        table = torch.zeros(E, 56, device=DEV)
        poses = []
        for p, (a, b) in enumerate(pairs):
            SE3 = synthetic.relative_pose(a, b).float()
            if p == 5:
                SE3[:3, 3] += torch.tensor([40.0, -25.0, 0.5])
            table[p, :9] = SE3[:3, :3].reshape(9)
            table[p, 9:12] = SE3[:3, 3]
            poses.append(SE3)
        src = torch.tensor([a for a, _ in pairs], dtype=torch.int32, device=DEV)
        dst = torch.tensor([b for _, b in pairs], dtype=torch.int32, device=DEV)
        ops.information_matrix_batched(pts, src, dst, table[:, :12], table[:, 20:])
        # the two-call form (grids built before the poses exist) is the same computation bit for bit
        split = table.clone()
        split[:, 20:] = -1.0
        grids = ops.information_matrix_grids(pts, dst)
        ops.information_matrix_batched(pts, src, dst, split[:, :12], split[:, 20:], grids=grids)
        assert torch.equal(split, table)
        for p, (a, b) in enumerate(pairs):
            want = calculate_information_matrix_from_pcd(pts[a], pts[b], poses[p], device=DEV)
            got = table[p, 20:].view(6, 6).cpu()
            assert float(got[3, 3]) == float(want[3, 3]) and (p == 5 or float(got[3, 3]) > 1000)
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-3)
            ref = O.information_matrix(pts[a].cpu(), pts[b].cpu(), poses[p])
            assert float(got[3, 3]) == float(ref[3, 3])
            np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-4, atol=2e-4 * float(ref.abs().max()) + 1e-3)


def test_map_vs_map_registration_vs_oracle(dec, cfg_full, sd_dec):
    """Loop-closure shape (loop_closure.py:239-242): two map tiles of several keyframes each (M, N in the
    thousands).  The oracle is the checker; poses must agree within the north_star tolerance."""
    g = load_golden("decoder.npz")
    gen = torch.Generator().manual_seed(17)
    base_s, base_d = T(g["synthetic01.src_desc"]), T(g["synthetic01.dst_desc"])

    def tile(d, n):
        parts = []
        for i in range(n):
            jit = torch.cat([0.02 * torch.randn(128, 256, generator=gen).abs(), 0.3 * torch.randn(3, 256, generator=gen)])
            sh = torch.zeros(131, 1)
            sh[128, 0] = 4.0 * i
            parts.append(d + jit + sh)
        return torch.cat(parts, dim=1)

    src, dst = tile(base_s, 6), tile(base_d, 5)   # 1536 x 1280 descriptors
    R, Tt, conf, rmse = dec.registration_forward(src, dst, num_sample=0.5)
    Ro, To, co, ro = O.registration_forward(sd_dec, cfg_full, src, dst, 0.5)
    dT, dR = float((Tt.cpu() - To).norm()), rot_angle(R.cpu(), Ro)
    assert dT < 1e-4 and dR < 1e-4, (dT, dR)
    assert conf.shape[0] == co.shape[0] and abs(rmse - ro) < 1e-3


@pytest.mark.gpu
def test_registration_forward_replays_a_captured_graph_bit_for_bit(cfg_full):
    """A one-pair shape that keeps coming back is captured as a HIP graph (decoder.py: `graph_min_hits`) and replayed: R, T,
    the inlier confidences and rmse must be the eager launches' bits, for fresh inputs too, at the three shapes of a SLAM
    step's registrations; new weights invalidate the graph."""
    import threading
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.weights import init_procedural
    if threading.active_count() > 1:
        pytest.skip("graphs are only captured in single-threaded processes (decoder.py)")
    dev = torch.device("cuda:0")
    dec = init_procedural(Decoder(cfg_full)).to(dev)
    gen = torch.Generator().manual_seed(11)

    def pair(M, N):
        return (torch.cat([torch.rand(128, M, generator=gen), 60 * torch.randn(3, M, generator=gen)]).to(dev),
                torch.cat([torch.rand(128, N, generator=gen), 60 * torch.randn(3, N, generator=gen)]).to(dev))

    for M, N in ((256, 256), (1024, 256), (768, 512)):
        inputs = [pair(M, N) for _ in range(5)]
        dec.graph_min_hits = 0
        eager = [dec.registration_forward(s, d, num_sample=0.5) for s, d in inputs]
        dec.graph_min_hits = 2
        replay = [dec.registration_forward(s, d, num_sample=0.5) for s, d in inputs]
        assert any(v["graph"] is not None for k, v in dec._graphs.items() if k[:2] == (M, N))
        for e, r in zip(eager, replay):
            assert torch.equal(e[0], r[0]) and torch.equal(e[1], r[1]) and torch.equal(e[2], r[2]) and e[3] == r[3]
    # weights edited in place: the captured graph would still read the right memory, but the stamp says "changed" and the
    # shape is captured again
    s, d = pair(256, 256)
    before = dec.registration_forward(s, d, num_sample=0.5)
    with torch.no_grad():
        dec.p("similarity_head.2.weight").mul_(1.5)
    after = dec.registration_forward(s, d, num_sample=0.5)
    dec.graph_min_hits = 0
    want = dec.registration_forward(s, d, num_sample=0.5)
    assert torch.equal(after[0], want[0]) and torch.equal(after[1], want[1]) and not torch.equal(before[2], after[2])


def test_qkv_attention_planes_equal_the_fp32_hand_over(ops, monkeypatch):
    """ops.qkv_attention (dpm_linear_bf16x3_kvplanes -> dpm_attention_planes: K / V split into the attention kernel's operand planes
    by the projection's epilogue) against linear() + attention() on the same rows: bit-identical -- self attention, both directions
    of a cross attention in one launch (kv_shift), sequences drawn from stored frames (seq_index), one pair, a map-sized tile."""
    from deeppointmap_amd import knobs
    gen = torch.Generator().manual_seed(77)
    E, H = 256, 8
    W = (torch.randn(3 * E, E, generator=gen) / 16).to(DEV)
    b = (0.1 * torch.randn(3 * E, generator=gen)).to(DEV)
    monkeypatch.setattr(ops, "KV_PLANES_MIN_ROWS", 0)
    for U, B, M, shift, indexed in ((6, 6, 256, 0, False), (8, 8, 256, 4, False), (5, 8, 256, 4, True), (2, 2, 64, 1, False),
                                    (2, 2, 4096, 1, False), (3, 3, 192, 0, False)):
        x = torch.randn(U * M, E, generator=gen).to(DEV)
        seq = torch.randint(0, U, (B,), generator=gen).int().to(DEV) if indexed else None
        got = ops.qkv_attention(x, W, b, B, M, H, kv_shift=shift, seq_index=seq)
        assert got is not None
        qkv = ops.linear(x, W, b)
        want = ops.attention(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], B, M, M, H, kv_shift=shift, seq_index=seq)
        assert torch.equal(got, want), (U, B, M, shift, indexed, float((got - want).abs().max()))
    # shapes the planes do not cover fall back (None): ragged token counts, key ranges (1024 tokens: attention_key_splits)
    assert ops.qkv_attention(torch.randn(2 * 100, E, device=DEV), W, b, 2, 100, H) is None
    assert ops.qkv_attention(torch.randn(2 * 1024, E, device=DEV), W, b, 2, 1024, H) is None
    monkeypatch.setattr(knobs, "KV_PLANES", False)
    assert ops.qkv_attention(torch.randn(2 * 256, E, device=DEV), W, b, 2, 256, H) is None
