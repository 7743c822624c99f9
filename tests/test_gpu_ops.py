"""GPU: every C-ABI entry point of the encoder against the oracle / reference fixtures."""
import numpy as np
import pytest
import torch

from conftest import T, idx_rows_equal_as_sets, load_golden
from oracle import dpm_oracle as O
from deeppointmap_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from deeppointmap_amd import ops as _ops
    return _ops


def _lengths(vals):
    return torch.tensor(vals, dtype=torch.int32, device=DEV)


def knn_rows_ok(idx_gpu, points, centers, radius, tol=4e-6):
    """Per row: same index set as the exact fp64 answer, except for members whose distance is
    within `tol` of the radius^2 cut or of the K-th distance (documented borderline margin)."""
    idx_gpu = np.asarray(idx_gpu)
    P, C = np.asarray(points, np.float64), np.asarray(centers, np.float64)
    S, K = idx_gpu.shape
    bad = 0
    for s in range(S):
        d = ((P - C[s]) ** 2).sum(1)
        order = np.argsort(d, kind="stable")[:K]
        want = set(int(i) if d[i] <= radius ** 2 else int(order[0]) for i in order)
        got = set(int(i) for i in idx_gpu[s])
        if got == want:
            continue
        dk = d[order[-1]]
        for i in got ^ want:
            if not (abs(d[i] - radius ** 2) <= tol or abs(d[i] - dk) <= tol):
                bad += 1
                break
    return bad


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("algo", [1, 2])
def test_fps_bit_exact_vs_reference_fixtures(ops, algo):
    g = load_golden("fps.npz")
    names = sorted({k.rsplit(".", 1)[0] for k in g if k.endswith(".points")})
    for n in names:
        pts, length, K = T(g[n + ".points"]), int(g[n + ".length"]), g[n + ".new"].shape[0]
        idx, new, nl = ops.fps(pts.unsqueeze(0).to(DEV), _lengths([length]), K, algo=algo)
        assert np.array_equal(new[0].cpu().numpy(), g[n + ".new"]), (n, algo)
        assert np.array_equal((idx[0] < 0).cpu().numpy(), g[n + ".mask"]), (n, algo)
        assert int(nl[0]) == min(length, K)


@pytest.mark.parametrize("algo", [0, 1, 2, 5])
def test_fps_full_size_synthetic_bit_exact(ops, algo):
    g = load_golden("fps.npz")
    pts = synthetic.frame(0).t().contiguous()
    idx, new, _ = ops.fps(pts.unsqueeze(0).to(DEV), _lengths([65536]), 4096, algo=algo)
    assert np.array_equal(new[0].cpu().numpy(), g["synthetic0_k4096.new"])
    want = O.fps_indices_fast(pts, 65536, 4096)
    assert torch.equal(idx[0].cpu().long(), want)


def test_fps_batched_ragged_and_ties(ops):
    gen = torch.Generator().manual_seed(4)
    B, N, K = 5, 20000, 700
    pts = torch.rand(B, N, 3, generator=gen)
    pts[1] = torch.round(pts[1] * 8) / 8  # heavy ties: first-index rule must hold
    lens = [20000, 20000, 17001, 300, 1]
    for algo in (1, 2, 5):
        idx, new, nl = ops.fps(pts.to(DEV), _lengths(lens), K, algo=algo)
        for b in range(B):
            want = O.fps_indices_fast(pts[b], lens[b], K)
            assert torch.equal(idx[b].cpu().long(), want), (algo, b)
            assert torch.equal(new[b].cpu(), O.gather_masked(pts[b], want))
        assert nl.cpu().tolist() == [min(l, K) for l in lens]


def test_fps_bucket_kernels_on_adversarial_clouds(ops):
    """the bucket kernels prune by bounding boxes; clouds built to stress the pruning and the first-index rule:
    a lattice (every distance tied many times over), two far clusters with duplicates (equal maxima in different
    buckets), points on a line (candidates that change each other), and a K close to the cloud size."""
    gen = torch.Generator().manual_seed(9)
    N = 30000
    g = torch.stack(torch.meshgrid(torch.arange(31.0), torch.arange(31.0), torch.arange(32.0), indexing="ij"), -1).reshape(-1, 3)
    lattice = g[torch.randperm(g.shape[0], generator=gen)][:N] * 0.25
    a = torch.rand(N // 2, 3, generator=gen)
    clusters = torch.cat([a, a + torch.tensor([50.0, 0.0, 0.0])])[torch.randperm(N, generator=gen)]
    line = torch.zeros(N, 3)
    line[:, 0] = torch.rand(N, generator=gen) * 100
    uniform = torch.rand(N, 3, generator=gen) * torch.tensor([60.0, 60.0, 2.0])
    pts = torch.stack([lattice, clusters, line, uniform])
    for K, lens in ((2500, [N, N, N, N]), (20000, [N, 25000, N, 20001])):
        wants = [O.fps_indices_fast(pts[b], lens[b], K) for b in range(4)]
        for algo in (2, 5):  # buckets over grid cells; buckets over the STR packing
            idx, new, nl = ops.fps(pts.to(DEV), _lengths(lens), K, algo=algo)
            for b in range(4):
                assert torch.equal(idx[b].cpu().long(), wants[b]), (algo, K, b)
                assert torch.equal(new[b].cpu(), O.gather_masked(pts[b], wants[b]))


def test_fps_str_packing_ragged_batch_full_size(ops):
    """algo 5 (bucket kernel over the Sort-Tile-Recursive packing): full-size frames of different valid lengths
    in one launch, including lengths that leave slabs and leaves partly empty, and K beyond a short frame."""
    pts, _ = synthetic.frames(4, 65536)
    pts = pts.transpose(1, 2).contiguous()
    pts[3] = torch.round(pts[3] * 64) / 64  # quantised copy: many exactly equal distances
    lens = [65536, 40001, 16385, 65535]
    for algo in (5,):
        idx, new, nl = ops.fps(pts.to(DEV), _lengths(lens), 4096, algo=algo)
        for b in range(4):
            want = O.fps_indices_fast(pts[b], lens[b], 4096)
            assert torch.equal(idx[b].cpu().long(), want), (algo, b)
            assert torch.equal(new[b].cpu(), O.gather_masked(pts[b], want))
    ref2, _, _ = ops.fps(pts[:, :17000].contiguous().to(DEV), _lengths([17000, 100, 0, 1]), 300, algo=2)
    # an empty frame keeps slot 0 = index 0 (utils.py:249-250)
    idx5, _, nl5 = ops.fps(pts[:, :17000].contiguous().to(DEV), _lengths([17000, 100, 0, 1]), 300, algo=5)
    assert torch.equal(idx5, ref2) and nl5.cpu().tolist() == [300, 100, 1, 1]
    for b, l in enumerate([17000, 100, 0, 1]):
        if l > 0:
            assert torch.equal(idx5[b].cpu().long(), O.fps_indices_fast(pts[b, :17000], l, 300)), b


def test_fps_beyond_65536_points_per_frame(ops):
    """The bucket kernels end at 65 536 points per frame; larger frames (the reference takes any size) fall back to the
    plain kernel and stay bit-exact."""
    gen = torch.Generator().manual_seed(6)
    xyz = torch.randn(2, 70000, 3, generator=gen) * torch.tensor([30.0, 30.0, 2.0])
    lens = torch.tensor([70000, 66000], dtype=torch.int32)
    got = ops.fps(xyz.to(DEV), lens.to(DEV), 300)[0].cpu()
    for b in range(2):
        assert torch.equal(got[b].long(), O.fps_indices_fast(xyz[b], int(lens[b]), 300)), b


def test_fps_all_levels_sizes(ops):
    gen = torch.Generator().manual_seed(6)
    for N, K in [(4096, 1024), (1024, 256), (256, 64), (64, 16), (16, 16), (3, 8)]:
        pts = torch.rand(2, N, 3, generator=gen)
        idx, _, _ = ops.fps(pts.to(DEV), _lengths([N, max(N - 2, 1)]), K)
        for b, l in enumerate([N, max(N - 2, 1)]):
            assert torch.equal(idx[b].cpu().long(), O.fps_indices_fast(pts[b], l, K)), (N, K, b)


# ------------------------------------------------------------------------------------------------
def test_knn_vs_reference_fixtures(ops):
    g = load_golden("knn.npz")
    names = sorted({k.rsplit(".", 1)[0] for k in g if k.endswith(".idx")})
    for n in names:
        pts, ctr, length = T(g[n + ".points"]), T(g[n + ".centers"]), int(g[n + ".length"])
        r, K = float(g[n + ".radius"]), g[n + ".idx"].shape[1]
        idx = ops.knn_hybrid(pts.unsqueeze(0).to(DEV), _lengths([length]), ctr.unsqueeze(0).to(DEV), K, r)[0].cpu().numpy()
        # the kernel reproduces the reference's expanded-form arithmetic bit for bit, so the index
        # SETS are the reference's (only exact distance ties at the K-th slot may pick differently)
        same = idx_rows_equal_as_sets(idx, g[n + ".idx"])
        assert same.all(), (n, same.mean())  # every row: tied K-th places follow torch.topk (topk_emulate.h)
        assert knn_rows_ok(idx, pts[:length].numpy(), ctr.numpy(), r) == 0, n
        assert (idx[:, 0] == g[n + ".idx"][:, 0]).all()  # slot 0 = the reference's nearest


def test_knn_dense_cluster_overflows_candidate_list(ops):
    # > CAP points inside the radius forces the in-kernel compaction path
    gen = torch.Generator().manual_seed(8)
    pts = torch.cat([torch.randn(3000, 3, generator=gen) * 0.01, torch.rand(1000, 3, generator=gen)]).unsqueeze(0)
    ctr = pts[:, :37].contiguous()
    idx = ops.knn_hybrid(pts.to(DEV), _lengths([4000]), ctr.to(DEV), 32, 0.2)[0].cpu().numpy()
    assert knn_rows_ok(idx, pts[0].numpy(), ctr[0].numpy(), 0.2) == 0


def test_knn_boundary_ties_follow_reference_topk(ops):
    # lattice points: many bit-equal distances straddle the K-th slot.  The reference's choice there is
    # libstdc++ heap-select (torch.topk CPU, K*64 <= N); the kernel re-runs such rows through an exact
    # emulation, so the index sets must be identical, not merely equivalent.
    gen = torch.Generator().manual_seed(31)
    # 8192 / 20000 points: beyond the 4096-point head of the tie kernel, i.e. through its filter pass (and, on this
    # lattice where thousands of points undercut the early top, through the list-overflow fallback as well)
    for N, K, r in [(4096, 32, 0.3), (8192, 32, 0.25), (2048, 16, 0.4), (20000, 32, 0.2)]:
        pts = (torch.randint(0, 24, (1, N, 3), generator=gen).float() / 24.0)
        ctr = pts[:, torch.randperm(N, generator=gen)[:200]].contiguous()
        pad = torch.zeros(1, N, dtype=torch.bool)
        want, dk = O.hybrid_query(r, K, pts, ctr, pad, return_dist=True)
        got = ops.knn_hybrid(pts.to(DEV), _lengths([N]), ctr.to(DEV), K, r)[0].cpu().numpy()
        # make sure the case really contains boundary ties inside the radius
        full = O.expanded_sqdist(ctr, pts)[0]
        srt = torch.sort(full, dim=1)[0]
        n_tie_rows = int(((srt[:, K - 1] == srt[:, K]) & (srt[:, K] <= r * r)).sum())
        assert n_tie_rows > 20, n_tie_rows
        assert idx_rows_equal_as_sets(got, want[0].numpy()).all(), (N, K)


def test_knn_grid_equals_brute_force(ops):
    # the grid search must return exactly what the all-pairs scan returns (sets, slot 0, padded centres)
    gen = torch.Generator().manual_seed(77)
    pts = torch.cat([synthetic.frame(2, 16384).t(), torch.zeros(100, 3)]).unsqueeze(0).repeat(2, 1, 1).contiguous()
    pts[1, :16384] = pts[1, :16384] * torch.tensor([1.0, 0.3, 1.0])  # different extent per frame
    lens = _lengths([16384, 12000])
    ctr = torch.cat([pts[:, torch.randperm(12000, generator=gen)[:500]], torch.zeros(2, 12, 3) + 7.0], dim=1).contiguous()
    for r, K in [(0.05, 32), (0.1, 32), (0.4, 16)]:
        a = ops.knn_hybrid(pts.to(DEV), lens, ctr.to(DEV), K, r).cpu().numpy()
        b = ops.knn_hybrid(pts.to(DEV), lens, ctr.to(DEV), K, r, brute=True).cpu().numpy()
        for f in range(2):
            assert idx_rows_equal_as_sets(a[f], b[f]).all(), (r, K, f)
            assert (a[f][:, 0] == b[f][:, 0]).all()
        assert (a[:, -12:] == a[:, -12:, :1]).all()  # far-away centres: every slot is the nearest point


def test_knn_grid_far_from_the_origin_equals_brute_force(ops):
    """The reference's expanded-form distance loses absolute precision with the squared magnitude of the coordinates: 65
    units from the origin its quantum is 5e-4, a fifth of r^2 = 0.0025, and points truly outside the radius are computed
    inside it.  The grid's cell edge has to grow with that error (csrc/knn.hip, knn_grid_build_kernel) or the 3x3
    neighbourhood misses them: found by scripts/fuzz_encoder.py on synthetic frames that drift away from the origin."""
    for shift in ([-65.4, -29.1, 0.0], [5.0, -4.0, 0.3]):
        pts = (synthetic.frame(4, 8192).t() + torch.tensor(shift)).unsqueeze(0).contiguous()
        lens = _lengths([8192])
        ctr = pts[:, ::8].contiguous()
        for r, K in [(0.05, 32), (0.1, 32)]:
            a = ops.knn_hybrid(pts.to(DEV), lens, ctr.to(DEV), K, r).cpu().numpy()
            b = ops.knn_hybrid(pts.to(DEV), lens, ctr.to(DEV), K, r, brute=True).cpu().numpy()
            assert idx_rows_equal_as_sets(a[0], b[0]).all(), (shift, r, K)
            assert (a[0][:, 0] == b[0][:, 0]).all()


def test_knn_prebuilt_grid_equals_one_call(ops):
    """dpm_knn_build_grid + dpm_knn_hybrid_prebuilt (grid sorted ahead of time, e.g. on another stream) against the
    one-call form: identical neighbour sets and nearest slots (the order of the other slots follows the grid's
    atomically scattered point order and is free in both forms), ragged frames and boundary-tie rows included."""
    gen = torch.Generator().manual_seed(41)
    pts = torch.cat([synthetic.frame(3, 8192).t().unsqueeze(0),
                     (torch.randint(0, 24, (1, 8192, 3), generator=gen).float() / 24.0)]).contiguous().to(DEV)
    lens = _lengths([8192, 5000])
    ctr = pts[:, :700].contiguous()
    for r, K in [(0.05, 32), (0.25, 32), (0.4, 16)]:
        one = ops.knn_hybrid(pts, lens, ctr, K, r)
        grid = ops.knn_grid(pts, lens, r)
        two = ops.knn_hybrid(pts, lens, ctr, K, r, grid=grid)
        for f in range(2):
            assert idx_rows_equal_as_sets(one[f].cpu().numpy(), two[f].cpu().numpy()).all(), (r, K, f)
        assert torch.equal(one[..., 0], two[..., 0])
    with pytest.raises(ValueError):
        ops.knn_hybrid(pts[:, :4096].contiguous(), lens, ctr, 32, 0.1, grid=grid)   # built for another N


def test_knn_reuse_of_self_query_is_identical(ops):
    # SetAbstraction centres = FPS picks of the points: rows copied from the self-query must equal a fresh search,
    # including frames with padded centres (fewer valid points than picks), on both the grid and the brute path
    gen = torch.Generator().manual_seed(90)
    for N, S, K, r in [(4096, 1024, 32, 0.1), (600, 256, 32, 0.3), (64, 16, 16, 1.6)]:
        pts = torch.rand(3, N, 3, generator=gen)
        lens = _lengths([N, max(S // 2, 4), N - 7])  # frame 1 has fewer valid points than S -> padded centres
        fidx, ctr, clen = ops.fps(pts.to(DEV), lens, S)
        full = ops.knn_hybrid(pts.to(DEV), lens, pts.to(DEV), K, r)
        fresh = ops.knn_hybrid(pts.to(DEV), lens, ctr, K, r)
        reused = ops.knn_hybrid(pts.to(DEV), lens, ctr, K, r, reuse_idx=full, center_src=fidx)
        # same SETS and same slot 0 (the order of the other slots is unspecified: the grid is built with atomics)
        for b in range(3):
            assert idx_rows_equal_as_sets(fresh[b].cpu().numpy(), reused[b].cpu().numpy()).all(), (N, S, b)
        assert torch.equal(fresh[..., 0], reused[..., 0])
        assert int((fidx[1] < 0).sum()) > 0


def test_knn_sparse_rows_pad_with_nearest(ops):
    pts = torch.tensor([[[0.0, 0, 0], [0.01, 0, 0], [5, 5, 5], [9, 9, 9]]])
    ctr = torch.tensor([[[0.0, 0, 0], [5.2, 5, 5], [100, 100, 100]]])
    idx = ops.knn_hybrid(pts.to(DEV), _lengths([4]), ctr.to(DEV), 4, 0.5)[0].cpu().tolist()
    assert sorted(idx[0]) == [0, 0, 0, 1] and idx[0][0] == 0
    assert idx[1] == [2, 2, 2, 2]
    assert idx[2] == [3, 3, 3, 3]  # nothing within the radius: every slot is the nearest point


# ------------------------------------------------------------------------------------------------
def test_linear_layernorm_interp_vs_torch(ops):
    gen = torch.Generator().manual_seed(2)
    for R, Cin, Cout in [(1000, 3, 16), (257, 515, 512), (64, 768, 256), (130, 128, 3), (5, 512, 1)]:
        x, W, b = torch.randn(R, Cin, generator=gen), torch.randn(Cout, Cin, generator=gen) / Cin ** 0.5, torch.randn(Cout, generator=gen)
        res = torch.randn(R, Cout, generator=gen)
        y = ops.linear(x.to(DEV), W.to(DEV), b.to(DEV), act=ops.ACT_RELU, residual=res.to(DEV)).cpu()
        want = torch.relu(x.double() @ W.double().t() + b.double() + res.double()).float()
        torch.testing.assert_close(y, want, rtol=1e-5, atol=2e-5)
    y = ops.linear(x.to(DEV), W.to(DEV), None, act=ops.ACT_SIGMOID).cpu()
    torch.testing.assert_close(y, torch.sigmoid(x @ W.t()), rtol=1e-5, atol=1e-6)
    # strided output (column slice of a wider buffer)
    buf = torch.zeros(7, 40, device=DEV)
    x, W = torch.randn(7, 9, generator=gen), torch.randn(12, 9, generator=gen)
    ops.linear(x.to(DEV), W.to(DEV), None, out=buf[:, 20:32])
    torch.testing.assert_close(buf[:, 20:32].cpu(), x @ W.t(), rtol=1e-5, atol=1e-5)
    assert float(buf[:, :20].abs().sum()) == 0 and float(buf[:, 32:].abs().sum()) == 0
    for R, C in [(300, 32), (301, 64), (1001, 128), (256, 256), (77, 512), (33, 1024), (17, 2048), (19, 96), (5, 131)]:
        x, pre, post = (torch.randn(R, C, generator=gen) * 3 for _ in range(3))
        gm, bt = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
        y = ops.layernorm(x.to(DEV), gm.to(DEV), bt.to(DEV), act=ops.ACT_RELU, pre=pre.to(DEV), post=post.to(DEV)).cpu()
        want = torch.relu(torch.nn.functional.layer_norm(x + pre, (C,), gm, bt, 1e-5) + post)
        torch.testing.assert_close(y, want, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("ln_b3", [False, True])
def test_fused_linear_layernorm_kernel_vs_torch(ops, monkeypatch, ln_b3):
    """dpm_linear_layernorm (GEMM with the LayerNorm in its epilogue) at every fused width, ragged row counts, with and
    without the pre / post residuals -- the row threshold of ops.linear_layernorm lifted so that small inputs reach it --
    and the two-kernel form the wrapper takes below the threshold, both against an fp64 reference."""
    from deeppointmap_amd import knobs
    monkeypatch.setattr(knobs, "GEMM_LN_BF16X3", ln_b3)   # True: the bf16x3 forms for 128 <= K <= 512 (opt-in, knobs.py)
    gen = torch.Generator(device=DEV).manual_seed(9)
    both = {}
    for min_rows in (0, 1 << 30):
        monkeypatch.setattr(ops, "FUSED_LN_MIN_ROWS", min_rows)
        gen.manual_seed(9)
        for R, Cin, Cout, relu, use_pre, use_post in [(1000, 64, 32, True, False, False), (130, 128, 64, False, True, False),
                                                     (257, 256, 128, True, True, True), (4096, 1024, 256, True, False, True),
                                                     (65, 32, 256, False, False, False), (20000, 256, 256, True, True, False)]:
            x = torch.randn(R, Cin, device=DEV, generator=gen)
            W = torch.randn(Cout, Cin, device=DEV, generator=gen) / Cin ** 0.5
            b, gm, bt = (torch.randn(Cout, device=DEV, generator=gen) for _ in range(3))
            pre = torch.randn(R, Cout, device=DEV, generator=gen) if use_pre else None
            post = torch.randn(R, Cout, device=DEV, generator=gen) if use_post else None
            y = ops.linear_layernorm(x, W, b, gm, bt, act=ops.ACT_RELU if relu else ops.ACT_NONE, pre=pre, post=post)
            z = x.double() @ W.double().t() + b.double()
            z = z + pre.double() if use_pre else z
            z = torch.nn.functional.layer_norm(z, (Cout,), gm.double(), bt.double(), 1e-5)
            z = z + post.double() if use_post else z
            want = (torch.relu(z) if relu else z).float()
            torch.testing.assert_close(y, want, rtol=2e-5, atol=5e-5)
            both.setdefault((R, Cin, Cout), []).append(y)
    # the wrapper picks between the two forms by row count: a frame's result must not depend on that
    for key, (fused, split) in both.items():
        assert torch.equal(fused, split), key


def test_bf16x3_linear_vs_fp64_and_layout_independence(ops, monkeypatch):
    """The bf16x3 GEMM (csrc/gemm_b3.hip: every fp32 operand split exactly into three bf16 terms, six term products accumulated
    in fp32) is what ops.linear runs for layers with K <= 512 when knobs.GEMM_BF16X3 is set: (i) its error against fp64 is at the level of the exact-fp32 MFMA
    kernel's (the tolerance every other GEMM test uses), also on values spanning many binades; (ii) a row's bits do not depend
    on the tile variant (row count), the row stride of the input (131- against 132-float token rows), a weight being a row block
    of a larger parameter, or the batch the row travels in."""
    from deeppointmap_amd import knobs
    monkeypatch.setattr(knobs, "GEMM_BF16X3", True)
    gen = torch.Generator(device=DEV).manual_seed(31)
    for R, Cin, Cout, relu in [(32768, 256, 768, False), (1000, 96, 132, True), (257, 512, 256, True), (70000, 32, 32, False),
                               (64, 128, 512, False), (3, 64, 4, True)]:
        x = torch.randn(R, Cin, device=DEV, generator=gen) * torch.exp2(torch.randint(-12, 12, (R, 1), device=DEV, generator=gen).float())
        W = torch.randn(Cout, Cin, device=DEV, generator=gen) / Cin ** 0.5
        b, res = torch.randn(Cout, device=DEV, generator=gen), torch.randn(R, Cout, device=DEV, generator=gen)
        y3 = ops.linear(x, W, b, act=ops.ACT_RELU if relu else ops.ACT_NONE, residual=res)
        y32 = ops.linear(x, W, b, act=ops.ACT_RELU if relu else ops.ACT_NONE, residual=res, exact=True)
        assert not torch.equal(y3, y32) or R < 8          # (two different kernels ran)
        n = min(R, 2048)
        want = x[:n].double() @ W.double().t() + b.double() + res[:n].double()
        want = torch.relu(want) if relu else want
        scale = (x[:n].double().abs() @ W.double().abs().t()) + 1.0     # size of the terms that were summed
        e3 = ((y3[:n].double() - want).abs() / scale).max()
        e32 = ((y32[:n].double() - want).abs() / scale).max()
        assert float(e3) < 4e-7 and float(e3) < 2.0 * float(e32) + 1e-8, (R, Cin, Cout, float(e3), float(e32))
    x = torch.randn(4096, 256, device=DEV, generator=gen)
    W = torch.randn(768, 256, device=DEV, generator=gen) / 16
    b = torch.randn(768, device=DEV, generator=gen)
    big = ops.linear(x, W, b)
    assert torch.equal(ops.linear(x[:256].contiguous(), W, b), big[:256])          # 32 x 32 tiles against 64 x 64
    assert torch.equal(ops.linear(x[1000:2280].contiguous(), W, b), big[1000:2280])
    assert torch.equal(ops.linear(x, W[256:], b[256:]), big[:, 256:])               # row block of the parameter: the same planes
    wide = torch.zeros(4096, 259, device=DEV)                                       # rows of 259 floats: scalar loads
    wide[:, :256] = x
    assert torch.equal(ops.linear(wide[:, :256], W, b), big)
    wide2 = torch.zeros(4096, 260, device=DEV)
    wide2[:, 4:] = x
    assert torch.equal(ops.linear(wide2[:, 4:], W, b), big)
    # 64 x 128 tiles (launches of >= 1024 of them; 768 and 384 columns) with a ragged last row tile, against the 64 x 64 /
    # 32 x 32 variants and the four-tiles-in-flight variants (small launches) on slices of the same rows
    xb = torch.randn(32768 + 17, 256, device=DEV, generator=gen)
    for Wt, bt in ((W, b), (W[:384], b[:384])):
        bigw = ops.linear(xb, Wt, bt, act=ops.ACT_RELU)
        for lo, hi in [(0, 200), (30000, 31000), (32768 - 40, 32768 + 17), (5000, 5000 + 4096)]:
            assert torch.equal(ops.linear(xb[lo:hi].contiguous(), Wt, bt, act=ops.ACT_RELU), bigw[lo:hi]), (Wt.shape[0], lo, hi)
        want = torch.relu(xb[-64:].double() @ Wt.double().t() + bt.double())
        torch.testing.assert_close(bigw[-64:].double(), want, rtol=1e-5, atol=2e-5)
    res = torch.randn(32768 + 17, 768, device=DEV, generator=gen)                    # the wide tile's residual path
    assert torch.equal(ops.linear(xb[100:612].contiguous(), W, b, residual=res[100:612].contiguous()),
                       ops.linear(xb, W, b, residual=res)[100:612])
    # weights edited in place: the planes follow
    with torch.no_grad():
        W.mul_(0.5)
    torch.testing.assert_close(ops.linear(x, W, b), (big - b) * 0.5 + b, rtol=1e-5, atol=1e-5)


def test_bf16x3_non_finite_inputs_give_nan_not_garbage(ops, monkeypatch):
    """The exact three-way split is defined for finite values (csrc/dpm_common.h, split3): an infinite operand becomes NaN in the
    rows it touches -- where the fp32 kernel would carry the infinity -- and every other row is untouched.  Pinned so that the
    difference is a documented one."""
    from deeppointmap_amd import knobs
    monkeypatch.setattr(knobs, "GEMM_BF16X3", True)
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(4096, 256, device=DEV, generator=gen)
    W, b = torch.randn(256, 256, device=DEV, generator=gen) / 16, torch.randn(256, device=DEV, generator=gen)
    clean = ops.linear(x, W, b)
    x2 = x.clone()
    x2[7, 3], x2[100, 200], x2[4000, 0] = float("inf"), float("-inf"), float("nan")
    y = ops.linear(x2, W, b)
    bad = torch.zeros(4096, dtype=torch.bool, device=DEV)
    bad[[7, 100, 4000]] = True
    assert torch.isnan(y[bad]).all() and torch.equal(y[~bad], clean[~bad])
    y32 = ops.linear(x2, W, b, exact=True)                      # the fp32 kernel: infinities stay infinities, NaN stays NaN
    assert torch.isinf(y32[7]).all() and torch.isinf(y32[100]).all() and torch.isnan(y32[4000]).all()


def test_linear_large_tiles_vs_torch(ops):
    """The shapes that take the 128x128 persistent kernel (K >= 1024, >= 512 output tiles) and the 64x64 kernel's
    staged epilogue with ragged edges, against an fp64 product."""
    gen = torch.Generator(device=DEV).manual_seed(5)
    for R, Cin, Cout, relu in [(8192 + 40, 1024, 1024, True), (4096, 1028, 2048, False), (1000, 96, 68, True), (333, 64, 260, False)]:
        x = torch.randn(R, Cin, device=DEV, generator=gen)
        W = torch.randn(Cout, Cin, device=DEV, generator=gen) / Cin ** 0.5
        b, res = torch.randn(Cout, device=DEV, generator=gen), torch.randn(R, Cout, device=DEV, generator=gen)
        y = ops.linear(x, W, b, act=ops.ACT_RELU if relu else ops.ACT_NONE, residual=res)
        want = x.double() @ W.double().t() + b.double() + res.double()
        want = (torch.relu(want) if relu else want).float()
        torch.testing.assert_close(y, want, rtol=1e-5, atol=3e-5)
    # batched entry through the big kernel: per-batch strides
    xb = torch.randn(2, 4096, 1024, device=DEV, generator=gen)
    Wb = torch.randn(2, 1024, 1024, device=DEV, generator=gen) / 32
    yb = ops.similarity_batched(xb, Wb)
    torch.testing.assert_close(yb, torch.bmm(xb.double(), Wb.double().transpose(1, 2)).float(), rtol=1e-5, atol=3e-5)


def test_three_interp_vs_oracle(ops):
    gen = torch.Generator().manual_seed(12)
    B, N, S, D1, D2 = 2, 64, 16, 24, 40
    xyz1 = torch.rand(B, N, 3, generator=gen)
    xyz2 = xyz1[:, :S].contiguous()  # coarse points coincide with fine ones (d = 0 -> clamp 1e-8)
    f1, f2 = torch.randn(B, N, D1, generator=gen), torch.randn(B, S, D2, generator=gen)
    out = ops.three_interp_cat(xyz1.to(DEV), xyz2.to(DEV), _lengths([S, S]), f1.to(DEV), f2.to(DEV)).cpu()
    d, i = torch.topk(O.expanded_sqdist(xyz1, xyz2), 3, dim=-1, largest=False)
    w = 1.0 / d.clamp(min=1e-8)
    w = w / w.sum(2, keepdim=True)
    bi = torch.arange(B).view(B, 1, 1)
    want = torch.cat([f1, (f2[bi, i] * w.unsqueeze(-1)).sum(2)], dim=-1)
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-4)


def test_three_interp_ties_at_the_third_neighbour_follow_torch_topk(ops):
    """Mirror-symmetric key points put two coarse points at EXACTLY the same distance from a fine one; when that tie
    straddles the third place, the reference keeps whichever torch.topk's std::nth_element (3 * 64 > S) or heap-select
    (S >= 192) leaves in front -- not the smallest index.  One-hot coarse features make the output row the weight vector."""
    gen = torch.Generator().manual_seed(4)
    for S in (16, 64, 256):
        half = (torch.randint(-8, 9, (1, S // 2, 3), generator=gen).float() * 0.125)
        half[..., 1] = half[..., 1].abs() + 0.125
        mirror = half * torch.tensor([1.0, -1.0, 1.0])
        xyz2 = torch.cat([half, mirror], 1)[:, torch.randperm(S, generator=gen)].contiguous()   # every point has its mirror image
        xyz1 = torch.randint(-8, 9, (1, 96, 3), generator=gen).float() * 0.125
        xyz1[..., 1] = 0.0                                                                  # fine points on the mirror plane
        f1, f2 = torch.zeros(1, 96, 4), torch.eye(S).unsqueeze(0)
        got = ops.three_interp_cat(xyz1.to(DEV), xyz2.to(DEV), _lengths([S]), f1.to(DEV), f2.to(DEV)).cpu()[..., 4:]
        d, i = torch.topk(O.expanded_sqdist(xyz1, xyz2), 3, dim=-1, largest=False)
        w = 1.0 / d.clamp(min=1e-8)
        want = torch.zeros(1, 96, S).scatter_add_(2, i, w / w.sum(-1, keepdim=True))
        all_d = O.expanded_sqdist(xyz1, xyz2)
        straddle = int(((all_d == d[..., 2:3]).sum(-1) > (d == d[..., 2:3]).sum(-1)).sum())
        assert straddle > 10, (S, straddle)          # the case this test is for does occur
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)


def test_group_mlp_max_vs_oracle(ops):
    gen = torch.Generator().manual_seed(21)
    for (N, S, K, Cin, Cout) in [(500, 40, 32, 16, 32), (128, 16, 16, 256, 512), (300, 9, 32, 128, 256), (700, 33, 32, 32, 64),
                                 (256, 7, 32, 64, 128), (64, 5, 16, 512, 512), (100, 6, 8, 10, 24)]:
        B = 2
        xyz, fea = torch.rand(B, N, 3, generator=gen), torch.randn(B, N, Cin, generator=gen)
        ctr = xyz[:, :S].contiguous()
        idx = torch.randint(0, N, (B, S, K), generator=gen)
        sd = {"m.0.weight": torch.randn(Cout, Cin + 3, 1, 1, generator=gen) / (Cin + 3) ** 0.5,
              "m.0.bias": torch.randn(Cout, generator=gen) * 0.1,
              "m.1.ln.weight": 1 + 0.1 * torch.randn(Cout, generator=gen), "m.1.ln.bias": 0.1 * torch.randn(Cout, generator=gen)}
        want = O.grouped_mlp_max(sd, "m", 0.3, xyz, fea, ctr, idx)
        # the three implementations: project-before-gather (default), one-kernel gather-GEMM, plain VALU
        for kw in ({}, {"fused": True}, {"generic": True}):
            got = ops.group_mlp_max(xyz.to(DEV), fea.to(DEV), ctr.to(DEV), idx.int().to(DEV), sd["m.0.weight"].to(DEV),
                                    sd["m.0.bias"].to(DEV), sd["m.1.ln.weight"].to(DEV), sd["m.1.ln.bias"].to(DEV), 0.3,
                                    **kw).cpu()
            torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


def test_group_mlp_from_xyz_equals_materialised_features(ops):
    gen = torch.Generator().manual_seed(23)
    B, N, S, K, Cin, Cout = 2, 3000, 100, 32, 16, 32
    xyz = torch.rand(B, N, 3, generator=gen)
    W0, b0 = torch.randn(Cin, 3, 1, generator=gen), torch.randn(Cin, generator=gen) * 0.1
    ctr = xyz[:, :S].contiguous()
    idx = torch.randint(0, N, (B, S, K), generator=gen).int()
    W = torch.randn(Cout, Cin + 3, 1, 1, generator=gen) / (Cin + 3) ** 0.5
    bias, gm, bt = torch.randn(Cout, generator=gen) * 0.1, 1 + 0.1 * torch.randn(Cout, generator=gen), 0.1 * torch.randn(Cout, generator=gen)
    d = lambda t: t.to(DEV)
    fea = ops.linear(d(xyz), d(W0), d(b0))
    want = ops.group_mlp_max(d(xyz), fea, d(ctr), d(idx), d(W), d(bias), d(gm), d(bt), 0.3)
    for kw in ({}, {"fused": True}):  # affine-on-the-fly (default) and the per-neighbour feature evaluation
        got = ops.group_mlp_max_from_xyz(d(xyz), d(W0), d(b0), d(ctr), d(idx), d(W), d(bias), d(gm), d(bt), 0.3, **kw)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_folded_grouping_layers_against_the_unfolded_form(ops, monkeypatch):
    """knobs.FOLD_GATHER (csrc/group_mlp.hip, FOLD): W_r (p - c) / r as W_r' p - W_r' c -- the point half in the projection
    GEMM's epilogue (dpm_linear_bf16x3_rank3), the centre half one vector per centre -- against the form that builds the relative
    coordinates per gathered row, at every width the encoder uses, first-level radius included (|p| / r = 20: the worst
    cancellation), and against the plain-VALU kernel.  The two forms differ by rounding only."""
    from deeppointmap_amd import knobs
    gen = torch.Generator().manual_seed(41)
    d = lambda t: t.to(DEV)
    shipped_min_radius = knobs.FOLD_MIN_RADIUS
    monkeypatch.setattr(knobs, "FOLD_MIN_RADIUS", 0.0)     # the fold at EVERY radius (shipped: only radii >= 0.2, knobs.py)
    for Cin, Cout, K, radius in ((32, 32, 32, 0.05), (32, 64, 32, 0.1), (64, 128, 32, 0.2), (128, 256, 32, 0.4), (256, 512, 16, 0.8)):
        B, N, S = 2, 3000, 400
        xyz = d(torch.rand(B, N, 3, generator=gen) * 2 - 1)                      # coordinates in the unit box, as the encoder's
        fea = d(torch.randn(B, N, Cin, generator=gen))
        ctr = xyz[:, :S].contiguous()
        idx = d(torch.randint(0, N, (B, S, K), generator=gen).int())
        W = d(torch.randn(Cout, Cin + 3, 1, 1, generator=gen) / (Cin + 3) ** 0.5)
        bias, gm, bt = d(0.1 * torch.randn(Cout, generator=gen)), d(1 + 0.1 * torch.randn(Cout, generator=gen)), d(0.1 * torch.randn(Cout, generator=gen))
        monkeypatch.setattr(knobs, "FOLD_GATHER", True)
        folded = ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, radius)
        monkeypatch.setattr(knobs, "FOLD_GATHER", False)
        plain = ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, radius)
        generic = ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, radius, generic=True)
        assert not torch.equal(folded, plain)                                      # (two different kernels ran)
        torch.testing.assert_close(folded, plain, rtol=0, atol=2e-5)
        torch.testing.assert_close(folded, generic, rtol=1e-4, atol=1e-4)
        # ... and the shipped threshold keeps the small radii on the unfolded form, bit for bit
        monkeypatch.setattr(knobs, "FOLD_GATHER", True)
        monkeypatch.setattr(knobs, "FOLD_MIN_RADIUS", shipped_min_radius)
        shipped = ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, radius)
        assert torch.equal(shipped, folded if radius >= shipped_min_radius else plain)
        monkeypatch.setattr(knobs, "FOLD_MIN_RADIUS", 0.0)
    # the affine first level folds inside its kernel: against the per-neighbour feature evaluation
    B, N, S, K = 2, 5000, 600, 32
    xyz = d(torch.rand(B, N, 3, generator=gen) * 2 - 1)
    ctr, idx = xyz[:, :S].contiguous(), d(torch.randint(0, N, (B, S, K), generator=gen).int())
    W0, b0 = d(torch.randn(16, 3, 1, generator=gen)), d(0.1 * torch.randn(16, generator=gen))
    W = d(torch.randn(32, 19, 1, 1, generator=gen) / 19 ** 0.5)
    bias, gm, bt = d(0.1 * torch.randn(32, generator=gen)), d(1 + 0.1 * torch.randn(32, generator=gen)), d(0.1 * torch.randn(32, generator=gen))
    a = ops.group_mlp_max_from_xyz(xyz, W0, b0, ctr, idx, W, bias, gm, bt, 0.05)
    b = ops.group_mlp_max_from_xyz(xyz, W0, b0, ctr, idx, W, bias, gm, bt, 0.05, fused=True)
    torch.testing.assert_close(a, b, rtol=0, atol=5e-5)


def test_centred_grouping_layers_against_the_folded_form(ops, monkeypatch):
    """knobs.CENTRED_GATHER (csrc/group_mlp.hip, CENTRED): LayerNorm's mean removal applied to the layer's weights once
    ((I - 11^T / C) W, fp64) instead of to every gathered row -- against the folded form that computes the row mean, at every width
    the encoder uses and for the affine first level; weights and bias with a LARGE common offset over the channels (the part the
    centring removes) included, LayerNorm scales of BOTH signs (the centred form folds sign(gamma) into the layer and applies |gamma|,
    beta and the ReLU after the maximum; an exact zero among them).  The forms differ by rounding only."""
    from deeppointmap_amd import knobs
    gen = torch.Generator().manual_seed(43)
    d = lambda t: t.to(DEV)
    monkeypatch.setattr(knobs, "FOLD_MIN_RADIUS", 0.0)     # the centred form exists on folded layers: fold at every radius here
    for Cin, Cout, K, radius, offset in ((32, 32, 32, 0.05, 0.0), (32, 64, 32, 0.1, 3.0), (64, 128, 32, 0.2, 0.0), (128, 256, 32, 0.4, 1.0),
                                         (256, 512, 16, 0.8, 0.0)):
        B, N, S = 2, 3000, 400
        xyz = d(torch.rand(B, N, 3, generator=gen) * 2 - 1)
        fea = d(torch.randn(B, N, Cin, generator=gen))
        ctr = xyz[:, :S].contiguous()
        idx = d(torch.randint(0, N, (B, S, K), generator=gen).int())
        W = d(torch.randn(Cout, Cin + 3, 1, 1, generator=gen) / (Cin + 3) ** 0.5 + offset * torch.randn(1, Cin + 3, 1, 1, generator=gen) / (Cin + 3) ** 0.5)
        bias, gm, bt = d(0.1 * torch.randn(Cout, generator=gen) + offset), d(torch.randn(Cout, generator=gen)), d(0.3 * torch.randn(Cout, generator=gen))
        gm[3] = 0.0
        monkeypatch.setattr(knobs, "CENTRED_GATHER", True)
        centred = ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, radius)
        monkeypatch.setattr(knobs, "CENTRED_GATHER", False)
        folded = ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, radius)
        generic = ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, radius, generic=True)
        assert not torch.equal(centred, folded)                                    # (two different kernels ran)
        torch.testing.assert_close(centred, folded, rtol=0, atol=2e-5 * (1 + offset))
        torch.testing.assert_close(centred, generic, rtol=1e-4, atol=1e-4 * (1 + offset))
    B, N, S, K = 2, 5000, 600, 32
    xyz = d(torch.rand(B, N, 3, generator=gen) * 2 - 1)
    ctr, idx = xyz[:, :S].contiguous(), d(torch.randint(0, N, (B, S, K), generator=gen).int())
    W0, b0 = d(torch.randn(16, 3, 1, generator=gen)), d(0.1 * torch.randn(16, generator=gen))
    W = d(torch.randn(32, 19, 1, 1, generator=gen) / 19 ** 0.5)
    bias, gm, bt = d(0.1 * torch.randn(32, generator=gen) + 0.5), d(torch.randn(32, generator=gen)), d(0.3 * torch.randn(32, generator=gen))
    monkeypatch.setattr(knobs, "CENTRED_GATHER", True)
    a = ops.group_mlp_max_from_xyz(xyz, W0, b0, ctr, idx, W, bias, gm, bt, 0.05)
    monkeypatch.setattr(knobs, "CENTRED_GATHER", False)
    b = ops.group_mlp_max_from_xyz(xyz, W0, b0, ctr, idx, W, bias, gm, bt, 0.05)
    c = ops.group_mlp_max_from_xyz(xyz, W0, b0, ctr, idx, W, bias, gm, bt, 0.05, fused=True)
    assert not torch.equal(a, b)
    torch.testing.assert_close(a, b, rtol=0, atol=5e-5)
    torch.testing.assert_close(a, c, rtol=0, atol=5e-5)


def test_weight_derived_tensors_follow_the_weights(ops):
    """ops caches what depends on weights alone (packed feature columns, the stage-0 affine map).  In-place updates,
    and a new weight tensor that lands on a freed one's address, must both be seen."""
    gen = torch.Generator().manual_seed(29)
    B, N, S, K, Cin, Cout = 1, 600, 50, 16, 16, 32
    d = lambda t: t.to(DEV)
    xyz, fea = d(torch.rand(B, N, 3, generator=gen)), d(torch.randn(B, N, Cin, generator=gen))
    ctr, idx = xyz[:, :S].contiguous(), d(torch.randint(0, N, (B, S, K), generator=gen).int())
    bias, gm, bt = d(torch.randn(Cout, generator=gen) * 0.1), d(1 + 0.1 * torch.randn(Cout, generator=gen)), d(0.1 * torch.randn(Cout, generator=gen))
    W0, b0 = d(torch.randn(Cin, 3, 1, generator=gen)), d(torch.randn(Cin, generator=gen) * 0.1)

    def both(W):
        a = ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, 0.3)
        torch.testing.assert_close(a, ops.group_mlp_max(xyz, fea, ctr, idx, W, bias, gm, bt, 0.3, generic=True), rtol=1e-4, atol=1e-4)
        b = ops.group_mlp_max_from_xyz(xyz, W0, b0, ctr, idx, W, bias, gm, bt, 0.3)
        torch.testing.assert_close(b, ops.group_mlp_max_from_xyz(xyz, W0, b0, ctr, idx, W, bias, gm, bt, 0.3, fused=True), rtol=1e-4, atol=1e-4)
        return a.clone(), b.clone()

    W = d(torch.randn(Cout, Cin + 3, 1, 1, generator=gen) / 4)
    a0, b0_ = both(W)
    a1, b1 = both(W)                         # second call: served from the cache
    assert torch.equal(a0, a1) and torch.equal(b0_, b1)
    W.mul_(-1.5)                             # in-place update (an optimiser step, load_state_dict)
    a2, _ = both(W)
    assert not torch.allclose(a0, a2)
    b0.add_(0.25)                            # any of the sources
    both(W)
    for _ in range(3):                       # a fresh tensor, most likely at the freed one's address
        del W
        W = d(torch.randn(Cout, Cin + 3, 1, 1, generator=gen) / 4)
        both(W)


def test_derived_cache_overflow_evicts_dead_weights_first_and_ends_the_generation_otherwise(ops):
    """A long-lived process that re-creates models reaches the cache's 512 entries: entries of freed weights go silently;
    when LIVE entries have to go, the generation captured decoder graphs are stamped with must change (they replay kernels that
    read those tensors)."""
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(64, 32, generator=gen).to(DEV)
    g0 = ops.derived_generation()
    for _ in range(600):                       # ad-hoc weights, each freed at once: only dead entries pile up
        W = (torch.randn(32, 32, generator=gen) / 6).to(DEV)
        ops.linear_bf16x3(x, W)
        del W
    assert ops.derived_generation() == g0 and len(ops._DERIVED) < 512
    keep = [(torch.randn(32, 32, generator=gen) / 6).to(DEV) for _ in range(520)]   # live weights beyond the bound
    want = ops.linear_bf16x3(x, keep[0]).clone()
    for W in keep:
        ops.linear_bf16x3(x, W)
    assert ops.derived_generation() > g0
    assert torch.equal(ops.linear_bf16x3(x, keep[0]), want)      # re-derived after the clear, same result
    torch.cuda.synchronize()


def test_prepare_and_channel_first(ops):
    gen = torch.Generator().manual_seed(5)
    pts = torch.randn(3, 4, 1000, generator=gen)
    pad = torch.arange(1000).unsqueeze(0) >= torch.tensor([[1000], [640], [1]])
    xyz, lens = ops.prepare_points(pts.to(DEV), pad.to(DEV))
    assert torch.equal(xyz.cpu(), pts[:, :3].transpose(1, 2).contiguous())
    assert lens.cpu().tolist() == [1000, 640, 1]
    x = torch.randn(2, 77, 131, generator=gen)
    assert torch.equal(ops.to_channel_first(x.to(DEV)).cpu(), x.transpose(1, 2).contiguous())


def test_operator_tables_match_reference_semantics():
    """Sampler / Querier mirrors (deeppointmap_amd/operators.py) against the oracle's restatement."""
    from deeppointmap_amd.operators import Querier, Sampler
    gen = torch.Generator().manual_seed(40)
    pts = torch.rand(2, 3000, 3, generator=gen)
    pad = torch.arange(3000).unsqueeze(0) >= torch.tensor([[3000], [2100]])
    new, mask = Sampler("fps-t3d")(points=pts.to(DEV), points_padding=pad.to(DEV), K=400)
    wn, wm, _ = O.fps(pts, pad, 400)
    assert torch.equal(new.cpu(), wn) and torch.equal(mask.cpu(), wm)
    ctr = wn.contiguous()
    idx = Querier("hybrid-t3d")(radius=0.15, K=32, points=pts.to(DEV), centers=ctr.to(DEV), points_padding=pad.to(DEV))
    assert idx.dtype == torch.int64
    want = O.hybrid_query(0.15, 32, pts, ctr, pad)
    for b in range(2):
        assert idx_rows_equal_as_sets(idx[b].cpu().numpy(), want[b].numpy()).all()
    # knn: no radius mask
    kn = Querier("knn")(K=8, points=pts.to(DEV), centers=ctr.to(DEV), points_padding=pad.to(DEV)).cpu()
    p2 = O.push_padding_far(pts, pad)
    wk = torch.topk(O.expanded_sqdist(ctr, p2), 8, dim=-1, largest=False)[1]
    for b in range(2):
        assert idx_rows_equal_as_sets(kn[b].numpy(), wk[b].numpy()).all()
    # ball: first K indices within the radius, padded with the first (utils.py:57-73)
    bl = Querier("ball")(radius=0.2, K=16, points=pts.to(DEV), centers=ctr.to(DEV), points_padding=pad.to(DEV)).cpu()
    d = O.expanded_sqdist(ctr, p2)
    gi = torch.arange(3000).view(1, 1, -1).repeat(2, 400, 1)
    gi[d > 0.2 ** 2] = 3000
    gi = gi.sort(dim=-1)[0][:, :, :16]
    first = gi[:, :, :1].expand_as(gi)
    gi = torch.where(gi == 3000, first, gi)
    valid = first[..., 0] < 3000  # rows with at least one point in range (the reference indexes out of range otherwise)
    assert torch.equal(bl[valid], gi[valid])


def _voxel_cases():
    import sys
    from conftest import GOLDEN
    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    import voxel_cases
    return voxel_cases.cases()


def test_voxel_sampler_vs_reference_fixture():
    """Sampler('voxel') (utils.py:150-207) against the reference's own outputs (tests/golden/make_golden_voxel.py): the
    same points in the same order, bit for bit -- both torch.topk branches over tied voxel populations, K=None, ragged
    frames with feature channels, fewer voxels than K, an empty frame, and the exact-distance-tie frames (lattices,
    duplicated points) whose pick follows the reference's unstable torch.sort."""
    from deeppointmap_amd.operators import Sampler
    g = load_golden("voxel_sampler.npz")
    cases = _voxel_cases()
    assert len(cases) >= 10
    for name, (pts, pad, K, vs, sr) in cases.items():
        new, mask = Sampler("voxel")(points=pts.to(DEV), points_padding=pad.to(DEV), K=K, voxel_size=vs, sample_range=sr)
        assert new.dtype == torch.float32 and mask.dtype == torch.bool
        assert tuple(new.shape) == g[name + ".sampled"].shape, name
        assert np.array_equal(new.cpu().numpy(), g[name + ".sampled"]), name
        assert np.array_equal(mask.cpu().numpy(), g[name + ".mask"]), name


def test_voxel_sampler_fuzz_vs_oracle(ops):
    """random clouds / grids / K against the oracle's restatement (itself held to the reference fixture on the CPU), and
    the slow-path flag: set on lattice frames, clear on generic ones"""
    g = torch.Generator().manual_seed(77)
    flagged = 0
    for trial in range(40):
        B = 1 if trial % 5 == 0 else int(torch.randint(1, 5, (1,), generator=g))
        N = int(torch.randint(40, 9000, (1,), generator=g))
        D = 3 + trial % 3
        pts = torch.rand(B, N, D, generator=g) * 2 - 1
        pts[..., :3] *= torch.tensor([1.3, 0.9, 0.2])
        lattice = trial % 4 == 1
        if lattice:
            pts[..., :3] = torch.round(pts[..., :3] * 8) / 8
        pad = torch.zeros(B, N, dtype=torch.bool)
        if trial % 2:
            for b in range(B):
                pad[b, int(torch.randint(1, N, (1,), generator=g)):] = True
        vs, sr = [0.3, 0.07, 0.15, 0.03][trial % 4], [1.0, 0.6, 1.5][trial % 3]
        K = None if B == 1 else [8, 50, 400, 3000][trial % 4]
        want, wmask, widx = O.voxel_sample(pts, pad, K, vs, sr)
        sel, n_unique = ops.voxel_sample(pts.to(DEV), pad.to(DEV), K, vs, sr)
        assert torch.equal(sel.cpu().long(), widx), (trial, B, N, K, vs, sr)
        flagged += int(lattice)
    assert flagged >= 5
    # the slow path is taken exactly when a voxel's nearest distance is shared
    lib_pts = torch.rand(1, 2000, 3, generator=g) * 2 - 1
    from deeppointmap_amd import _lib
    from deeppointmap_amd.ops import _ptr, _stream
    for lattice in (False, True):
        p = (torch.round(lib_pts * 8) / 8 if lattice else lib_pts).to(DEV).contiguous()
        padm = torch.zeros(1, 2000, dtype=torch.uint8, device=DEV)
        hdr = torch.empty(1, 8, device=DEV)
        lib = _lib.load()
        _lib.check(lib.dpm_voxel_sampler_bounds(_ptr(p), _ptr(padm), 1, 2000, 3, 0.3, 1.0, _ptr(hdr), _stream(p)), "bounds")
        cells = int(hdr[0, 3:6].double().prod().item())
        ws = torch.empty(lib.dpm_voxel_sampler_workspace_bytes(1, 2000, cells), dtype=torch.uint8, device=DEV)
        sel = torch.empty(1, 16, dtype=torch.int32, device=DEV)
        nu = torch.empty(1, dtype=torch.int32, device=DEV)
        _lib.check(lib.dpm_voxel_sampler_select(_ptr(p), _ptr(padm), 1, 2000, 3, 0.3, 1.0, _ptr(hdr), cells, 16, _ptr(sel), 16,
                                                _ptr(nu), _ptr(ws), _stream(p)), "select")
        assert float(hdr[0, 7]) == (1.0 if lattice else 0.0) and float(hdr[0, 6]) == 0.0
    with pytest.raises(ValueError):   # a grid finer than the cell budget is refused, not truncated
        ops.voxel_sample(lib_pts.to(DEV), torch.zeros(1, 2000, dtype=torch.bool, device=DEV), 8, 1e-4, 1.0)


def test_c_abi_from_a_program_without_torch():
    """tests/c/abi_gpu_smoke.cpp: a HIP host program (hipMalloc / hipMemcpy, its own stream; no Python, no torch) calls
    dpm_fps through include/dpm_hip.h and checks every pick of a ragged batch against a plain loop of the reference's rule.
    The binary is built with the library (csrc/build.py::build_abi_smoke, __graft_entry__.build()) and only RUN here."""
    import os
    import subprocess
    from deeppointmap_amd.csrc import build
    exe = os.path.join(build.HERE, "build", "abi_gpu_smoke")
    if not os.path.exists(exe):
        pytest.skip("csrc/build/abi_gpu_smoke is built by __graft_entry__.build() / python deeppointmap_amd/csrc/build.py")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "c abi gpu ok" in out.stdout


def test_pwconv_pair_kernel_vs_two_fused_layers_and_fp64(ops):
    """InvResMLP's pw_conv pair of the first level as one kernel (dpm_pwconv_pair_bf16x3: the 128-wide intermediate stays in
    registers) against the two Linear + LayerNorm calls it replaces and against an fp64 evaluation; ragged row counts, with
    and without the residual, values over many binades."""
    gen = torch.Generator(device=DEV).manual_seed(77)
    C, H = 32, 128
    W1 = torch.randn(H, C, 1, device=DEV, generator=gen) / C ** 0.5
    W2 = torch.randn(C, H, 1, device=DEV, generator=gen) / H ** 0.5
    b1, g1, be1 = (torch.randn(H, device=DEV, generator=gen) for _ in range(3))
    b2, g2, be2 = (torch.randn(C, device=DEV, generator=gen) for _ in range(3))
    for R, with_post in [(262144, True), (1000, True), (17, False), (16 * 4 * 4 * 3 + 5, True)]:
        x = torch.randn(R, C, device=DEV, generator=gen) * torch.exp2(torch.randint(-6, 6, (R, 1), device=DEV, generator=gen).float())
        post = torch.randn(R, C, device=DEV, generator=gen) if with_post else None
        got = ops.pwconv_pair(x, W1, b1, g1, be1, W2, b2, g2, be2, post)
        assert got is not None and got.shape == x.shape
        u = ops.linear_layernorm(x, W1, b1, g1, be1, act=ops.ACT_RELU)
        two = ops.linear_layernorm(u, W2, b2, g2, be2, act=ops.ACT_RELU, post=post)
        n = min(R, 4096)
        xd = x[:n].double()
        h = torch.relu(torch.nn.functional.layer_norm(xd @ W1[:, :, 0].double().t() + b1.double(), (H,), g1.double(), be1.double(), 1e-5))
        y = torch.nn.functional.layer_norm(h @ W2[:, :, 0].double().t() + b2.double(), (C,), g2.double(), be2.double(), 1e-5)
        want = torch.relu(y + (post[:n].double() if with_post else 0.0))
        e_one, e_two = (got[:n].double() - want).abs().max(), (two[:n].double() - want).abs().max()
        assert float(e_one) < 2e-5 and float(e_one) < 2.0 * float(e_two) + 2e-6, (R, float(e_one), float(e_two))
        torch.testing.assert_close(got, two, rtol=2e-5, atol=2e-5)
    # a row's result does not depend on the rows it travels with
    a = ops.pwconv_pair(x[:37].contiguous(), W1, b1, g1, be1, W2, b2, g2, be2, post[:37].contiguous())
    assert torch.equal(a, got[:37])
