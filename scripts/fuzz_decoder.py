"""One-off fuzz of Decoder.registration_forward / loop_detection_forward against the oracle on random token counts
(also odd ones: the kernels' tile edges), with and without padding masks."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O

torch.set_grad_enabled(False)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
dump = sys.argv[3] if len(sys.argv) > 3 else None   # directory: every case above half the pose tolerance is written there (inputs + both results)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import margin_cases   # the random stream of a seed lives there: tests/golden/margin.npz refers to its cases by (seed, ordinal)
stream = margin_cases.Stream(seed)
cfg = default_args()
dec = init_procedural(Decoder(cfg)).to("cuda:0")
sd = {k: v.detach().cpu() for k, v in dec.flat().items()}


def rot_angle(A, B):
    M = A.double().T @ B.double()
    return float(np.arctan2(float(torch.linalg.norm(M - M.T)) / (2 * 2 ** 0.5), float((torch.trace(M) - 1) / 2)))


t0, n, bad, worst = time.time(), 0, 0, (0.0, 0.0)
while time.time() - t0 < budget:
    _, s, d, ms_, md_, ns = stream.next()
    M, N = s.shape[1], d.shape[1]
    masks = (ms_, md_)
    tr, tro = {}, {}
    R, T, conf, rmse = dec.registration_forward(s, d, masks[0], masks[1], num_sample=ns, trace=tr)
    Ro, To, co, ro = O.registration_forward(sd, cfg, s, d, ns, trace=tro, src_padding_mask=masks[0], dst_padding_mask=masks[1])
    dT, dR = float((T.cpu() - To).norm()), rot_angle(R.cpu(), Ro)
    worst = (max(worst[0], dT), max(worst[1], dR))
    n = stream.n
    if dump and (dT > 5e-5 or dR > 5e-5 or conf.numel() != co.numel()):
        os.makedirs(dump, exist_ok=True)
        np.savez_compressed(os.path.join(dump, f"case_s{seed}_n{n}.npz"), src=s.numpy(), dst=d.numpy(), num_sample=np.float64(ns), num_sample_is_int=isinstance(ns, int),
                            src_mask=(masks[0].numpy() if masks[0] is not None else np.zeros(0, bool)), dst_mask=(masks[1].numpy() if masks[1] is not None else np.zeros(0, bool)),
                            hip_R=R.cpu().numpy(), hip_T=T.cpu().numpy(), hip_n=conf.numel(), hip_rmse=float(rmse), oracle_R=Ro.numpy(), oracle_T=To.numpy(), oracle_n=co.numel(), dT=dT, dR=dR)
    if dT > 1e-4 or dR > 1e-4 or conf.numel() != co.numel():
        # classify: same pairs selected?  was one of the oracle's inlier decisions at rounding level (margin = relative
        # distance of the closest residual to the cut mean + 3 std)?  was the k-th pair confidence (nearly) tied?
        pairs_g = set(zip(tr["src_index"].flatten().tolist(), tr["dst_index"].flatten().tolist()))
        pairs_o = set(zip(tro["src_index"].flatten().tolist(), tro["dst_index"].flatten().tolist()))
        cs = tro["conf"].flatten().sort(descending=True)[0]
        kgap = float((cs[-2] - cs[-1]) / cs[-1]) if cs.numel() > 1 else 1.0
        # pair confidences agree with the reference to ~1e-3 relative (fp32 sums in another order through three attention
        # blocks, then exp(./0.1) twice): a k-th confidence closer than that to its neighbour can fall either way
        # ... and so can the 64 seed correspondences of the Kabsch loop (decoder.py:233-235: topk(w, 64)) when no weight
        # exceeds 0.5 and the 64th / 65th DISTINCT weights are closer than the confidences' agreement
        ws = tro["w"].unique().sort(descending=True)[0]
        sgap = 1.0
        if int((tro["w"] > 0.5).sum()) < 64 and tro["w"].numel() > 64:
            w64 = tro["w"].sort(descending=True)[0][63]
            below = ws[ws < w64]
            sgap = float((w64 - below[0]) / w64) if below.numel() else 1.0
        cond = min(tro["margins"]) < 1e-4 or (pairs_g != pairs_o and kgap < 2e-3) or sgap < 1e-4
        bad += 0 if cond else 1
        print(f"{'ill-conditioned' if cond else 'MISMATCH'} seed {seed}: M {M} N {N} num_sample {ns} masks {masks[0] is not None}: dT {dT:.2e} dR {dR:.2e} "
              f"inliers {conf.numel()} vs {co.numel()} rmse {rmse:.4f} vs {ro:.4f}; same pairs {pairs_g == pairs_o} "
              f"(k-th confidence gap {kgap:.1e}), n_corr {tr['n_corr']} vs {tro['w'].numel()}, oracle margins {[f'{m:.1e}' for m in tro['margins']]}")
        if not cond:
            from deeppointmap_amd import ops
            # the Kabsch kernel on the ORACLE's correspondences: is the difference made before it or inside it?
            res = ops.corr_kabsch(None, tro["src"].t().contiguous().cuda(), tro["dst"].t().contiguous().cuda(), None, None, tro["w"].cuda(), 2.0).cpu()
            print(f"   kernel Kabsch on the oracle's correspondences: dT {float((res[9:12].view(3, 1) - To).norm()):.2e}, inliers {int(res[14])} vs {co.numel()}, iterations {int(res[15])} vs {len(tro['margins'])}")
            go = {(a, b): (c, o) for a, b, c, o in zip(tr["src_index"].flatten().tolist(), tr["dst_index"].flatten().tolist(), tr["conf"].flatten().tolist(), tr["offsets"].cpu().view(-1, 3).tolist())}
            k = tro["conf"].numel()
            dc = max(abs(go[(a, b)][0] - c) / c for a, b, c in zip(tro["src_index"].tolist(), tro["dst_index"].tolist(), tro["conf"].tolist()) if (a, b) in go)
            print(f"   pair confidences: max relative difference {dc:.2e}; offsets tensor {tuple(tr['offsets'].shape)}; inlier weights equal as multisets: {sorted(conf.cpu().tolist()) == sorted(co.tolist())}; "
                  f"|x| diff {float((tr['x'].cpu().view(-1) - tro['x'].reshape(-1)).abs().max()):.2e}")
            w = tro["w"]; print(f"   weights: {w.numel()} values, {w.unique().numel()} distinct; 64th/65th largest {w.sort(descending=True)[0][62:66].tolist()}; > 0.5: {int((w > 0.5).sum())}")
    if n % 5 == 0:  # loop detection on a small batch of the same shapes
        S, D = stream.loop_batch(M, N)
        C = S.shape[0]
        p, po = dec.loop_detection_forward(S, D).cpu(), O.loop_detection_forward(sd, cfg, S, D)
        if float((p - po).abs().max()) > 5e-5:
            bad += 1
            print(f"LOOP MISMATCH seed {seed}: C {C} M {M} N {N}: {float((p - po).abs().max()):.2e}")
print(f"seed {seed}: {n} registrations, {bad} mismatches, worst dT {worst[0]:.2e} m dR {worst[1]:.2e} rad, {time.time() - t0:.0f} s")
