"""Seeded INPUTS of the round-6 margin fixture (margin.npz): the registrations the decoder fuzz (scripts/fuzz_decoder.py) found
above HALF the pose tolerance against the oracle on the GPU box, regenerated from (seed, ordinal) by replaying the fuzz's random
stream on the CPU -- the fixture stores only what the reference answered.  Imported by tests/golden/make_golden_margin.py (runs
the reference in fp32 and fp64 on them), by scripts/fuzz_decoder.py (which draws from the same stream) and by the tests.
No reference code here."""
import random

import torch


class Stream:
    """the random stream of one fuzz seed: `next()` -> (ordinal n, src (131, M), dst (131, N), src_mask, dst_mask, num_sample);
    `loop_batch(M, N)` is the draw scripts/fuzz_decoder.py makes after every fifth registration (it advances the stream)."""

    def __init__(self, seed: int):
        self.rng = random.Random(seed)
        self.g = torch.Generator().manual_seed(seed)
        self.n = 0

    def desc(self, n):
        fea = torch.rand(128, n, generator=self.g) * self.rng.choice([0.2, 1.0, 3.0])
        xyz = torch.cat([(torch.rand(2, n, generator=self.g) * 2 - 1) * 50, torch.randn(1, n, generator=self.g) * 2])
        return torch.cat([fea, xyz], 0)

    def next(self):
        rng = self.rng
        M = rng.choice([rng.randint(40, 300), 256, 512, rng.randint(300, 1100)])
        N = rng.choice([M, rng.randint(40, 300), 256])
        s, d = self.desc(M), self.desc(N)
        masks = (None, None)
        if rng.random() < 0.3:
            ms, md = torch.zeros(1, M, dtype=torch.bool), torch.zeros(1, N, dtype=torch.bool)
            ms[0, M - rng.randint(1, M // 3):] = True
            md[0, N - rng.randint(1, N // 3):] = True
            masks = (ms, md)
        ns = rng.choice([0.5, 0.5, 0.25, 64])
        self.n += 1
        return self.n, s, d, masks[0], masks[1], ns

    def loop_batch(self, M, N):
        C = self.rng.randint(1, 5)
        return torch.stack([self.desc(M) for _ in range(C)]), torch.stack([self.desc(N) for _ in range(C)])


def replay(seed: int, ordinals):
    """-> {n: (src, dst, src_mask, dst_mask, num_sample)} for the requested ordinals of one seed"""
    want, out, st = set(ordinals), {}, Stream(seed)
    while want:
        n, s, d, ms, md, ns = st.next()
        if n in want:
            out[n] = (s, d, ms, md, ns)
            want.discard(n)
        if n % 5 == 0:
            st.loop_batch(s.shape[1], d.shape[1])
    return out


# (seed, ordinal): every case of eight 100-second seeds on the GPU box (round 6, gpurun_out/r06a) whose pose differed from the
# oracle's by more than 5e-5 m / 5e-5 rad or whose inlier count differed.  Classes as the fuzz printed them:
#   'margin'    same pairs, same inliers, pose 0.5-2.2 tolerances apart (what this fixture is about)
#   'boundary'  a discrete decision of the reference sits at rounding level (k-th pair confidence, 64th seed weight, an inlier cut
#               or the offset-outlier cut): the two fp32 evaluations take different branches and the poses are centimetres apart
CASES = {
    (604, 184): "margin", (606, 59): "margin", (607, 97): "margin", (604, 132): "margin", (601, 201): "margin", (607, 2): "margin",
    (605, 223): "margin", (606, 260): "margin", (604, 74): "margin", (607, 115): "margin", (602, 52): "margin",
    (507, 234): "boundary", (603, 193): "boundary", (605, 87): "boundary", (605, 154): "boundary", (507, 52): "boundary",
    (507, 154): "boundary", (602, 53): "boundary", (603, 47): "boundary", (607, 126): "boundary",
}


def all_cases():
    """-> [((seed, n), class, (src, dst, src_mask, dst_mask, num_sample))] in a fixed order"""
    out = []
    for seed in sorted({s for s, _ in CASES}):
        got = replay(seed, [n for s, n in CASES if s == seed])
        for n in sorted(got):
            out.append(((seed, n), CASES[(seed, n)], got[n]))
    return out
