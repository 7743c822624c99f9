"""`Encoder` -- drop-in for the reference's network/encoder/encoder.py::Encoder.

Same constructor (`Encoder(args)`), same state-dict keys/shapes (params.encoder_shapes), same
call contract: `encoder(points (B,C>=3,N) f32, points_padding (B,N) bool) ->
[coor (B,3,S), fea (B,out_channel,S), padding (B,S)]` (encoder.py:51-69).  Inputs may arrive on
the CPU (ScanPack keeps CPU tensors, system/modules/pose_graph.py:43-45); they are staged to the
module's GPU.  All arithmetic runs in libdpm_hip.so; there is no torch fallback.

Internally everything is point-major fp32 with a per-frame valid length (valid points lead,
exactly what the reference's FPS assumes, utils.py:255).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from .params import ParamTree, encoder_shapes


class Encoder(ParamTree):
    def __init__(self, args):
        super().__init__(encoder_shapes(args))
        self.args = args
        self.encoder_cfg = args.encoder
        self.in_channel = self.encoder_cfg.in_channel
        self.out_channel = self.encoder_cfg.out_channel
        self.downsample_layers = len(self.encoder_cfg.npoint)
        self.upsample_layers = self.encoder_cfg.upsample_layers
        for s in self.encoder_cfg.sample:
            if s["type"] not in ("fps", "fps-t3d", "voxel"):  # pointnext.py:21
                raise ValueError(f"{dict(s)} is not a supported sampling way, please use 'fps' or 'voxel'")
            if s["type"] == "voxel" and not ("size" in s and "range" in s):  # pointnext.py:30-31
                raise ValueError("a voxel sampler needs 'size' and 'range'")
        self.all_fps = all(s["type"] != "voxel" for s in self.encoder_cfg.sample)
        # Farthest point sampling is nested: level i+1 samples the level-i picks starting from the same first point,
        # and the k-th level-i pick is the farthest of ALL level-(i-1) points from the picks before it -- in
        # particular the farthest among the level-i picks themselves -- so level i+1 reproduces the level-i pick
        # ORDER: its result is the first npoint[i+1] picks of level i (ties included: the first-index rule on
        # positions agrees with the pick order).  Only the first level is computed; False runs every level
        # (tests assert both give identical tensors).
        self.nested_fps = self.all_fps and all(b <= a for a, b in zip(self.encoder_cfg.npoint, self.encoder_cfg.npoint[1:]))
        # presample() also answers the neighbour queries (coordinates only): a pipeline knob -- where they run moves
        # work between the geometry and the feature stage, the results are the same tensors either way
        self.presample_neighbours = False
        # ... or only the queries of the downsampling levels from this one on (None: none): the lower levels' searches are
        # microseconds of work each but launches of the feature stream's dependent chain, and the geometry streams have slack
        self.presample_neighbours_from = None
        self._price_tail = 0   # measurement only (scripts/price_tail.py): extra evaluations of levels 3+ and the upsamplers
        self.eval()

    # -- helpers -------------------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self.p("point_mlp0.weight").device

    def _mlp_ln(self, x: torch.Tensor, conv: str, ln: str, act: int, post: Optional[torch.Tensor] = None):
        return ops.linear_layernorm(x, self.p(conv + ".weight"), self.p(conv + ".bias"), self.p(ln + ".weight"),
                                    self.p(ln + ".bias"), act=act, post=post)

    def _group(self, prefix: str, radius: float, xyz, fea, centers, idx):
        return ops.group_mlp_max(xyz, fea, centers, idx, self.p(prefix + ".0.weight"), self.p(prefix + ".0.bias"),
                                 self.p(prefix + ".1.ln.weight"), self.p(prefix + ".1.ln.bias"), radius)

    # -- forward -------------------------------------------------------------------------------
    def _sample_level(self, i: int, xyz: torch.Tensor, lengths: torch.Tensor):
        """The sampler of downsampling stage i (pointnext.py:29-35, 45-46) -> (idx (B,K) int32 into the level's points,
        -1 = padding; new_xyz (B,K,3), zero rows at padding; new_lengths (B,)).  'voxel' stages (no shipped config has
        one) go through the voxel-grid kernels of csrc/voxel_sample.hip, which read the grid size back once per call."""
        st, k = self.encoder_cfg.sample[i], self.encoder_cfg.npoint[i]
        if st["type"] != "voxel":
            return ops.fps(xyz, lengths, k)
        N = xyz.shape[1]
        pad = torch.arange(N, device=xyz.device).unsqueeze(0) >= lengths.unsqueeze(1)
        sel, _ = ops.voxel_sample(xyz, pad, k, st["size"], st["range"])
        mask = sel < 0
        new = torch.gather(xyz, 1, sel.clamp(min=0).long().unsqueeze(-1).expand(-1, -1, 3)).masked_fill(mask.unsqueeze(-1), 0.0)
        return sel, new.contiguous(), (~mask).sum(1).to(torch.int32)

    @torch.no_grad()
    def sample_first_level(self, points_list, padding_list):
        """First-level farthest point sampling of SEVERAL batches in one launch (the sampling kernel runs one wave per
        frame: more frames per launch = more of the chip's latency-bound chains in flight).  Returns one
        (idx, new_xyz, new_lengths) per batch, views of the joint result."""
        dev = self.device
        with torch.cuda.device(dev):
            pts = torch.cat([p.to(device=dev, dtype=torch.float32) for p in points_list], 0)
            pad = torch.cat([p.to(device=dev) for p in padding_list], 0)
            xyz, lengths = ops.prepare_points(pts.contiguous(), pad.contiguous())
            fidx, new, nl = self._sample_level(0, xyz, lengths)
        out, o = [], 0
        for p in points_list:
            b = p.shape[0]
            out.append((fidx[o:o + b], new[o:o + b], nl[o:o + b]))
            o += b
        return out

    @torch.no_grad()
    def presample(self, points: torch.Tensor, points_padding: torch.Tensor, levels: Optional[int] = None,
                  sampled0=None) -> dict:
        """Input staging + the whole farthest-point-sampling chain (all levels), on the CURRENT stream.
        Sampling depends on coordinates only (level i+1 samples the points level i kept), never on features;
        it is a serial chain of dependent rounds that occupies one CU per frame, so a streaming caller runs it
        for batch i+1 on a side stream while batch i finishes on the main stream (pipeline.HotPath.submit).
        Pass the result to forward(..., presampled=...).  `levels` limits the pass to the first FPS levels (the
        remaining, much shorter ones then run inside forward) -- a knob for balancing pipeline stages.
        `sampled0` = (idx, new_xyz, new_lengths) of the first level when the caller sampled it already."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("deeppointmap_amd.Encoder runs on the GPU only: call .to('cuda') first "
                               "(there is no CPU fallback)")
        with torch.cuda.device(dev):
            pts = points.to(device=dev, dtype=torch.float32).contiguous()
            pad = points_padding.to(device=dev).contiguous()
            xyz, lengths = ops.prepare_points(pts, pad)
            out = dict(pts=pts, xyz=xyz, lengths=lengths)
            n_levels = len(self.encoder_cfg.npoint) if levels is None else levels
            npoint = list(self.encoder_cfg.npoint[:n_levels])
            if n_levels >= 1 and (self.nested_fps or n_levels == 1):
                first = sampled0 if sampled0 is not None else self._sample_level(0, xyz, lengths)
                # every lower level is a prefix of the first level's picks: one bookkeeping launch for all of them
                lower = ops.nested_levels(first[1], first[2], npoint[1:]) if n_levels > 1 else []
                for i, (fidx, cur, cur_len) in enumerate([first] + lower):
                    out[f"fidx{i}"], out[f"xyz{i}"], out[f"len{i}"] = fidx, cur, cur_len
            else:
                cur, cur_len = xyz, lengths
                for i, k in enumerate(npoint):
                    fidx, cur, cur_len = sampled0 if (i == 0 and sampled0 is not None) else self._sample_level(i, cur, cur_len)
                    out[f"fidx{i}"], out[f"xyz{i}"], out[f"len{i}"] = fidx, cur, cur_len
            # The search grids of the neighbour queries depend on coordinates and radii only: they are sorted here,
            # next to the sampling, and forward() runs just the searches.  Same bookkeeping as forward(): a
            # SetAbstraction whose (radius, K) the previous level's LocalAggregation already answered needs none.
            grids, pts_i, len_i, answered = {}, xyz, lengths, set()
            for i in range(n_levels):
                radii, ks = self.encoder_cfg.radius_list[i], self.encoder_cfg.nsample_list[i]
                if (float(radii[0]), int(ks[0])) not in answered and pts_i.shape[1] >= ops.GRID_MIN_N:
                    grids[("sa", i)] = ops.knn_grid(pts_i, len_i, radii[0])
                pts_i, len_i, answered = out[f"xyz{i}"], out[f"len{i}"], set()
                for j in range(1, len(radii)):
                    key = (float(radii[j]), int(ks[j]))
                    if key not in answered and pts_i.shape[1] >= ops.GRID_MIN_N:
                        grids[("la", i, key)] = ops.knn_grid(pts_i, len_i, radii[j])
                    answered.add(key)
            out["grids"] = grids
            if n_levels == len(self.encoder_cfg.npoint):
                if self.presample_neighbours:
                    out["knn"] = self._neighbour_queries(out)
                elif self.presample_neighbours_from is not None:
                    out["knn"] = self._neighbour_queries(out, first_level=int(self.presample_neighbours_from))
        return out

    def _neighbour_queries(self, samp: dict, first_level: int = 0) -> dict:
        """The neighbour queries of the downsampling levels from `first_level` on -- they depend on coordinates only, like the
        sampling.  Same reuse rules as forward(): ("sa", i) / ("la", i, (radius, K)) -> idx (a reused answer is a copy of rows
        of an identical query: the same tensors whether a level's first query finds a predecessor here or not)."""
        enc, grids, knn = self.encoder_cfg, samp.get("grids", {}), {}
        xyz, lengths, self_q = samp["xyz"], samp["lengths"], {}
        for i in range(len(enc.npoint)):
            radii, ks = enc.radius_list[i], enc.nsample_list[i]
            fidx, new_xyz, new_len = samp[f"fidx{i}"], samp[f"xyz{i}"], samp[f"len{i}"]
            if i < first_level:
                xyz, lengths, self_q = new_xyz, new_len, {}
                continue
            prev = self_q.get((float(radii[0]), int(ks[0])))
            knn[("sa", i)] = ops.knn_hybrid(xyz, lengths, new_xyz, ks[0], radii[0], reuse_idx=prev,
                                            center_src=fidx if prev is not None else None,
                                            grid=grids.get(("sa", i)) if prev is None else None)
            self_q = {}
            for j in range(1, len(radii)):
                key = (float(radii[j]), int(ks[j]))
                if key not in self_q:
                    self_q[key] = ops.knn_hybrid(new_xyz, new_len, new_xyz, ks[j], radii[j], grid=grids.get(("la", i, key)))
                knn[("la", i, key)] = self_q[key]
            xyz, lengths = new_xyz, new_len
        return knn

    @torch.no_grad()
    def forward(self, points: torch.Tensor, points_padding: torch.Tensor, trace: Optional[dict] = None,
                presampled: Optional[dict] = None, descriptor_scale: float = 0.0, spare_frames: int = 0,
                stop_level: Optional[int] = None, resume: Optional[dict] = None) -> List[torch.Tensor]:
        """-> [coor (B,3,S), fea (B,out_channel,S), padding (B,S)] (encoder.py:51-69).  descriptor_scale > 0 (not in the
        reference signature; used by the batched hot path): return instead the unified descriptor (B,out_channel+3,S) =
        [fea ; coor * descriptor_scale] that ExtractionThread.process builds from the triple (odometry.py:47-49).
        stop_level = i: run the downsampling levels below i and return the state (a dict) instead; resume = that state: run
        the rest (same kernels in the same order: the two halves may sit on different HIP streams, pipeline.py)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("deeppointmap_amd.Encoder runs on the GPU only: call .to('cuda') first "
                               "(there is no CPU fallback)")
        enc = self.encoder_cfg
        if resume is not None:
            samp = resume["samp"]
        else:
            samp = presampled if presampled is not None else self.presample(points, points_padding)
        with torch.cuda.device(dev):
            pts, xyz, lengths = samp["pts"], samp["xyz"], samp["lengths"]
            # level-0 features = point_mlp0(points).  With xyz-only input (every shipped config) they are only ever
            # consumed by the first SetAbstraction, which evaluates them inside its gather (fea = None here).
            w0 = self.p("point_mlp0.weight")
            fuse0 = (self.in_channel == 3 and w0.shape[0] % 4 == 0 and 2 * w0.shape[0] in (32, 64, 128)
                     and enc.nsample_list[0][0] in (16, 32))
            if resume is not None or fuse0:
                fea = None
            elif self.in_channel == 3:
                fea = ops.linear(xyz, w0, self.p("point_mlp0.bias"))
            else:  # extra input channels: point-major copy of the first in_channel rows
                fea = ops.linear(pts[:, :self.in_channel].transpose(1, 2).contiguous(), w0, self.p("point_mlp0.bias"))
            levels = [(xyz, fea, lengths)] if resume is None else resume["levels"]
            # Neighbour queries repeat: a SetAbstraction asks, for the FPS-picked subset of a level's points, exactly
            # what the preceding LocalAggregation answered for ALL of them when radius and K coincide (they do in every
            # shipped config), and consecutive LocalAggregations of a stage may share (radius, K) too.  `self_q`
            # remembers the self-queries of the current level: (radius, K) -> idx (B,N,K).
            self_q = {} if resume is None else resume["self_q"]
            for i, npoint in enumerate(enc.npoint):
                if resume is not None and i < resume["next"]:
                    continue
                if stop_level is not None and i >= stop_level:
                    return dict(samp=samp, levels=levels, self_q=self_q, next=i)
                xyz, fea, lengths = levels[-1]
                radii, ks = enc.radius_list[i], enc.nsample_list[i]
                pre = f"downsampler.{i}"
                if f"fidx{i}" in samp:
                    fidx, new_xyz, new_len = samp[f"fidx{i}"], samp[f"xyz{i}"], samp[f"len{i}"]
                else:  # levels the geometry pass left to this stream
                    fidx, new_xyz, new_len = self._sample_level(i, xyz, lengths)
                # (self._price_tail > 0: levels 3+ and the upsamplers are computed that many extra times, results identical --
                # scripts/price_tail.py measures what the launch-bound tail of the encoder costs a pipelined step)
                for _rep in range(1 + (self._price_tail if i >= 3 else 0)):
                    prev = self_q.get((float(radii[0]), int(ks[0])))
                    grids, knn = samp.get("grids", {}), samp.get("knn", {})
                    gidx = knn.get(("sa", i))
                    if gidx is None:
                        gidx = ops.knn_hybrid(xyz, lengths, new_xyz, ks[0], radii[0], reuse_idx=prev,
                                              center_src=fidx if prev is not None else None,
                                              grid=grids.get(("sa", i)) if prev is None else None)
                    self_q = {}  # from here on the level is the sampled one
                    if fea is None:
                        m = pre + ".sa.mlp"
                        new_fea = ops.group_mlp_max_from_xyz(xyz, w0, self.p("point_mlp0.bias"), new_xyz, gidx,
                                                             self.p(m + ".0.weight"), self.p(m + ".0.bias"),
                                                             self.p(m + ".1.ln.weight"), self.p(m + ".1.ln.bias"), radii[0])
                    else:
                        new_fea = self._group(pre + ".sa.mlp", radii[0], xyz, fea, new_xyz, gidx)
                    if trace is not None:
                        trace[pre + ".fps.idx"], trace[pre + ".fps.new"] = fidx, new_xyz
                        trace[pre + ".sa.idx"], trace[pre + ".sa.out"] = gidx, new_fea
                    for j in range(1, len(radii)):
                        q = f"{pre}.irm.{j - 1}"
                        key = (float(radii[j]), int(ks[j]))
                        if key not in self_q:
                            self_q[key] = knn[("la", i, key)] if ("la", i, key) in knn else ops.knn_hybrid(
                                new_xyz, new_len, new_xyz, ks[j], radii[j], grid=grids.get(("la", i, key)))
                        lidx = self_q[key]
                        t = self._group(q + ".la.mlp", radii[j], new_xyz, new_fea, new_xyz, lidx)
                        pair = ops.pwconv_pair(t, self.p(q + ".pw_conv.0.weight"), self.p(q + ".pw_conv.0.bias"),
                                               self.p(q + ".pw_conv.1.ln.weight"), self.p(q + ".pw_conv.1.ln.bias"),
                                               self.p(q + ".pw_conv.3.weight"), self.p(q + ".pw_conv.3.bias"),
                                               self.p(q + ".pw_conv.4.ln.weight"), self.p(q + ".pw_conv.4.ln.bias"), post=new_fea)
                        if pair is not None:     # the first level: both layers in one kernel, the 4C-wide intermediate in registers
                            new_fea = pair
                        else:
                            u = self._mlp_ln(t, q + ".pw_conv.0", q + ".pw_conv.1.ln", ops.ACT_RELU)
                            new_fea = self._mlp_ln(u, q + ".pw_conv.3", q + ".pw_conv.4.ln", ops.ACT_RELU, post=new_fea)
                        if trace is not None:
                            trace[q + ".la.idx"], trace[q + ".la.out"], trace[q + ".out"] = lidx, t, new_fea
                levels.append((new_xyz, new_fea, new_len))
            L = self.downsample_layers
            for i in range(self.upsample_layers):
                xyz1, fea1, len1 = levels[L - i - 1]
                xyz2, fea2, len2 = levels[-1]
                q = f"upsampler.{i}"
                for _rep in range(1 + self._price_tail):
                    x = ops.three_interp_cat(xyz1, xyz2, len2, fea1, fea2)
                    x = self._mlp_ln(x, q + ".mlp.0", q + ".mlp.1.ln", ops.ACT_RELU)
                    x = self._mlp_ln(x, q + ".mlp.3", q + ".mlp.4.ln", ops.ACT_RELU)
                if trace is not None:
                    trace[q + ".out"] = x
                levels.append((xyz1, x, len1))
            xyz, fea, lengths = levels[-1]
            coor, feat, padding, desc = ops.emit_descriptors(xyz, fea, lengths, descriptor_scale, spare_frames)
            if descriptor_scale > 0:
                return desc
        return [coor, feat, padding]
