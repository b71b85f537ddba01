"""``Net`` -- drop-in for Pose_Estimation_Model/model/pose_estimation_model.py on MI355X.

Selected exactly like the reference model module:
``importlib.import_module(cfg.model_name).Net(cfg.model)`` (test_bop.py:203-204,
run_inference_custom.py:265-266) with ``--model sam6d_amd.pem.pose_estimation_model``.
Same constructor argument (the ``model:`` node of config/base.yaml), same dict-in/dict-out
forward, same state_dict keys (checked against the reference in tests/).

One addition: if ``end_points['coarse_rand_u']`` (B, 3*nproposal1) is present it replaces the
``torch.rand`` draw of compute_coarse_Rt (model_utils.py:219) so results are reproducible
across devices; otherwise uniforms are drawn on the device like the reference does.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import policy
from .feature_extraction import ViTEncoder
from .layers import GeometricStructureEmbedding, GeometricTransformer, SparseToDenseTransformer, plinear
from .solvers import coarse_Rt, fine_Rt


def _get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def feature_similarity(f1, f2, temp):
    """compute_feature_similarity (model_utils.py:114-136), cosine, normalize_feat=True."""
    return F.normalize(f1, p=2, dim=2) @ F.normalize(f2, p=2, dim=2).transpose(1, 2) / temp


def sample_pts_feats(pts, feats, npoint):
    """model_utils.py:53-66 without the four transposes: FPS + two row gathers."""
    idx = ops.furthest_point_sampling(pts.contiguous(), npoint)
    return ops.gather_rows(pts.contiguous(), idx), ops.gather_rows(feats.contiguous(), idx), idx


class CoarsePointMatching(nn.Module):
    """coarse_point_matching.py:14-81 (inference branch)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.nblock = cfg.nblock
        self.in_proj = nn.Linear(cfg.input_dim, cfg.hidden_dim)
        self.out_proj = nn.Linear(cfg.hidden_dim, cfg.out_dim)
        self.bg_token = nn.Parameter(torch.randn(1, 1, cfg.hidden_dim) * .02)
        self.transformers = nn.ModuleList([GeometricTransformer(cfg.hidden_dim) for _ in range(self.nblock)])

    def forward(self, p1, f1, geo1, p2, f2, geo2, radius, end_points):
        B = f1.size(0)
        bg = self.bg_token.expand(B, -1, -1)
        f1 = torch.cat([bg, plinear(self, self.in_proj, f1)], dim=1)
        f2 = torch.cat([bg, plinear(self, self.in_proj, f2)], dim=1)
        for blk in self.transformers:
            f1, f2 = blk(f1, geo1, f2, geo2)
        atten = feature_similarity(plinear(self, self.out_proj, f1), plinear(self, self.out_proj, f2), self.cfg.temp)
        n1 = self.cfg.nproposal1
        rand_u = end_points.get("coarse_rand_u")
        if rand_u is None:
            rand_u = torch.rand(B, n1 * 3, device=p1.device)
        model = end_points["model"] / (radius.reshape(-1, 1, 1) + 1e-6)
        end_points["init_R"], end_points["init_t"] = coarse_Rt(atten, p1, p2, model, rand_u, n1, self.cfg.nproposal2)
        return end_points


class _ConvBN(nn.Module):
    """pytorch_utils.Conv2d with bn=True: 1x1 conv (no bias) -> BatchNorm2d -> ReLU; key names
    ``conv.weight`` and ``normlayer.bn.*`` (pytorch_utils.py:86-134)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=(1, 1), bias=False)
        self.normlayer = nn.Sequential()
        self.normlayer.add_module("bn", nn.BatchNorm2d(cout))

    def folded(self):
        """(W (cout,cin), b (cout)) of conv+BN(eval) folded into one affine map."""
        bn = self.normlayer.bn
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return self.conv.weight.flatten(1) * s.unsqueeze(1), bn.bias - bn.running_mean * s


class _SharedMLP(nn.Module):
    def __init__(self, dims):
        super().__init__()
        for i in range(len(dims) - 1):
            self.add_module(f"layer{i}", _ConvBN(dims[i], dims[i + 1]))

    def layers(self):
        return [m for m in self.children()]


class _Conv1dNoAct(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, kernel_size=1, bias=True)

    # the kernel-size-1 convolution seen as the Linear it is (plinear reads .weight (N,K) and .bias)
    @property
    def weight(self):
        return self.conv.weight.squeeze(-1)

    @property
    def bias(self):
        return self.conv.bias

    def forward(self, x):                                   # (..., cin) -> (..., cout): the library statement of the same map
        return F.linear(x, self.weight, self.bias)


class PositionalEncoding(nn.Module):
    """fine_point_matching.py:90-125: two ball-query groupings (r, nsample) -> SharedMLP
    [6,32,64,128] (BN folded at eval) -> max over the group -> concat -> 1x1 conv."""

    def __init__(self, out_dim, r1=0.1, r2=0.2, nsample1=32, nsample2=64):
        super().__init__()
        self.scales = ((r1, nsample1), (r2, nsample2))
        self.mlp1 = _SharedMLP([6, 32, 64, 128])
        self.mlp2 = _SharedMLP([6, 32, 64, 128])
        self.mlp3 = _Conv1dNoAct(256, out_dim)

    def forward(self, pts):
        pts = pts.contiguous()
        B, N, _ = pts.shape
        outs = []
        for mlp, (r, ns) in zip((self.mlp1, self.mlp2), self.scales):
            idx = ops.ball_query(pts, pts, r, ns)                                   # (B,N,ns) i32
            if policy.guard("pem.PositionalEncoding", cuda=pts.is_cuda, have=ops.have("pe_group"), eval_mode=not self.training):
                (W0, b0), (W1, b1), (W2, b2) = (layer.folded() for layer in mlp.layers())
                outs.append(ops.pe_group_mlp(pts, idx, W0, b0, W1, b1, W2, b2))     # (B,N,128), nothing else hits HBM
                continue
            nbr = ops.gather_rows(pts, idx.view(B, N * ns)).view(B, N, ns, 3)       # absolute xyz
            x = torch.cat([nbr - pts.unsqueeze(2), nbr], dim=-1)                    # (B,N,ns,6)
            for layer in mlp.layers():
                W, b = layer.folded()
                x = F.relu(F.linear(x, W, b))
            outs.append(x.max(dim=2)[0])                                            # (B,N,128)
        x = torch.cat(outs, dim=-1)
        return plinear(self, self.mlp3, x)


class FinePointMatching(nn.Module):
    """fine_point_matching.py:12-86 (inference branch)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.nblock = cfg.nblock
        self.in_proj = nn.Linear(cfg.input_dim, cfg.hidden_dim)
        self.out_proj = nn.Linear(cfg.hidden_dim, cfg.out_dim)
        self.bg_token = nn.Parameter(torch.randn(1, 1, cfg.hidden_dim) * .02)
        self.PE = PositionalEncoding(cfg.hidden_dim, r1=cfg.pe_radius1, r2=cfg.pe_radius2)
        self.transformers = nn.ModuleList([
            SparseToDenseTransformer(cfg.hidden_dim, focusing_factor=cfg.focusing_factor) for _ in range(self.nblock)])

    def forward(self, p1, f1, geo1, fps_idx1, p2, f2, geo2, fps_idx2, radius, end_points):
        B = p1.size(0)
        init_R, init_t = end_points["init_R"], end_points["init_t"]
        # (p1 - t) R spelled out per component: a batched 2048 x 3 x 3 library GEMM picks its kernel by the batch size, so the
        # last bits of a row followed the instances around it (round 5: group-vs-single identity of FramePipeline)
        d = p1 - init_t.unsqueeze(1)
        p1_ = d[..., 0:1] * init_R[:, None, 0, :] + d[..., 1:2] * init_R[:, None, 1, :] + d[..., 2:3] * init_R[:, None, 2, :]
        bg = self.bg_token.expand(B, -1, -1)
        # (bg token, dense rows) kept apart through the blocks: one concatenation per side at the end instead of one per block
        f1 = (bg, plinear(self, self.in_proj, f1, residual=self.PE(p1_)))
        f2 = (bg, plinear(self, self.in_proj, f2, residual=self.PE(p2)))
        for blk in self.transformers:
            f1, f2 = blk(f1, geo1, fps_idx1, f2, geo2, fps_idx2)
        o1 = torch.cat([plinear(self, self.out_proj, f1[0].contiguous()), plinear(self, self.out_proj, f1[1])], dim=1)
        o2 = torch.cat([plinear(self, self.out_proj, f2[0].contiguous()), plinear(self, self.out_proj, f2[1])], dim=1)
        model = end_points["model"] / (radius.reshape(-1, 1, 1) + 1e-6)
        if policy.guard("pem.FinePointMatching", cuda=o1.is_cuda, have=ops.have("fine_match"), f32=o1.dtype == torch.float32, C256=o1.shape[2] == 256):
            # similarity tiles are formed inside the assignment kernel: the (B,2049,2049) matrix is never written
            R, t, score = fine_Rt(None, p1, p2, model, feats=(o1, o2, self.cfg.temp))
        else:
            R, t, score = fine_Rt(feature_similarity(o1, o2, self.cfg.temp), p1, p2, model)
        end_points["pred_R"] = R
        end_points["pred_t"] = t * (radius.reshape(-1, 1) + 1e-6)
        end_points["pred_pose_score"] = score
        return end_points


class Net(nn.Module):
    """pose_estimation_model.py:11-53."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.coarse_npoint = cfg.coarse_npoint
        self.fine_npoint = cfg.fine_npoint
        self.feature_extraction = ViTEncoder(cfg.feature_extraction, self.fine_npoint)
        self.geo_embedding = GeometricStructureEmbedding(cfg.geo_embedding)
        self.coarse_point_matching = CoarsePointMatching(cfg.coarse_point_matching)
        self.fine_point_matching = FinePointMatching(cfg.fine_point_matching)

    def match(self, dense_pm, dense_fm, dense_po, dense_fo, radius, end_points):
        """Everything after feature extraction (pose_estimation_model.py:26-51)."""
        B = dense_pm.size(0)
        bg_point = torch.full((B, 1, 3), 100.0, device=dense_pm.device, dtype=dense_pm.dtype)
        sparse_pm, sparse_fm, fps_idx_m = sample_pts_feats(dense_pm, dense_fm, self.coarse_npoint)
        geo_m = self.geo_embedding(torch.cat([bg_point, sparse_pm], dim=1))
        sparse_po, sparse_fo, fps_idx_o = sample_pts_feats(dense_po, dense_fo, self.coarse_npoint)
        geo_o = self.geo_embedding(torch.cat([bg_point, sparse_po], dim=1))
        end_points = self.coarse_point_matching(sparse_pm, sparse_fm, geo_m, sparse_po, sparse_fo, geo_o,
                                                radius, end_points)
        return self.fine_point_matching(dense_pm, dense_fm, geo_m, fps_idx_m, dense_po, dense_fo, geo_o,
                                        fps_idx_o, radius, end_points)

    def forward(self, end_points):
        inputs = dict(end_points)
        dense_pm, dense_fm, dense_po, dense_fo, radius = self.feature_extraction(end_points)
        out = self.match(dense_pm, dense_fm, dense_po, dense_fo, radius, end_points)
        return self._f16_range_guard(inputs, out)

    _BATCHED = ("pts", "rgb", "rgb_choose", "model", "dense_po", "dense_fo", "coarse_rand_u")

    def _f16_range_guard(self, inputs, out):
        """The IEEE-half ViT-B (S6D_PEM_VIT_DTYPE=fp16) overflows at 65504; ``ViT_AE.tokens_up`` leaves a per-instance flag on the
        device.  It is read ONCE here, after the whole forward has been enqueued (one host wait at the end of the stage, where the
        caller reads the poses anyway); flagged instances are re-run with the fp32 extractor and their rows replaced, with a warning.
        Under hipGraph capture the flag is returned as ``out["f16_overflow"]`` for the graph's owner to act on (sam6d_amd.pipeline).
        S6D_PEM_F16_GUARD=0 turns the host read off (the flag is still returned)."""
        import os
        import warnings

        from .feature_extraction import force_vit_dtype
        bad = self.feature_extraction.rgb_net.overflow
        self.feature_extraction.rgb_net.overflow = None
        if bad is None:
            return out
        out["f16_overflow"] = bad
        if (bad.is_cuda and torch.cuda.is_current_stream_capturing()) or policy.current().pem_f16_guard == "0":
            return out
        if not bool(bad.any()):
            return out
        idx = torch.nonzero(bad).squeeze(1)
        warnings.warn(f"PEM ViT-B in IEEE half overflowed (|x| > 65504) for instance(s) {idx.tolist()} of {bad.numel()}: "
                      "re-running them with the fp32 extractor (set S6D_PEM_VIT_DTYPE=fp32 or bf16 for this checkpoint)",
                      RuntimeWarning, stacklevel=3)
        B = bad.numel()
        sub = {k: (v[idx].contiguous() if k in self._BATCHED and torch.is_tensor(v) and v.shape[:1] == (B,) else v)
               for k, v in inputs.items()}
        with force_vit_dtype("fp32"):
            redo = self.forward(sub)
        for k, v in redo.items():
            if torch.is_tensor(v) and torch.is_tensor(out.get(k)) and out[k].shape[:1] == (B,) and v.shape[:1] == (idx.numel(),) \
                    and k not in inputs:
                out[k] = out[k].clone()
                out[k][idx] = v.to(out[k].dtype)
        out["f16_overflow"] = bad
        return out


def default_cfg():
    """The ``model:`` node of Pose_Estimation_Model/config/base.yaml:16-55 (values restated)."""
    class AD(dict):
        __getattr__ = dict.__getitem__

    return AD(coarse_npoint=196, fine_npoint=2048,
              feature_extraction=AD(vit_type="vit_base", up_type="linear", embed_dim=768, out_dim=256,
                                    use_pyramid_feat=True, pretrained=False),
              geo_embedding=AD(sigma_d=0.2, sigma_a=15, angle_k=3, reduction_a="max", hidden_dim=256),
              coarse_point_matching=AD(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, temp=0.1,
                                       sim_type="cosine", normalize_feat=True, loss_dis_thres=0.15,
                                       nproposal1=6000, nproposal2=300),
              fine_point_matching=AD(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, pe_radius1=0.1,
                                     pe_radius2=0.2, focusing_factor=3, temp=0.1, sim_type="cosine",
                                     normalize_feat=True, loss_dis_thres=0.15))
