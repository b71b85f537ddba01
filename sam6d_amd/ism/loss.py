"""``model.loss`` classes of the Instance Segmentation Model on MI355X.

Hydra instantiates these by ``_target_`` (configs/model/ISM_sam.yaml:19-23:
``model.loss.PairwiseSimilarity``) and detector.py constructs
``MaskedPatch_MatrixSimilarity(metric="cosine", chunk_size=64)`` inline (:305,:312);
constructor arguments and method signatures are kept.

Re-derivations:
  * PairwiseSimilarity: one normalised GEMM (P,C)x(C,O*T) instead of a Python loop over objects
    on a (P,O,T,C) ``repeat`` (loss.py:30-40: 1.03 GB at P=200,O=30).
  * MaskedPatch_MatrixSimilarity: compute_straight and compute_visible_ratio are two reductions
    (row-max, column-max) of the SAME (S,256,256) similarity; ``both()`` computes them in one
    pass (the reference runs the batched GEMM twice, loss.py:54,66).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import policy


_FLOATS = (torch.float32, torch.float16, torch.bfloat16)


class PairwiseSimilarity(nn.Module):
    def __init__(self, metric="cosine", chunk_size=64):
        super().__init__()
        self.metric = metric
        self.chunk_size = chunk_size

    def forward(self, query, reference):
        """query (P,C), reference (O,T,C) -> (P,O,T) in [0,1]."""
        O, T, C = reference.shape
        if policy.guard("ism.PairwiseSimilarity", cuda=query.is_cuda, have=ops.have("pairwise_cosine"),
                        float_dtypes=query.dtype in _FLOATS and reference.dtype in _FLOATS):
            # fp16 / bf16 descriptors (the BOP flow runs under Lightning precision=16, configs/machine/trainer/local.yaml:9) take
            # the same kernel: it accumulates in fp32 either way, and under autocast the reference's cosine_similarity is an
            # fp32 op with an fp32 result -- which is what comes back here
            return ops.pairwise_cosine(query.float().contiguous(), reference.reshape(O * T, C).float().contiguous()).view(-1, O, T)
        q = F.normalize(query.float(), dim=-1)
        r = F.normalize(reference.float().reshape(O * T, C), dim=-1)
        return (q @ r.t()).clamp(min=0.0, max=1.0).view(-1, O, T).to(query.dtype)


class MaskedPatch_MatrixSimilarity(nn.Module):
    def __init__(self, metric="cosine", chunk_size=64):
        super().__init__()
        self.metric = metric
        self.chunk_size = chunk_size

    def both(self, query, reference, thred=0.5):
        """(appearance score (S), visible ratio (S)) from one similarity pass."""
        S, N2 = reference.shape[0], reference.shape[1]
        if policy.guard("ism.MaskedPatch_MatrixSimilarity", cuda=query.is_cuda, have=ops.have("patch_scores"),
                        float_dtypes=query.dtype in _FLOATS and reference.dtype in _FLOATS, patches_le_256=N2 <= 256,
                        C32=query.shape[-1] % 32 == 0):
            # the kernel addresses a resident (O,T,N2,C) store by (object, template): a materialised (S,N2,C) reference
            # is the store of ONE object whose "templates" are the S rows
            obj = torch.zeros(S, dtype=torch.int32, device=query.device)
            tmpl = torch.arange(S, dtype=torch.int32, device=query.device)
            appe, ratio = ops.patch_scores(query.float().contiguous(), reference.float().contiguous()[None], obj, tmpl, float(thred))
            return appe.to(query.dtype), ratio.to(query.dtype)       # half inputs: fp32 arithmetic inside, the caller's dtype outside
        sim = query @ reference.transpose(1, 2)
        factor = torch.count_nonzero(query.sum(dim=-1), dim=-1) + 1e-6
        appe = (sim.max(dim=-1).values.sum(dim=-1) / factor).clamp(min=0.0, max=1.0)
        col = sim.max(dim=1).values
        valid = torch.count_nonzero(col, dim=1) + 1e-6
        ratio = torch.count_nonzero(col * (col > thred), dim=1) / valid
        return appe, ratio

    def compute_straight(self, query, reference):
        return self.both(query, reference)[0]

    def compute_visible_ratio(self, query, reference, thred=0.5):
        return self.both(query, reference, thred)[1]
