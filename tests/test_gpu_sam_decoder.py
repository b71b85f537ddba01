"""GPU parity of the drop-in SAM prompt encoder + mask decoder vs the reference goldens."""
import numpy as np
import pytest
import torch

from oracle import sam_decoder as osd
from sam6d_amd.utils import seeded
from tests import util
from tests.test_host_sam_decoder import build, case, run

pytestmark = pytest.mark.gpu


def _cuda(inp):
    return {k: v.cuda() for k, v in inp.items()}


@pytest.mark.parametrize("force_lib", [False, True])
def test_mini_fp32_vs_reference_golden(monkeypatch, force_lib):
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "fp32")
    g, c, cfg, inp = case("mini")
    inp = _cuda(inp)
    m = seeded.load_seeded(build(cfg), c["weight_seed"]).cuda()
    with torch.no_grad():
        for tag, kw in (("", dict(points=(inp["points"], inp["labels"]))),
                        ("2", dict(points=(inp["points2"], inp["labels2"]), multi=False)),
                        ("_box", dict(boxes=inp["boxes"]))):
            s, mk, iou = run(m, inp["emb"], force_lib=force_lib, **kw)
            np.testing.assert_allclose(s.cpu().numpy(), g["mini_sparse" + tag], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(mk.cpu().numpy(), g["mini_masks" + tag], rtol=1e-3, atol=1e-4)
            np.testing.assert_allclose(iou.cpu().numpy(), g["mini_iou" + tag], rtol=1e-3, atol=1e-4)


def test_released_config_fp32_and_bf16_vs_reference_golden(monkeypatch):
    g, c, cfg, inp = case("sam")
    inp = _cuda(inp)
    m = seeded.load_seeded(build(cfg), c["weight_seed"]).cuda()
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "fp32")
    with torch.no_grad():
        _, mk, iou = run(m, inp["emb"], points=(inp["points"], inp["labels"]))
    np.testing.assert_allclose(iou.cpu().numpy(), g["sam_iou"], rtol=1e-3, atol=1e-4)
    util.assert_digest_close(mk, g["sam_masks_sum"], g["sam_masks_smp"], 211, 1e-3, 1e-4, "low-res logits fp32")
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "bf16")
    with torch.no_grad():
        _, mk16, iou16 = run(m, inp["emb"], points=(inp["points"], inp["labels"]))
    smp = mk16.float().cpu().reshape(-1)[::211].numpy()
    assert np.corrcoef(smp, g["sam_masks_smp"])[0, 1] > 0.999, np.corrcoef(smp, g["sam_masks_smp"])[0, 1]
    assert np.abs(smp - g["sam_masks_smp"]).mean() < 0.02 * np.abs(g["sam_masks_smp"]).mean() + 1e-3
    assert np.abs(iou16.cpu().numpy() - g["sam_iou"]).max() < 2e-2


def test_postprocess_matches_oracle():
    g, c, cfg, _ = case("mini")
    x = torch.from_numpy(g["mini_masks"][:3]).cuda()
    y = osd.postprocess_masks(x, cfg["img"], c["mini_input_size"], c["mini_orig"])      # torch ops on the device
    np.testing.assert_allclose(y.cpu().numpy(), g["mini_post"], rtol=1e-5, atol=1e-6)


def test_img2tok_kernel_vs_restated_algebra():
    """s6d_samdec_img2tok_bf16 against the same computation in torch fp32 on the same bf16 operands, shared and
    per-prompt q / residual, strided q, 5..8 prompt tokens."""
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(1)
    B, N = 5, 256
    for T, shared in ((7, True), (8, False), (5, False)):
        Bq = 1 if shared else B
        qw = (torch.randn(Bq, N, 384, generator=g) * 0.5).cuda().to(torch.bfloat16)
        q = qw[..., 256:]                                               # last-dim slice: row stride 384
        q_add = None if shared else (torch.randn(N, 128, generator=g) * 0.5).cuda().to(torch.bfloat16)
        kexp = torch.zeros(B, 8, 8, 8, 16)
        kt = torch.randn(B, 8, T, 16, generator=g) * 0.5
        for hh in range(8):
            kexp[:, hh, :T, hh] = kt[:, hh]
        kexp = kexp.reshape(B, 64, 128).cuda().to(torch.bfloat16)
        vpt = torch.zeros(B, 256, 8, 8)
        vpt[..., :T] = torch.randn(B, 256, 8, T, generator=g)
        vpt = vpt.reshape(B, 256, 64).cuda().to(torch.bfloat16)
        resid = torch.randn(Bq, N, 256, generator=g).cuda().to(torch.bfloat16)
        bo, lw, lb = (torch.randn(256, generator=g).cuda() for _ in range(3))
        out = ops.samdec_img2tok(q, q_add, kexp, vpt, resid, bo, lw, lb, 1e-5, T).float()
        qq = q.float() if q_add is None else (q.float() + q_add.float()).to(torch.bfloat16).float()
        # comparand: the oracle's attention core on the UN-expanded keys (B, T, 128) and the values with the output projection
        # folded in (the kernel's algebra: out_proj(softmax(qk^T) v) = softmax(qk^T)(v W_o^T)), + bias + residual + LayerNorm.
        # The kernel rounds the probabilities to bf16 before the value product; so does the comparand.
        kk = kt.permute(0, 2, 1, 3).reshape(B, T, 128).to(torch.bfloat16).float()                        # (B, T, heads * 16)
        s = torch.einsum("bnk,bjk->bnj", qq.cpu().expand(B, -1, -1), kexp.float().cpu()).view(B, N, 8, 8)[..., :T]
        s_or = (qq.cpu().expand(B, -1, -1).reshape(B, N, 8, 16).transpose(1, 2) @ kk.reshape(B, T, 8, 16).permute(0, 2, 3, 1))
        assert torch.allclose(s.permute(0, 2, 1, 3), s_or, atol=1e-4)                                     # the key expansion is exact
        p = torch.softmax(s_or, -1).to(torch.bfloat16).float()                                           # (B, heads, N, T)
        vv = vpt.float().cpu().view(B, 256, 8, 8)[..., :T]                                               # (B, c, head, T)
        y = torch.einsum("bhnt,bcht->bnc", p, vv) + bo.cpu() + resid.float().cpu()
        ref = torch.nn.functional.layer_norm(y, (256,), lw.cpu(), lb.cpu(), 1e-5)
        out = out.cpu()
        err = (out - ref).abs()
        assert err.max() < 0.06 and err.mean() < 4e-3, (T, shared, err.max().item(), err.mean().item())


def test_img2tok_raw_kernel_vs_explicit_q_projection():
    """s6d_samdec_img2tok_raw_bf16 (round 4: q projection folded into the expanded keys, raw image tokens + positional encoding read
    in place) against the statement with the EXPLICIT projection q = W_q (x + pe) + b_q in float32: attention of every image token
    over the T prompt tokens per head, out_proj folded into the values, + bias + residual + LayerNorm.  Shared and per-prompt x."""
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(5)
    B, N = 4, 256
    wq, bq = torch.randn(128, 256, generator=g) / 16, 0.5 * torch.randn(128, generator=g)
    for T, shared, use_pe in ((7, False, True), (8, True, True), (5, False, False)):
        Bx = 1 if shared else B
        x = torch.randn(Bx, N, 256, generator=g).to(torch.bfloat16)
        pe = torch.randn(N, 256, generator=g).to(torch.bfloat16) if use_pe else None
        kt = torch.randn(B, 8, T, 16, generator=g) * 0.5                               # scaled keys per (head, token)
        kexp = torch.zeros(B, 8, 8, 8, 16)
        for hh in range(8):
            kexp[:, hh, :T, hh] = kt[:, hh]
        kexp = kexp.reshape(B, 64, 128)
        k256, cb = (kexp @ wq).to(torch.bfloat16), (kexp @ bq).contiguous()
        vpt = torch.zeros(B, 256, 8, 8)
        vpt[..., :T] = torch.randn(B, 256, 8, T, generator=g)
        vpt = vpt.reshape(B, 256, 64).to(torch.bfloat16)
        bo, lw, lb = (torch.randn(256, generator=g) for _ in range(3))
        out = ops.samdec_img2tok_raw(x.cuda(), pe.cuda() if use_pe else None, k256.cuda(), cb.cuda(), vpt.cuda(), x.cuda(), bo.cuda(),
                                     lw.cuda(), lb.cuda(), 1e-5, T).float().cpu()
        xf = x.float().expand(B, -1, -1)
        q = ((xf + pe.float()) if use_pe else xf) @ wq.t() + bq                           # (B, N, 128)
        s_ = q.reshape(B, N, 8, 16).transpose(1, 2) @ kt.transpose(-1, -2)               # (B, heads, N, T)
        p = torch.softmax(s_, -1)
        vv = vpt.float().view(B, 256, 8, 8)[..., :T]
        y = torch.einsum("bhnt,bcht->bnc", p, vv) + bo + xf
        ref = torch.nn.functional.layer_norm(y, (256,), lw, lb, 1e-5)
        err = (out - ref).abs()
        # measured 0.039 / 2.5e-3: the bf16 output grid at |x| <= 4 (0.016) + bf16 probabilities and folded keys
        assert err.max() < 0.08 and err.mean() < 4e-3, (T, shared, use_pe, err.max().item(), err.mean().item())


def test_upscale_heads_kernel_vs_restated_algebra():
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(2)
    B, h, w, M = 3, 8, 8, 4
    yw = torch.randn(B, h * w, 512, generator=g).cuda().to(torch.bfloat16)
    y0 = yw[..., 256:]
    lw, lb = (1 + 0.1 * torch.randn(64, generator=g)).cuda(), (0.1 * torch.randn(64, generator=g)).cuda()
    w2t = (torch.randn(128, 64, generator=g) / 8).cuda().to(torch.bfloat16)
    b2 = (0.1 * torch.randn(32, generator=g)).cuda()
    hyper = torch.randn(B, M, 32, generator=g).cuda()
    masks = ops.samdec_upscale_heads(y0, lw, lb, 1e-6, w2t, b2, hyper, h, w)
    x = y0.float().view(B, h, w, 2, 2, 64)                              # (b, y, x, dy, dx, c)
    u = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x, (64,), lw, lb, 1e-6)).to(torch.bfloat16).float()
    v = torch.nn.functional.gelu(u @ w2t.float().t() + b2.repeat(4))    # (..., (dy2, dx2, ch))
    v = v.view(B, h, w, 2, 2, 2, 2, 32)
    lg = torch.einsum("byxijklc,bmc->bmyikxjl", v, hyper).reshape(B, M, 4 * h, 4 * w)
    err = (masks - lg).abs()
    assert err.max() < 5e-3 * lg.abs().max() + 1e-3, (err.max().item(), lg.abs().max().item())


def test_tok2img_kernel_vs_oracle():
    """s6d_samdec_tok2img_f32 (token -> image attention, 8 heads x 16, keys / values read in place from the concatenated
    [q | k | v] projection of the image tokens) against the oracle's attention core (oracle/sam_decoder.py attention_core, the
    statements between the projections of the reference Attention), fp32 on the same bf16-rounded operands."""
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(4)
    B, N, T = 6, 512, 7
    qt = torch.randn(B, T, 128, generator=g).cuda()
    for shared, pe in ((False, True), (True, False)):
        kv = torch.randn(1 if shared else B, N, 384, generator=g).cuda().to(torch.bfloat16)
        kpe = torch.randn(N, 128, generator=g).cuda().to(torch.bfloat16) if pe else None
        out = ops.samdec_tok2img(qt, kv, 128, 256, kpe, 0.25)
        k = kv[..., 128:256].float()
        if pe:
            k = (k + kpe.float()).to(torch.bfloat16).float()
        v = kv[..., 256:384].float()
        ref = osd.attention_core(qt.cpu(), k.cpu().expand(B, -1, -1), v.cpu().expand(B, -1, -1), 8)    # 1 / sqrt(16) = 0.25
        assert (out.cpu() - ref).abs().max() < 2e-4, (shared, (out.cpu() - ref).abs().max().item())


def test_tok2img_raw_kernel_vs_oracle_attention_with_explicit_projections():
    """s6d_samdec_tok2img_raw_bf16 (round 4: k / v projections folded into the queries, raw image tokens attended on the matrix
    cores) against the oracle's attention core fed with EXPLICIT projections k = W_k (x + pe) + b_k, v = W_v x + b_v in float32 on
    the same bf16 image tokens.  Tolerance: the folded queries and x + pe are rounded to bf16 (2^-9 relative) before a 256-term dot
    product with O(1) scores, and P is rounded to bf16 before the value product."""
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(7)
    B, N, T = 5, 512, 7
    qt = torch.randn(B, T, 128, generator=g)
    wk, wv = torch.randn(128, 256, generator=g) / 16, torch.randn(128, 256, generator=g) / 16
    bk, bv = torch.randn(128, generator=g), torch.randn(128, generator=g)
    for shared, use_pe in ((False, True), (True, False), (False, False)):
        x = torch.randn(1 if shared else B, N, 256, generator=g).to(torch.bfloat16)
        pe = torch.randn(N, 256, generator=g).to(torch.bfloat16) if use_pe else None
        out = ops.samdec_tok2img_raw(qt.cuda(), x.cuda(), pe.cuda() if use_pe else None, wk.cuda(), wv.cuda(), bv.cuda(), 0.25).cpu()
        xf = x.float().expand(B, -1, -1)
        k = (xf + pe.float() if use_pe else xf) @ wk.t() + bk
        v = xf @ wv.t() + bv
        ref = osd.attention_core(qt, k, v, 8)
        err = (out - ref).abs().max().item()
        assert out.shape == (B, T, 128) and err < 2e-2 and (out - ref).abs().mean().item() < 1e-3, (shared, use_pe, err)   # 7e-3 / 3.4e-4
    with pytest.raises(Exception):
        ops.samdec_tok2img_raw(qt.cuda(), torch.zeros(B, 100, 256, dtype=torch.bfloat16).cuda(), None, wk.cuda(), wv.cuda(), bv.cuda(), 0.25)


def test_mask_post_kernel_bit_exact_vs_oracle_and_golden():
    from sam6d_amd import ops
    g, c, _, _ = case("mini")
    from sam6d_amd.utils import synth
    low = synth.sam_lowres_logits(c["post_B"], 3, 256, c["post_seed"])
    (ih, iw), (H, W) = c["post_input_size"], c["post_orig"]
    mb, st, boxes = ops.sam_mask_post(low.cuda(), 1024, (ih, iw), (H, W), 0.0, 1.0)
    rb, rs, rbox = osd.mask_postprocess(low, 1024, (ih, iw), (H, W))
    assert torch.equal(mb.cpu(), rb)
    np.testing.assert_array_equal(st.cpu().numpy(), rs.numpy())
    np.testing.assert_array_equal(boxes.cpu().numpy(), rbox.numpy())
    np.testing.assert_array_equal(boxes.cpu().numpy(), g["post_boxes"])
    np.testing.assert_array_equal(st.cpu().numpy(), g["post_stability"])
    # odd sizes: non-multiple-of-4 rows, a frame larger than the padded square's valid region, n = 64
    low2 = synth.sam_lowres_logits(2, 2, 64, 3)
    mb, st, boxes = ops.sam_mask_post(low2.cuda(), 256, (171, 256), (333, 499), 0.0, 1.0)
    rb, rs, rbox = osd.mask_postprocess(low2, 256, (171, 256), (333, 499))
    assert torch.equal(mb.cpu(), rb) and torch.equal(boxes.cpu(), rbox)
    np.testing.assert_array_equal(st.cpu().numpy(), rs.numpy())
    e = ops.sam_mask_post(low2[:0].cuda(), 256, (171, 256), (33, 49))
    assert e[0].shape == (0, 33, 49) and e[1].shape == (0,) and e[2].shape == (0, 4)
    # a frame much smaller than the resized input (3.2 intermediate rows per frame row: the strip of a workgroup shortens)
    low3 = synth.sam_lowres_logits(2, 3, 128, 5)
    mb, st, boxes = ops.sam_mask_post(low3.cuda(), 512, (384, 512), (120, 160), 0.0, 1.0)
    rb, rs, rbox = osd.mask_postprocess(low3, 512, (384, 512), (120, 160))
    assert torch.equal(mb.cpu(), rb) and torch.equal(boxes.cpu(), rbox)
    np.testing.assert_array_equal(st.cpu().numpy(), rs.numpy())
    mb, st, boxes = ops.sam_mask_post(low3.cuda(), 512, (384, 512), (3, 5), 0.0, 1.0)          # 128 intermediate rows per frame row
    rb, rs, rbox = osd.mask_postprocess(low3, 512, (384, 512), (3, 5))
    assert torch.equal(mb.cpu(), rb) and torch.equal(boxes.cpu(), rbox)
    # a channel slice of the decoder's output ((B, 4, n, n)[:, 1:], what process_point_batch passes) is read in place
    full = synth.sam_lowres_logits(3, 4, 64, 9).cuda()
    sl = full[:, 1:]
    assert not sl.is_contiguous()
    mb, st, boxes = ops.sam_mask_post(sl, 256, (171, 256), (120, 160), 0.0, 1.0)
    rb, rs, rbox = osd.mask_postprocess(sl.cpu().contiguous(), 256, (171, 256), (120, 160))
    assert mb.dtype == torch.bool and torch.equal(mb.cpu(), rb) and torch.equal(boxes.cpu(), rbox)
    np.testing.assert_array_equal(st.cpu().numpy(), rs.numpy())


def test_process_point_batch_matches_reference_sequence(monkeypatch):
    """Generator batch body: decoder -> fused post-processing -> both filters, against the reference's sequence
    (oracle post-processing + amg-style filtering) applied to the same low-resolution logits."""
    from sam6d_amd.sam import amg
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "bf16")
    g, c, cfg, inp = case("sam")
    inp = _cuda(inp)
    m = seeded.load_seeded(build(cfg), c["weight_seed"]).cuda()
    pts = torch.rand(24, 2, generator=torch.Generator().manual_seed(0)).cuda() * torch.tensor([1024.0, 768.0]).cuda()
    off = 0.02                                   # seeded weights give |logit| < 0.3: a +-1 band would swallow every mask
    every = amg.process_point_batch(m.prompt_encoder, m.mask_decoder, inp["emb"], pts, (768, 1024), (480, 640),
                                    pred_iou_thresh=0.0, stability_score_thresh=0.0, stability_score_offset=off)
    assert every["masks"].shape == (72, 480, 640)
    t_iou = every["iou_preds"].float().median().item()
    t_st = every["stability_score"][torch.isfinite(every["stability_score"])].median().item()
    out = amg.process_point_batch(m.prompt_encoder, m.mask_decoder, inp["emb"], pts, (768, 1024), (480, 640),
                                  pred_iou_thresh=t_iou, stability_score_thresh=t_st, stability_score_offset=off)
    low = out["low_res_logits"].float().cpu()
    rb, rs, rbox = osd.mask_postprocess(low, 1024, (768, 1024), (480, 640), 0.0, off)
    keep = (every["iou_preds"].float().cpu() > t_iou) & (rs >= t_st)
    assert keep.any() and not keep.all()
    idx = torch.nonzero(keep).squeeze(1)
    assert torch.equal(out["point_index"].cpu(), idx // 3)
    assert torch.equal(out["masks"].cpu(), rb[idx]) and torch.equal(out["boxes"].cpu(), rbox[idx])
    np.testing.assert_array_equal(out["stability_score"].cpu().numpy(), rs[idx].numpy())


def test_nms_kernel_vs_torchvision_algorithm():
    from sam6d_amd import ops
    g = torch.Generator().manual_seed(7)
    for N in (1, 63, 64, 200, 1500):
        xy = torch.rand(N, 2, generator=g) * 500
        wh = 10 + torch.rand(N, 2, generator=g) * 120
        boxes = torch.cat([xy, xy + wh], 1).round()                    # integer-valued like mask boxes (exact IoU ties)
        scores = torch.rand(N, generator=g)
        keep = ops.nms(boxes.cuda(), scores.cuda(), 0.7).cpu()
        assert torch.equal(keep, osd.nms(boxes, scores, 0.7)), N
    assert ops.nms(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), 0.7).shape == (0,)
    dup = torch.tensor([[0.0, 0, 10, 10]] * 5).cuda()                  # identical boxes: only the best survives
    assert ops.nms(dup, torch.tensor([0.1, 0.9, 0.5, 0.9, 0.2]).cuda(), 0.7).tolist() == [1]


def test_generate_proposals_pipeline(monkeypatch):
    """Embedding -> proposals on the device: every output mask is one of the batch body's masks, the kept set is what
    the torchvision-style NMS keeps among the filtered ones, boxes match the masks."""
    from sam6d_amd.sam import amg
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "bf16")
    g, c, cfg, inp = case("sam")
    inp = _cuda(inp)
    m = seeded.load_seeded(build(cfg), c["weight_seed"]).cuda()
    kw = dict(pred_iou_thresh=0.08, stability_score_thresh=0.3, stability_score_offset=0.02, box_nms_thresh=0.7)
    out = amg.generate_proposals(m.prompt_encoder, m.mask_decoder, inp["emb"], (480, 640), points_per_side=8,
                                 points_per_batch=24, **kw)
    K = out["masks"].shape[0]
    assert 0 < K < 8 * 8 * 3 and out["masks"].shape[1:] == (480, 640) and out["boxes"].shape == (K, 4)
    assert torch.equal(out["boxes"].cpu(), osd.mask_to_box(out["masks"].cpu()))
    assert (out["iou_preds"][:-1] >= out["iou_preds"][1:]).all()                   # NMS returns by decreasing score
    # recompute the candidate set batch by batch and check the NMS decision against the oracle's
    pts = torch.as_tensor(amg.build_point_grid(8) * [[640, 480]] * [[1024 / 640, 768 / 480]], device="cuda")
    cand = [amg.process_point_batch(m.prompt_encoder, m.mask_decoder, inp["emb"], pts[a:a + 24], (768, 1024), (480, 640),
                                    pred_iou_thresh=0.08, stability_score_thresh=0.3, stability_score_offset=0.02)
            for a in range(0, 64, 24)]
    boxes = torch.cat([r["boxes"] for r in cand]).float().cpu()
    scores = torch.cat([r["iou_preds"] for r in cand]).float().cpu()
    keep = osd.nms(boxes, scores, 0.7)
    assert torch.equal(out["boxes"].cpu().float(), boxes[keep])


def test_decoder_batch_graph_replay_equals_eager(monkeypatch):
    """process_point_batch replays the prompt-encoder + mask-decoder + post-processing batch as a hipGraph (round 4): the replay
    gives the eager path's bits, a second frame's embedding goes through the same graph, and a weight change re-captures."""
    from sam6d_amd.sam import amg
    monkeypatch.setenv("S6D_SAM_DECODER_DTYPE", "bf16")
    g, c, cfg, inp = case("sam")
    inp = _cuda(inp)
    m = seeded.load_seeded(build(cfg), c["weight_seed"]).cuda()
    pts = torch.rand(128, 2, generator=torch.Generator().manual_seed(0)).cuda() * torch.tensor([1024.0, 768.0]).cuda()
    kw = dict(pred_iou_thresh=0.0, stability_score_thresh=0.0, stability_score_offset=0.02)
    emb2 = inp["emb"] * 0.5 + 0.1

    def run(emb):
        r = amg.process_point_batch(m.prompt_encoder, m.mask_decoder, emb, pts, (768, 1024), (480, 640), **kw)
        return {k: r[k].clone() for k in ("masks", "iou_preds", "stability_score", "boxes", "low_res_logits")}
    monkeypatch.setenv("S6D_AMG_GRAPH", "0")
    amg.invalidate_graphs()
    e1, e2 = run(inp["emb"]), run(emb2)
    assert not torch.equal(e1["iou_preds"], e2["iou_preds"])
    monkeypatch.setenv("S6D_AMG_GRAPH", "1")
    g1 = run(inp["emb"])
    assert len(amg._GRAPHS) == 1
    g2, g1b = run(emb2), run(inp["emb"])
    assert len(amg._GRAPHS) == 1                                      # one capture served the three calls
    for k in e1:
        assert torch.equal(g1[k], e1[k]) and torch.equal(g2[k], e2[k]) and torch.equal(g1b[k], e1[k]), k
    with torch.no_grad():
        m.mask_decoder.iou_prediction_head.layers[-1].bias.add_(0.25)   # an in-place weight update: the stale capture must not be replayed
    g3 = run(inp["emb"])
    assert torch.allclose(g3["iou_preds"].float(), e1["iou_preds"].float() + 0.25, atol=2e-2) and len(amg._GRAPHS) == 2
    amg.invalidate_graphs()


def test_cast_cached_linear_is_autocast_linear_bit_for_bit_and_follows_weight_updates():
    """utils.linear.CastCachedLinear (the decoder's small Linear layers): under autocast its output equals nn.Linear's bit for bit (same
    cast operands, same library product), the cast copies are reused across calls, and an in-place weight update or a
    load_state_dict is picked up (the cache is keyed on the parameters' versions and storage)."""
    from sam6d_amd.utils.linear import CastCachedLinear
    g = torch.Generator().manual_seed(0)
    ref = torch.nn.Linear(256, 128).cuda()
    lin = CastCachedLinear(256, 128).cuda()
    lin.load_state_dict(ref.state_dict())
    x = torch.randn(1024, 7, 256, generator=g).cuda()
    with torch.no_grad():
        assert torch.equal(lin(x), ref(x))                                   # no autocast: nn.Linear.forward
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            a, b = lin(x), ref(x)
            assert a.dtype == torch.bfloat16 and torch.equal(a, b)
            w0 = lin.__dict__["_s6d_cast"][1]
            lin(x)
            assert lin.__dict__["_s6d_cast"][1] is w0                        # reused
        # (autocast's own cast cache is per region and is NOT invalidated by an in-place update inside the region: nn.Linear is
        # compared in a fresh region after every change; CastCachedLinear follows the parameter's version in any region)
        ref.weight.mul_(1.5)
        lin.weight.mul_(1.5)                                                 # in-place update: version changes
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            assert torch.equal(lin(x), ref(x)) and lin.__dict__["_s6d_cast"][1] is not w0
        sd = {k: v * 0.5 for k, v in ref.state_dict().items()}
        ref.load_state_dict(sd)
        lin.load_state_dict(sd)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            assert torch.equal(lin(x), ref(x))
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            assert torch.equal(lin(x), ref(x))                               # another autocast dtype: its own copies
    y = lin(x)                                                               # grad enabled: nn.Linear.forward, differentiable
    assert y.requires_grad


def _bf(x):
    return x.to(torch.bfloat16).float()


def _autocast_linear(x, m):
    """nn.Linear under bf16 autocast, spelled out in fp32: bf16-rounded operands, fp32 accumulation, bf16-rounded result."""
    return _bf(_bf(x) @ _bf(m.weight.detach().float()).t() + m.bias.detach().float())


def _token_side_statement(L, queries, pe, t2i):
    """TwoWayAttentionBlock steps 1-3 for the sparse tokens + the image->token k / v projections (transformer.py:109-186) with the
    autocast roundings written out -- the arithmetic csrc/s6d_samtok.hip states in its header."""
    import math
    import torch.nn.functional as F
    sa, ca, ci = L.self_attn, L.cross_attn_token_to_image, L.cross_attn_image_to_token
    x = queries if L.skip_first_layer_pe else queries + pe
    B, T, _ = x.shape
    H = sa.num_heads
    q, k, v = (_autocast_linear(a, m).view(B, T, H, -1).transpose(1, 2) for a, m in ((x, sa.q_proj), (x, sa.k_proj), (queries, sa.v_proj)))
    s = _bf(q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    o = _bf(_bf(torch.softmax(s, dim=-1)) @ v).transpose(1, 2).reshape(B, T, -1)
    a = _autocast_linear(o, sa.out_proj)
    q1 = F.layer_norm(a if L.skip_first_layer_pe else queries + a, (256,), L.norm1.weight, L.norm1.bias, L.norm1.eps)
    qp = _autocast_linear(q1 + pe, ca.q_proj)
    att = t2i(qp)
    q2 = F.layer_norm(q1 + _autocast_linear(att, ca.out_proj), (256,), L.norm2.weight, L.norm2.bias, L.norm2.eps)
    h = torch.relu(_autocast_linear(q2, L.mlp.lin1))
    q3 = F.layer_norm(q2 + _autocast_linear(h, L.mlp.lin2), (256,), L.norm3.weight, L.norm3.bias, L.norm3.eps)
    return q1, qp, q3, _autocast_linear(q3 + pe, ci.k_proj), _autocast_linear(q3, ci.v_proj)


@pytest.mark.parametrize("B,T", [(6, 7), (1, 5), (9, 8)])
def test_token_side_kernels_vs_autocast_statement(B, T):
    """s6d_samdec_tokens_pre_bf16 / _post_bf16 (round 6: the sparse-token side of a TwoWayAttentionBlock in two launches) against the
    same layer written with library ops and the autocast roundings spelled out, both layers (with and without skip_first_layer_pe),
    prompt counts that do not fill a workgroup, T = 5 .. 8 tokens.  The token->image attention between the two kernels is a stand-in
    (a fixed random linear map of the projected queries): the kernels' own arithmetic is what is under test.  Tolerance: a value that
    sits on a bf16 rounding boundary may round the other way after a differently ordered fp32 sum (one bf16 ulp = 2^-8 relative on one
    element of a 256-term dot product); everything else agrees to fp32 accumulation order.  Measured: max 7.8e-3, mean <= 1.0e-4
    (one prompt x five tokens, where a single flipped value weighs most); bounds 4e-2 / 5e-4."""
    from sam6d_amd import ops, policy
    from sam6d_amd.sam.mask_decoder import build_sam_decoder
    m = seeded.load_seeded(build_sam_decoder(), 3).cuda()
    dec = m.mask_decoder
    g = torch.Generator().manual_seed(B * 10 + T)
    queries = torch.randn(B, T, 256, generator=g).cuda()
    pe = torch.randn(B, T, 256, generator=g).cuda()
    mix = (torch.randn(128, 128, generator=g) / 11.0).cuda()

    def t2i(qp):
        return torch.tanh(qp.float() @ mix)
    for li in (0, 1):
        L = dec.transformer.layers[li]
        assert policy.guard("test", have=ops.have("samdec_tokens"))
        with torch.no_grad():
            want = _token_side_statement(L, queries, pe, t2i)
            seen = {}

            def spy(qp):
                seen["qp"] = qp.clone()
                return t2i(qp)
            q3, (kt, vt) = dec._token_side(li, queries, pe, spy)
            lw, nw, _ = dec._token_weights(li)
            q1, _ = ops.samdec_tokens_pre(queries, pe, not L.skip_first_layer_pe, lw[0], lw[1], lw[2], lw[3], nw[0], lw[4])
        for name, a, b in (("q1", q1, want[0]), ("qp", seen["qp"], want[1]), ("q3", q3, want[2]), ("kt", kt, want[3]), ("vt", vt, want[4])):
            err = (a - b).abs()
            util.record_margin(f"samdec_token_side_L{li}_B{B}_T{T}_{name}", max_abs=err.max().item(), mean_abs=err.mean().item(), ref_abs_max=b.abs().max().item())
            assert err.max().item() < 4e-2 and err.mean().item() < 5e-4, (li, name, err.max().item(), err.mean().item())


@pytest.mark.parametrize("B,T", [(5, 7), (2, 5)])
def test_token_side_kernels_with_the_folds_inside(B, T):
    """The same two kernels with the per-head products around the attention cores made INSIDE them (round 6): the token->image
    attention's k projection folded into the queries (first kernel), its W_v product, and the operands of the image->token attention
    kernels -- block-diagonal scaled keys or their W_q fold + bias term, values with out_proj folded in -- (second kernel), against
    the library glue they replace (ops.samdec_tok2img_raw's einsums in float32, MaskDecoder._expand) on the same image tokens.
    Differences: the folds multiply by the bf16-rounded weights (as the autocast Linear of the reference would) where the glue used
    float32 weights, so the attention output moves by ~2^-9 relative before it is rounded to bf16 for the out projection anyway."""
    import math
    from sam6d_amd import ops
    from sam6d_amd.sam.mask_decoder import build_sam_decoder
    dec = seeded.load_seeded(build_sam_decoder(), 3).cuda().mask_decoder
    g = torch.Generator().manual_seed(B * 100 + T)
    queries = torch.randn(B, T, 256, generator=g).cuda()
    pe = torch.randn(B, T, 256, generator=g).cuda()
    N = 256
    x = torch.randn(B, N, 256, generator=g).cuda().to(torch.bfloat16)
    pe_bf = torch.randn(N, 256, generator=g).cuda().to(torch.bfloat16)
    for li, fold_q in ((0, False), (1, True)):
        L = dec.transformer.layers[li]
        ca, ci = L.cross_attn_token_to_image, L.cross_attn_image_to_token
        sc = 1.0 / math.sqrt(ca.internal_dim // ca.num_heads)

        def t2i(qp):
            return ops.samdec_tok2img_raw(qp.float(), x, pe_bf, ca.k_proj.weight, ca.v_proj.weight, ca.v_proj.bias, sc)
        with torch.no_grad():
            q3_w, ktvt = dec._token_side(li, queries, pe, t2i)                     # glue between the kernels
            want = dec._expand(ci, q3_w, pe, fold_q=fold_q, ktvt=ktvt)
            q3, got = dec._token_side(li, queries, pe, None, x=x, pe_bf=pe_bf, fold_q=fold_q)
        assert isinstance(got, dict)
        pairs = [("q3", q3, q3_w)]
        if fold_q:
            pairs += [("k256", got["k256"].float(), want[0].float()), ("cb", got["cb"], want[1]), ("vpt", got["vpt"].float(), want[2].float())]
        else:
            pairs += [("kexp", got["kexp"].float(), want[0].float()), ("vpt", got["vpt"].float(), want[1].float())]
        for name, a, b in pairs:
            assert a.shape == b.shape, (name, a.shape, b.shape)
            err = (a - b).abs()
            util.record_margin(f"samdec_token_side_folds_L{li}_B{B}_T{T}_{name}", max_abs=err.max().item(), mean_abs=err.mean().item(), ref_abs_max=b.abs().max().item())
            assert err.max().item() < 8e-2 and err.mean().item() < 3e-3, (li, name, err.max().item(), err.mean().item())
        if not fold_q:
            assert torch.equal(got["kexp"].float() == 0, want[0].float() == 0)      # the block structure and the zero slots
