"""The PEM sampler kernel (csrc/s6d_pempre.hip) on the emulator against the library formulation it replaces (the top-k over
64-bit composite keys in sam6d_amd/pem/preprocess.py): identical indices, ties resolved by position, duplicated keys flagged."""
import numpy as np
import torch

from sam6d_amd.pem import preprocess as pre


def _lib_path(n, keys, n_sample, monkeypatch):
    monkeypatch.setenv("S6D_PEM_SAMPLER", "library")                 # the top-k formulation (the kernel is the default on the device)
    try:
        return pre._keyed_indices(n, keys, n_sample)
    finally:
        monkeypatch.delenv("S6D_PEM_SAMPLER")


def test_sampler_kernel_equals_the_library_path(emu, monkeypatch):
    g = torch.Generator().manual_seed(0)
    L, ns = 40000, 512
    n = torch.tensor([0, 1, 100, 512, 513, 600, 1024, 5000, 40000, 2049])
    keys = torch.rand(len(n), L, generator=g)
    keys[5] = (keys[5] * 300).floor() / 300                          # ties among 600 keys: the position decides
    keys[7, :5000] = (keys[7, :5000] * 1e4).floor() / 1e4
    want = _lib_path(n, keys, ns, monkeypatch)
    idx, overflow = emu.pem_sample_indices(keys, n, ns)
    assert overflow.tolist() == [0] * len(n)
    assert torch.equal(idx, want)
    # the selected positions are what a stable argsort gives
    row = 8
    np.testing.assert_array_equal(idx[row].numpy(), np.argsort(keys[row].numpy(), kind="stable")[:ns])
    # through observed_inputs' switch as well
    monkeypatch.setenv("S6D_PEM_SAMPLER", "kernel")
    assert torch.equal(pre._keyed_indices(n, keys, ns), want)


def test_heavily_duplicated_keys_are_flagged_and_fall_back(emu, monkeypatch):
    g = torch.Generator().manual_seed(1)
    n = torch.tensor([30000, 30000])
    keys = torch.rand(2, 30000, generator=g)
    keys[1] = (keys[1] * 4).floor() / 4                              # four distinct values: 7500 keys share the smallest
    want = _lib_path(n, keys, 2048, monkeypatch)
    idx, overflow = emu.pem_sample_indices(keys, n, 2048)
    assert overflow.tolist() == [0, 1] and torch.equal(idx[0], want[0])
    monkeypatch.setenv("S6D_PEM_SAMPLER", "kernel")
    assert torch.equal(pre._keyed_indices(n, keys, 2048), want)     # the switch falls back to the library path for the frame


def test_full_frame_with_the_kernel_sampler_matches_the_oracle(emu, monkeypatch):
    from oracle import pem_pre as opre
    from sam6d_amd.utils import synth
    monkeypatch.setenv("S6D_PEM_SAMPLER", "kernel")
    inp = synth.pem_pre_inputs(P=8, seed=3)
    kw = dict(radius=0.12, n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(),
                                keys=inp["keys"].numpy(), **kw)
    out = pre.observed_inputs(torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"], keys=inp["keys"], **kw)
    assert out["kept"].tolist() == ref["kept"].tolist()
    np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])
    np.testing.assert_array_equal(out["rgb_choose"].numpy(), ref["rgb_choose"])


def _frame(P=8, seed=3):
    from sam6d_amd.utils import synth
    inp = synth.pem_pre_inputs(P=P, seed=seed)
    return inp, (torch.from_numpy(inp["image"]), inp["depth"], inp["K"], inp["masks"])


def test_kernel_path_of_the_whole_preprocessing_matches_the_oracle(emu, monkeypatch):
    """S6D_PEM_PRE=kernels: compaction + back-projection, sequential centroid, radius filter (and the sampler kernel) against
    the oracle's per-detection loop -- bit-identical points, crops, indices and survivors, also at the radii whose sphere cuts
    through dense points (where the default path's float64-accumulated centroid flips boundary points)."""
    from oracle import pem_pre as opre
    monkeypatch.setenv("S6D_PEM_PRE", "kernels")                     # implies the sampler kernel
    called = []
    real = emu.pem_sample_indices
    monkeypatch.setattr(emu, "pem_sample_indices", lambda *a, **k: (called.append(1), real(*a, **k))[1])
    inp, args = _frame()
    kw = dict(n_sample=512, img_size=224, min_points=32, min_inliers=4, radius_factor=1.2)
    for radius in (0.12, np.array([0.12, 0.03, 0.5, 0.12, 0.06, 0.2, 0.07, 0.01])):
        ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(), radius,
                                    keys=inp["keys"].numpy(), **kw)
        r = torch.from_numpy(radius) if isinstance(radius, np.ndarray) else radius
        out = pre.observed_inputs(*args, r, keys=inp["keys"], **kw)
        assert out["kept"].tolist() == ref["kept"].tolist() and len(ref["kept"]) >= 6
        np.testing.assert_array_equal(out["bbox"].numpy(), ref["bbox"])
        np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])
        np.testing.assert_array_equal(out["rgb_choose"].numpy(), ref["rgb_choose"])
        np.testing.assert_array_equal(out["rgb"].numpy(), ref["rgb"])
    assert len(called) == 2
    # numpy-compatible draws on the kernel path too
    ref = opre.preprocess_frame(inp["image"], inp["depth"].numpy(), inp["K"].numpy(), inp["masks"].numpy(), 0.12,
                                rng=np.random.RandomState(5), **kw)
    out = pre.observed_inputs(*args, 0.12, rng=np.random.RandomState(5), **kw)
    np.testing.assert_array_equal(out["pts"].numpy(), ref["pts"])


def test_kernel_path_with_no_survivor_and_with_empty_masks(emu, monkeypatch):
    monkeypatch.setenv("S6D_PEM_PRE", "kernels")
    inp, args = _frame(P=3, seed=5)
    out = pre.observed_inputs(args[0], args[1] * 0, args[2], args[3], 0.1, inp["keys"])
    assert out["pts"].shape[0] == 0 and out["rgb"].shape == (0, 3, 224, 224) and out["kept"].numel() == 0
    masks = args[3].clone()
    masks[1] = False                                                 # one detection without a single pixel
    out = pre.observed_inputs(args[0], args[1], args[2], masks, 0.3, inp["keys"], n_sample=256)
    assert 1 not in out["kept"].tolist() and out["pts"].shape[1:] == (256, 3)


def test_mask_boxes_kernel_equals_the_library_ops(emu):
    """s6d_pem_mask_boxes_u8 against mask AND depth / count / square_boxes of the library path (itself pinned to the
    reference's get_bbox), incl. empty masks, masks touching the frame edge and a detection below the point threshold."""
    g = torch.Generator().manual_seed(4)
    P, H, W = 9, 48, 64
    masks = torch.zeros(P, H, W, dtype=torch.bool)
    for i, (y1, y2, x1, x2) in enumerate([(0, 48, 0, 64), (0, 5, 0, 64), (10, 40, 60, 64), (47, 48, 0, 1), (5, 45, 3, 9), (20, 21, 10, 50),
                                          (0, 0, 0, 0), (3, 44, 2, 63), (12, 30, 12, 30)]):
        masks[i, y1:y2, x1:x2] = torch.rand(y2 - y1, x2 - x1, generator=g) > 0.3
    depth = torch.rand(H, W, generator=g)
    depth[depth < 0.2] = 0.0
    m8, cnt, ok8, box = emu.pem_mask_boxes(masks.view(torch.uint8), depth, 32)
    m = masks & (depth > 0)[None]
    ok = m.flatten(1).sum(1) > 32
    assert torch.equal(m8.bool(), m) and torch.equal(cnt, m.flatten(1).sum(1)) and torch.equal(ok8.bool(), ok)
    assert 0 < int(ok.sum()) < P                                     # both kinds present
    assert torch.equal(box, pre.square_boxes(m | ~ok[:, None, None]))


def test_crops_kernel_equals_the_library_statement(emu):
    """s6d_pem_crops_f32 against preprocess._crops, value for value (same float32 operations in the same order)."""
    inp, (image, depth, K, masks) = _frame()
    m = masks & (depth > 0)[None]
    box = pre.square_boxes(m)
    kept = torch.tensor([6, 0, 2, 5])
    for flag in (True, False):
        want = pre._crops(image, m[kept].float(), box[kept], 224, flag)
        got = emu.pem_crops(image.contiguous(), m.to(torch.uint8), kept, box, 224, flag, pre.MEAN, pre.STD)
        assert torch.equal(got, want), (got - want).abs().max()
    want = pre._crops(image, m[kept].float(), box[kept], 56, True)
    assert torch.equal(emu.pem_crops(image.contiguous(), m.to(torch.uint8), kept, box, 56, True, pre.MEAN, pre.STD), want)


def test_crops_follow_the_cv2_restatement_at_every_ratio(emu):
    """s6d_pem_crops_f32 and preprocess._crops against oracle.pem_pre.cv2_resize_linear_u8 (cv2.resize INTER_LINEAR restated),
    value for value, at the ratios with their own code path: 1:1 (copy), exactly 2:1 (box mean), up- and down-scaling with odd
    sides, a crop touching the frame border."""
    import numpy as np

    from oracle import pem_pre as o
    from sam6d_amd.pem import preprocess as pre
    S = 16
    H, W = 72, 80
    g = torch.Generator().manual_seed(11)
    image = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
    sides = [16, 32, 7, 23, 48, 33, 5]
    box = torch.tensor([[y, y + n, x, x + n] for n, (y, x) in zip(sides, [(0, 0), (10, 20), (3, 70), (40, 5), (24, 32), (39, 47), (67, 75)])])
    m = torch.zeros(len(sides), H, W, dtype=torch.uint8)
    for i, (y1, y2, x1, x2) in enumerate(box.tolist()):
        m[i, y1:y2, x1:x2] = (torch.rand(y2 - y1, x2 - x1, generator=g) > 0.3).to(torch.uint8)
    kept = torch.arange(len(sides))
    for flag in (True, False):
        got = emu.pem_crops(image.contiguous(), m, kept, box, S, flag, pre.MEAN, pre.STD)
        lib = pre._crops(image, m, box, S, flag)
        assert torch.equal(got, lib)
        for i, (y1, y2, x1, x2) in enumerate(box.tolist()):
            rgb = image.numpy()[y1:y2, x1:x2, :][:, :, ::-1]
            if flag:
                rgb = rgb * (m[i].numpy()[y1:y2, x1:x2, None] > 0).astype(np.uint8)
            want = (o.cv2_resize_linear_u8(rgb, S).astype(np.float32) / np.float32(255) - o.MEAN) / o.STD
            assert np.array_equal(got[i].numpy(), want.transpose(2, 0, 1)), (i, flag)
