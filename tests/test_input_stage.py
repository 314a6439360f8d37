"""Input stage: uint8 frames -> normalised patch matrix.  CPU: the oracle restatements against the reference's own
outputs (golden G7, bit-exact).  GPU: the fused kernel against the oracle (bit-exact on the bf16 patch matrix), the model on
uint8 frames against the model on the reference's float window (bit-identical logits), and the ring-buffer sliding window
against re-running on the shifted window frame by frame."""
import numpy as np
import pytest
import torch

import golden_recipe as R
from oracle import vit_oracle as O

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def test_oracle_normalisation_bit_exact_vs_reference(golden):
    g = golden("g7_input_stage")
    frame = R.uint8_for("g7.frame", (32, 48, 3)).numpy()
    assert np.array_equal(O.prepare_image(frame, MEAN, STD).numpy(), g["prepare_image"])
    assert np.array_equal(O.prepare_image(frame, (0.5,) * 3, (0.5,) * 3).numpy(), g["prepare_image_center"])
    clip = R.uint8_for("g7.clip", (4, 16, 16, 3))
    assert np.array_equal(O.tensor_normalize(clip, MEAN, STD).numpy(), g["tensor_normalize"])
    ramp = torch.arange(256, dtype=torch.uint8).view(1, 256, 1).repeat(1, 1, 3)
    assert np.array_equal(O.prepare_image(ramp.numpy(), MEAN, STD).numpy(), g["ramp_prepare_image"])
    assert np.array_equal(O.tensor_normalize(ramp.view(1, 1, 256, 3), MEAN, STD).numpy(), g["ramp_tensor_normalize"])
    # the two reference entry points agree with each other on every byte value (RGB ramp: the BGR flip is a no-op on equal channels)
    assert np.array_equal(g["ramp_prepare_image"].transpose(1, 2, 0).reshape(-1), g["ramp_tensor_normalize"].reshape(-1))
    with pytest.raises(TypeError):
        O.prepare_image(np.zeros((4, 4), np.uint8), MEAN, STD)


@pytest.mark.gpu
@pytest.mark.parametrize("bgr", [False, True])
def test_u8_patch_matrix_bit_exact(bgr, golden):
    from simple_tad_amd import kernels as K
    g = golden("g7_input_stage")
    B, T, H, W = 2, 4, 32, 48
    frames = R.uint8_for("u8.frames", (B, T, H, W, 3))
    # expected: reference arithmetic per frame (oracle, pinned above) -> [B,3,T,H,W] f32 -> the f32 path's im2col (bf16 RNE)
    ref = torch.stack([torch.stack([O.prepare_image(frames[b, t].numpy() if bgr else frames[b, t].numpy()[..., ::-1], MEAN, STD)
                                    for t in range(T)], dim=1) for b in range(B)])
    want = K.im2col_tubelets(ref.cuda().contiguous(), 2, 16)
    got = K.im2col_tubelets_u8(frames.cuda(), 2, 16, MEAN, STD, bgr=bgr)
    assert got.shape == want.shape and torch.equal(got.view(torch.int16), want.view(torch.int16))
    # every byte value through the kernel == the reference's table (after the same bf16 rounding)
    ramp = torch.arange(256, dtype=torch.uint8).repeat(2 * 16 * 16 * 3 // 256 + 1)[: 2 * 16 * 16].view(1, 2, 16, 16, 1).repeat(1, 1, 1, 1, 3)
    cols = K.im2col_tubelets_u8(ramp.cuda().contiguous(), 2, 16, MEAN, STD)
    table = torch.from_numpy(g["ramp_tensor_normalize"]).view(256, 3)
    exp = table[ramp[0, :, :, :, 0].reshape(-1).long()]                      # [(kt,kh,kw), c] in f32
    for c in range(3):
        assert torch.equal(cols[0, c * 512:(c + 1) * 512].cpu(), exp[:, c].to(torch.bfloat16))
    # ring offset: slot (t + off) % T holds frame t
    off = 3
    rolled = torch.roll(frames, shifts=off, dims=1)
    assert torch.equal(K.im2col_tubelets_u8(rolled.cuda(), 2, 16, MEAN, STD, bgr=bgr, t_offset=off), got)
    with pytest.raises(Exception):
        K.im2col_tubelets_u8(frames.cuda(), 2, 16, MEAN, (0.0, 1.0, 1.0))


@pytest.mark.gpu
def test_model_on_uint8_frames_and_sliding_window():
    import simple_tad_amd as T
    from simple_tad_amd.inference import SlidingWindow
    torch.manual_seed(0)
    m = T.VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True, all_frames=4,
                            tubelet_size=2, num_classes=2, init_scale=1.0).cuda().eval()
    frames = [R.uint8_for(f"sw.{i}", (32, 32, 3)).numpy() for i in range(9)]  # cv2-style BGR frames
    sw = SlidingWindow(m, MEAN, STD, bgr=True)
    with pytest.raises(Exception):
        sw.predict()  # "We need at least T frames!" (run_inference.py:73)
    outs = []
    for i, f in enumerate(frames):
        sw.push(f)
        if sw.full:
            outs.append(sw.predict())
            # reference flow: window of the last T prepared frames, [1,3,T,H,W] f32 (run_inference.py:86-96)
            win = O.clip_from_frames(frames[i - 3:i + 1], MEAN, STD)
            assert torch.allclose(sw.window_f32().cpu(), win, rtol=0, atol=1e-6)  # (torch's GPU scalar division is not IEEE-exact)
            with torch.no_grad():
                want = m(win.cuda())
            assert torch.equal(outs[-1], want), i    # same patch matrix bits -> same logits bits
    assert len(outs) == 6 and not torch.equal(outs[0], outs[-1])
    # HIP-graph replay (one captured graph per ring offset): same bits as the eager path, also on the second lap of the ring
    swg = SlidingWindow(m, MEAN, STD, bgr=True, use_graph=True)
    got = []
    for lap in range(2):
        for f in frames[:8]:
            swg.push(f)
            if swg.full:
                got.append(swg.predict())
    swe = SlidingWindow(m, MEAN, STD, bgr=True)
    want = []
    for lap in range(2):
        for f in frames[:8]:
            swe.push(f)
            if swe.full:
                want.append(swe.predict())
    assert len(got) == 13 and len(swg._graphs) == 4 and all(torch.equal(a, b) for a, b in zip(got, want))
    # the graphs follow the weights: an in-place update drops them (the old bf16 weight copies were baked in by address)
    with torch.no_grad():
        m.head.weight.mul_(2.0)
        m.blocks[0].mlp.fc1.weight.mul_(1.5)
    assert torch.equal(swg.predict(), swe.predict()) and len(swg._graphs) == 1 and not torch.equal(swg.predict(), got[-1])
    with pytest.raises(TypeError):
        sw.push(np.zeros((16, 16, 3), np.uint8))
    # training from uint8 clips (the loaders' [T,H,W,C] RGB buffers, dota.py:312): gradients equal the float path's
    m.train()
    m.patch_embed.set_input_normalization(MEAN, STD, bgr=False)
    clip = R.uint8_for("train.clip", (2, 4, 32, 32, 3))
    xf = torch.stack([O.tensor_normalize(clip[b], MEAN, STD).permute(3, 0, 1, 2) for b in range(2)]).contiguous()
    m(clip.cuda()).sum().backward()
    g_u8 = m.patch_embed.proj.weight.grad.clone()
    m.zero_grad()
    m(xf.cuda()).sum().backward()
    assert torch.equal(g_u8, m.patch_embed.proj.weight.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("bgr", [False, True])
def test_u8_patch_matrix_for_patch_14_bit_exact_and_model_equal(bgr):
    """Patch sizes that are even but not multiples of 8 (BASELINE configs[4] says ViT-L/14): the uint8 input stage writes the same
    zero-padded patch matrix (row stride tad_patch_embed_ldk: 1176 -> 1216) as the f32 path does from the reference's normalised clip,
    bit for bit, and the model's logits and patch-embed gradient on uint8 frames equal the float path's."""
    import simple_tad_amd as T
    from simple_tad_amd import kernels as K
    B, T_, H, W = 2, 4, 28, 42
    frames = R.uint8_for("u8p14.frames", (B, T_, H, W, 3))
    ref = torch.stack([torch.stack([O.prepare_image(frames[b, t].numpy() if bgr else frames[b, t].numpy()[..., ::-1], MEAN, STD)
                                    for t in range(T_)], dim=1) for b in range(B)])
    want = K.im2col_tubelets(ref.cuda().contiguous(), 2, 14)
    got = K.im2col_tubelets_u8(frames.cuda(), 2, 14, MEAN, STD, bgr=bgr)
    assert got.shape == want.shape == (B * 2 * 2 * 3, 1216) and torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert float(got[:, 1176:].float().abs().max()) == 0.0
    off = 1
    assert torch.equal(K.im2col_tubelets_u8(torch.roll(frames, shifts=off, dims=1).cuda(), 2, 14, MEAN, STD, bgr=bgr, t_offset=off), got)
    # the half format takes the same route
    assert torch.equal(K.im2col_tubelets_u8(frames.cuda(), 2, 14, MEAN, STD, bgr=bgr, dtype=torch.float16),
                       K.im2col_tubelets(ref.cuda().contiguous(), 2, 14, dtype=torch.float16))
    if bgr:
        return
    torch.manual_seed(0)
    m = T.VisionTransformer(img_size=28, patch_size=14, embed_dim=128, depth=1, num_heads=2, mlp_ratio=4, qkv_bias=True, all_frames=4,
                            tubelet_size=2, num_classes=2, init_scale=1.0).cuda().train()
    m.patch_embed.set_input_normalization(MEAN, STD, bgr=False)
    clip = R.uint8_for("u8p14.clip", (2, 4, 28, 28, 3))
    xf = torch.stack([O.tensor_normalize(clip[b], MEAN, STD).permute(3, 0, 1, 2) for b in range(2)]).contiguous()
    y8 = m(clip.cuda())
    y8.sum().backward()
    g8 = m.patch_embed.proj.weight.grad.clone()
    m.zero_grad()
    yf = m(xf.cuda())
    yf.sum().backward()
    assert torch.equal(y8, yf) and torch.equal(g8, m.patch_embed.proj.weight.grad)
