"""Module-surface variants and the training loop on the GPU: other widths (ViT-S D=384 -> N=1152 column tails, ViT-L D=1024),
final_reduction modes, learnable pos-embed, activation checkpointing, gradient accumulation through the engine, and a short
optimisation run whose loss must fall.  Reference semantics come from the oracle."""
import math

import pytest
import torch
import torch.nn.functional as F

import golden_recipe as R
import simple_tad_amd as T
from simple_tad_amd import engine as E
from simple_tad_amd.parallel import DataParallel
from oracle import vit_oracle as O

pytestmark = pytest.mark.gpu


def rell2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _model(embed_dim, heads, depth=2, **kw):
    torch.manual_seed(0)
    m = T.VisionTransformer(img_size=32, patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=heads, mlp_ratio=4, qkv_bias=True,
                            all_frames=4, tubelet_size=2, num_classes=2, init_scale=1.0, **kw)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.02)
    return m


@pytest.mark.parametrize("D,H", [(384, 6), (1024, 16), (1280, 20), (1280, 16)])
def test_other_widths_forward_backward(D, H):
    """(1280, 16) is the "huge" geometry (modeling_finetune.py:390-398): head_dim 80, since round 4 through 16-bit MFMA attention
    kernels of its own (the f32-kernel route of round 3 stays reachable for A/B: ops.set_attn_hd80_f32)"""
    m = _model(D, H).cuda().train()
    x = torch.randn(3, 3, 4, 32, 32)
    y = torch.tensor([0, 1, 1])
    logits = m(x.cuda())
    loss = F.cross_entropy(logits, y.cuda())
    loss.backward()
    P = {k: v.detach().double().cpu().requires_grad_() for k, v in m.state_dict().items()}
    ref = O.forward(x.double(), P, depth=2, num_heads=H, tubelet=2, patch=16)
    F.cross_entropy(ref, y).backward()
    assert rell2(logits, ref) < 1e-2
    for k in ("blocks.0.attn.qkv.weight", "blocks.1.mlp.fc2.weight", "patch_embed.proj.weight", "blocks.0.attn.q_bias", "fc_norm.weight"):
        got = dict(m.named_parameters())[k].grad
        assert rell2(got, P[k].grad) < 4e-2, (k, rell2(got, P[k].grad))
    if D // H == 80:  # the round-3 route (exact-f32 attention core between the 16-bit Linears) agrees within the 16-bit band
        from simple_tad_amd import ops
        ops.set_attn_hd80_f32(True)
        try:
            m.zero_grad()
            l2 = m(x.cuda())
            F.cross_entropy(l2, y.cuda()).backward()
        finally:
            ops.set_attn_hd80_f32(False)
        assert rell2(l2, ref) < 1e-2 and rell2(l2, logits) < 1e-2
        assert rell2(dict(m.named_parameters())["blocks.0.attn.qkv.weight"].grad, P["blocks.0.attn.qkv.weight"].grad) < 4e-2


@pytest.mark.parametrize("precise", [False, True])
def test_patch14_is_a_pure_config_change(precise):
    """BASELINE configs[4] names ViT-L/14: K = 3*2*14*14 = 1176 is not a multiple of the GEMM's K-tile.  The patch matrix and the
    weight operand are zero-padded to tad_patch_embed_ldk = 1216 inside the patch-embed path; nothing else changes.  Token / k
    bookkeeping bit-exact against the oracle, forward and every gradient of a small /14 model against the oracle."""
    from simple_tad_amd import kernels as K
    assert K.patch_embed_ldk(3, 2, 14) == 1216 and K.patch_embed_ldk(3, 2, 16) == 1536
    x = torch.randn(2, 3, 4, 28, 42)
    cols = K.im2col_tubelets(x.cuda().contiguous(), 2, 14).cpu()
    want = O.im2col_tubelets(x, 2, 14).reshape(-1, 1176)
    assert cols.shape == (2 * 2 * 2 * 3, 1216) and torch.equal(cols[:, :1176].float(), want.bfloat16().float())
    assert float(cols[:, 1176:].abs().sum()) == 0.0
    assert torch.equal(K.im2col_tubelets_f32(x.cuda().contiguous(), 2, 14).cpu()[:, :1176], want)
    torch.manual_seed(0)
    m = T.VisionTransformer(img_size=28, patch_size=14, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True, all_frames=4,
                            tubelet_size=2, num_classes=2, init_scale=1.0).cuda().train()
    assert m.patch_embed.num_patches == 8 and m.patch_embed.proj.weight.shape == (128, 3, 2, 14, 14)
    xs = torch.randn(3, 3, 4, 28, 28)
    y = torch.tensor([0, 1, 1])
    T.set_precision("precise" if precise else "fast")
    try:
        logits = m(xs.cuda())
        F.cross_entropy(logits, y.cuda()).backward()
    finally:
        T.set_precision("fast")
    P = {k: v.detach().double().cpu().requires_grad_() for k, v in m.state_dict().items()}
    ref = O.forward(xs.double(), P, depth=2, num_heads=2, tubelet=2, patch=14)
    F.cross_entropy(ref, y).backward()
    tol_f, tol_g = (1e-4, 1e-3) if precise else (1e-2, 4e-2)
    assert rell2(logits, ref) < tol_f, rell2(logits, ref)
    for k, p in m.named_parameters():
        assert rell2(p.grad, P[k].grad) < tol_g, (k, rell2(p.grad, P[k].grad))


def test_final_reduction_modes_learnable_pos_and_checkpointing():
    x = torch.randn(2, 3, 4, 32, 32)
    for fr in ("cls", "none"):
        m = _model(128, 2, final_reduction=fr).cuda().eval()
        with torch.no_grad():
            f = m.forward_features(x.cuda())
        P = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
        ref = O.forward_features(x.double(), P, depth=2, num_heads=2, tubelet=2, patch=16, final_reduction=fr)
        assert f.shape == ref.shape and rell2(f, ref) < 6e-3, fr
    # learnable positional embedding: pos_embed is a Parameter and appears in the state dict; it is added DETACHED, as in the
    # reference (modeling_finetune.py:312-313), so it receives no gradient
    m = _model(128, 2, use_learnable_pos_emb=True).cuda().train()
    assert "pos_embed" in m.state_dict()
    out = m(x.cuda())
    out.sum().backward()
    assert m.pos_embed.grad is None
    P = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
    ref = O.forward(x.double(), P, depth=2, num_heads=2, tubelet=2, patch=16, pos_embed=P["pos_embed"])
    assert rell2(out, ref) < 6e-3
    # activation checkpointing gives the same gradients as the plain path
    m1 = _model(128, 2).cuda().train()
    m2 = _model(128, 2, use_checkpoint=True).cuda().train()
    m2.load_state_dict(m1.state_dict())
    for mm in (m1, m2):
        F.cross_entropy(mm(x.cuda()), torch.tensor([1, 0], device="cuda")).backward()
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1.grad, p2.grad, rtol=0, atol=0), k   # same kernels, same order -> bit-identical


def test_training_loop_on_gpu_loss_falls_and_accumulation_matches():
    torch.manual_seed(0)
    m = _model(128, 2, drop_path_rate=0.1).cuda()
    dp = DataParallel(m)                       # world size 1: flat gradient buffer, no exchange
    opt = E.create_optimizer(dp, lr=2e-3, weight_decay=0.05, layer_decay=0.75)
    scaler = E.NativeScalerWithGradNormCount(dp)
    # learnable synthetic task: label = sign of the mean of channel 0
    xs = torch.randn(32, 3, 4, 32, 32)
    xs[:, 0] += (torch.arange(32) % 2).float().view(-1, 1, 1, 1) * 1.5 - 0.75
    ys = (torch.arange(32) % 2).long()
    data = [(xs[i:i + 8], ys[i:i + 8], None, None) for i in range(0, 32, 8)] * 12
    lr = E.cosine_scheduler(2e-3, 1e-5, 1, len(data), warmup_epochs=0)
    stats = E.train_one_epoch(dp, torch.nn.CrossEntropyLoss(), data, opt, torch.device("cuda"), 0, scaler, lr_schedule_values=lr,
                              num_training_steps_per_epoch=len(data), update_freq=1)
    first, last = sum(stats["loss"][:4]) / 4, sum(stats["loss"][-4:]) / 4
    assert all(math.isfinite(v) for v in stats["loss"]) and all(g is not None and g > 0 for g in stats["grad_norm"])
    assert last < 0.6 * first, (first, last)
    # gradient accumulation (update_freq=2 over two half batches) == one full batch, with drop-path off
    ma, mb = _model(128, 2).cuda().train(), _model(128, 2).cuda().train()
    mb.load_state_dict(ma.state_dict())
    xb, yb = xs[:8].cuda(), ys[:8].cuda()
    F.cross_entropy(ma(xb), yb).backward()
    for half in (slice(0, 4), slice(4, 8)):
        (F.cross_entropy(mb(xb[half]), yb[half]) / 2).backward()
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert rell2(pb.grad, pa.grad) < 2e-2, k
    # grad-norm definition on the GPU path == reference definition on the same gradients
    gn = E.get_grad_norm_(list(ma.parameters()))
    ref = O.grad_norm([p.grad.float().cpu() for p in ma.parameters()])
    assert abs(gn.item() - ref.item()) < 1e-4 * ref.item()


def test_ragged_and_tiny_inputs():
    """single clip, single token-row tiles, N not a multiple of any tile (8 tokens), batch that is not a multiple of anything"""
    m = _model(128, 2).cuda().eval()
    for B in (1, 5):
        x = torch.randn(B, 3, 4, 32, 32)
        with torch.no_grad():
            out = m(x.cuda())
        P = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
        assert rell2(out, O.forward(x.double(), P, depth=2, num_heads=2, tubelet=2, patch=16)) < 6e-3
    with pytest.raises(AssertionError):
        m(torch.randn(1, 3, 4, 48, 48).cuda())  # wrong spatial size: the reference's assert (modeling_finetune.py:188)


def test_weight_copies_follow_master_weights_across_fused_optimizer_steps():
    """torch's fused AdamW updates parameters without bumping Parameter._version, so the bf16 operand copies must not be keyed on
    it alone: after a few large steps, training-mode and inference-mode forwards must both see the CURRENT master weights."""
    m = _model(128, 2).cuda().train()
    x = torch.randn(4, 3, 4, 32, 32)
    y = torch.tensor([0, 1, 1, 0]).cuda()
    with torch.no_grad():
        m.eval()
        m(x.cuda())  # populate the inference-time cache before training starts
        m.train()
    w0 = m.blocks[0].mlp.fc1.weight.detach().clone()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-2, fused=True)
    for _ in range(3):
        opt.zero_grad()
        F.cross_entropy(m(x.cuda()), y).backward()
        opt.step()
    assert (m.blocks[0].mlp.fc1.weight - w0).abs().max() > 1e-2  # the weights did move
    P = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
    ref = O.forward(x.double(), P, depth=2, num_heads=2, tubelet=2, patch=16)
    out_train = m(x.cuda())
    m.eval()
    with torch.no_grad():
        out_eval = m(x.cuda())
    assert rell2(out_train, ref) < 1e-2, rell2(out_train, ref)
    assert rell2(out_eval, ref) < 1e-2, rell2(out_eval, ref)
    # and the gradient side: backward after the updates matches the oracle on the current weights
    m.train()
    opt.zero_grad()
    F.cross_entropy(m(x.cuda()), y).backward()
    Pg = {k: v.clone().requires_grad_() for k, v in P.items()}
    F.cross_entropy(O.forward(x.double(), Pg, depth=2, num_heads=2, tubelet=2, patch=16), y.cpu()).backward()
    for k in ("blocks.0.attn.qkv.weight", "blocks.1.mlp.fc1.weight", "patch_embed.proj.weight"):
        assert rell2(dict(m.named_parameters())[k].grad, Pg[k].grad) < 4e-2, k


def test_gradient_sinks_write_the_same_gradients_in_place():
    """With a flat gradient buffer (DataParallel / FusedAdamW) the dW GEMMs accumulate straight into p.grad and autograd sees None:
    the result must equal the plain autograd path bit for bit, accumulate over two backward passes, and still notify the
    bucket logic for every parameter."""
    ma, mb = _model(128, 2, drop_path_rate=0.0).cuda().train(), _model(128, 2, drop_path_rate=0.0).cuda().train()
    mb.load_state_dict(ma.state_dict())
    x = torch.randn(4, 3, 4, 32, 32).cuda()
    y = torch.tensor([0, 1, 1, 0]).cuda()
    F.cross_entropy(ma(x), y).backward()
    dp = DataParallel(mb)
    seen = []
    orig = dp._on_grad
    dp._on_grad = lambda p: (seen.append(id(p)), orig(p))[1]
    dp.space.install_sinks(dp._on_grad)
    F.cross_entropy(dp(x), y).backward()
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert pb.grad.data_ptr() == dp.space.grad_view(pb).data_ptr(), k
        assert torch.equal(pa.grad, pb.grad), k
    sunk = [k for k, p in mb.named_parameters() if id(p) in seen]
    assert {"blocks.0.attn.qkv.weight", "blocks.1.mlp.fc2.bias", "patch_embed.proj.weight", "blocks.0.attn.proj.weight"} <= set(sunk)
    F.cross_entropy(dp(x), y).backward()  # second micro-step accumulates
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert rell2(pb.grad, 2 * pa.grad) < 1e-6, k


def test_block_chain_handoff_is_bit_identical_and_taken():
    """Block i+1's LayerNorm backward emits the bf16, drop-path-scaled copy of the residual-stream gradient that block i's backward
    starts from (ops._ChainLink): same gradients bit for bit as the separate cast pass, and the hand-off is actually used."""
    from simple_tad_amd import ops
    m = _model(128, 2, depth=3, drop_path_rate=0.3).cuda().train()
    x = torch.randn(4, 3, 4, 32, 32).cuda()
    y = torch.tensor([0, 1, 1, 0]).cuda()
    grads = {}
    try:
        for on in (False, True):
            ops.set_block_chain(on)
            ops._chain.hits = 0
            m.zero_grad(set_to_none=True)
            torch.manual_seed(7)  # same stochastic-depth masks in both runs
            F.cross_entropy(m(x), y).backward()
            grads[on] = {k: p.grad.clone() for k, p in m.named_parameters()}
            assert ops._chain.hits == (2 if on else 0)  # blocks 0 and 1 take the copy made by blocks 1 and 2
            assert ops._chain.pending == 0
    finally:
        ops.set_block_chain(True)
    for k in grads[True]:
        assert torch.equal(grads[True][k], grads[False][k]), k


def test_block_chain_with_second_consumer_and_across_iterations():
    """(1) a block output with a second consumer (auxiliary loss on an intermediate feature map): autograd sums the two
    gradients, so the bf16 copy made by the next block's LayerNorm backward must NOT be taken -- gradients equal the chain-off run bit
    for bit; (2) several iterations through the MAE model (which never called the old reset) leave no deposit behind and take
    the hand-off in every encoder / decoder block pair."""
    from simple_tad_amd import ops
    import simple_tad_amd.modeling_pretrain as mp
    m = _model(128, 2, depth=3, drop_path_rate=0.0).cuda().train()
    x = torch.randn(2, 3, 4, 32, 32).cuda()
    grads = {}
    try:
        for on in (False, True):
            ops.set_block_chain(on)
            ops._chain.hits = 0
            m.zero_grad(set_to_none=True)
            t = m.patch_embed(x, pos_embed=m._pos_on(x.device))
            mids = []
            for blk in m.blocks:
                t = blk(t)
                mids.append(t)
            (t.mean() + 3.0 * mids[0].square().mean() + 2.0 * mids[1].abs().mean()).backward()
            grads[on] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
            assert ops._chain.hits == 0 and ops._chain.pending == 0   # both hand-offs are invalidated by the second consumers
    finally:
        ops.set_block_chain(True)
    for k in grads[True]:
        assert torch.equal(grads[True][k], grads[False][k]), k
    pm = mp.PretrainVisionTransformer(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2,
                                      decoder_num_classes=1536, decoder_embed_dim=64, decoder_depth=2, decoder_num_heads=1, mlp_ratio=4,
                                      qkv_bias=True, init_values=0., tubelet_size=2).cuda().train()
    clip = torch.randn(2, 3, 16, 32, 32).cuda()
    mask = torch.zeros(2, 32, dtype=torch.bool)
    mask[:, [1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14, 15, 17, 18, 19, 21, 22, 23, 25, 26, 27, 29, 30, 31]] = True
    for it in range(3):
        ops._chain.hits = 0
        pm.zero_grad(set_to_none=True)
        pm(clip, mask.cuda()).square().mean().backward()
        assert ops._chain.hits == 2 + 1 and ops._chain.pending == 0, (it, ops._chain.hits, ops._chain.pending)


def test_learnable_pos_embed_is_added_detached_as_in_the_reference():
    """modeling_finetune.py:312-313 adds ``pos_embed.expand(...).clone().detach()``: the learnable table shapes the output but never
    receives a gradient (so AdamW leaves it alone)."""
    m = _model(128, 2, depth=1, use_learnable_pos_emb=True).cuda().train()
    assert isinstance(m.pos_embed, torch.nn.Parameter) and float(m.pos_embed.abs().max()) > 0
    x = torch.randn(2, 3, 4, 32, 32).cuda()
    y0 = m.forward_features(x)
    y0.sum().backward()
    assert m.pos_embed.grad is None and m.patch_embed.proj.weight.grad is not None
    with torch.no_grad():
        m.pos_embed.add_(torch.randn_like(m.pos_embed) * 0.5)   # (a constant shift would be removed by the LayerNorms)
    assert (m.forward_features(x) - y0).abs().max() > 1e-3   # the table does take part in the forward


def test_no_grad_forward_reuses_weight_copies_and_keeps_nothing_for_backward():
    """Under torch.no_grad() ctx.needs_input_grad is still True for Parameters and grad mode is off inside Function.forward anyway,
    so the Functions sample the CALLER's grad mode (ops._Fn.apply): an inference forward must not re-cast the weights (after the first
    call) and must give the same logits as a differentiated forward."""
    from simple_tad_amd import kernels as K
    m = _model(128, 2, depth=2, drop_path_rate=0.0).cuda().eval()
    x = torch.randn(2, 3, 4, 32, 32).cuda()
    with torch.no_grad():
        y0 = m(x)  # fills the version-keyed copies
        prof = K.LaunchProfiler(only=["cast"])
        K.set_profiler(prof)
        try:
            y1 = m(x)
        finally:
            K.set_profiler(None)
    # the input clip / activations may still be cast (LayerNorm outputs are produced in bf16 directly): no WEIGHT-sized casts -> at most
    # the handful of activation casts of the unfused entry / exit
    assert prof.seen.get("cast", 0) <= 2, prof.seen
    y2 = m(x)  # grad mode on: fresh copies, statistics kept
    assert torch.equal(y0, y1) and torch.equal(y1, y2.detach())
    y2.sum().backward()
    assert all(p.grad is not None for p in m.parameters())
