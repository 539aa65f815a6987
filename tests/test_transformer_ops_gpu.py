"""Fused attention / LayerNorm HIP kernels against the PyTorch fp32 reference of the same op."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf16r(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("kind", ["softmax", "relu"])
@pytest.mark.parametrize("S,dk,dv", [(50, 16, 16), (96, 32, 32), (128, 64, 64), (17, 8, 24), (65, 24, 8), (40, 16, 16),
                                      (100, 32, 48)])      # 3, 5 and 7 blocks of 16 tokens have kernels of their own
@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_fused_attention_forward_backward(kind, S, dk, dv, drop):
    from mitune.ops.attention import attention_reference
    from mitune.ops.attention_hip import FusedShortAttention
    g = torch.Generator(device="cuda").manual_seed(S * 7 + dk)
    B, h = 3, 2
    q = _bf16r(torch.randn(B, h, S, dk, device="cuda", generator=g)).requires_grad_(True)
    k = _bf16r(torch.randn(B, h, S, dk, device="cuda", generator=g)).requires_grad_(True)
    v = _bf16r(torch.randn(B, h, S, dv, device="cuda", generator=g)).requires_grad_(True)
    keep = None
    if drop > 0:
        keep = torch.bernoulli(torch.full((B, h, S, S), 1 - drop, device="cuda"), generator=g)
    do = _bf16r(torch.randn(B, h, S, dv, device="cuda", generator=g))
    out = FusedShortAttention.apply(q, k, v, kind, keep, drop)
    out.backward(do)
    got = (out.detach(), q.grad.clone(), k.grad.clone(), v.grad.clone())
    q.grad = k.grad = v.grad = None
    ref = attention_reference(q, k, v, kind, keep, drop)
    ref.backward(do)
    exp = (ref.detach(), q.grad, k.grad, v.grad)
    for name, a, b in zip(("out", "dq", "dk", "dv"), got, exp):
        scale = b.abs().max().item() + 1e-6
        err = (a - b).abs().max().item() / scale
        assert err < 3e-2, (name, err)
        assert torch.isfinite(a).all(), name


def test_attention_asymmetric_values_catch_transposes():
    """Identity-like attention (one hot scores) must route V rows exactly."""
    from mitune.ops.attention_hip import FusedShortAttention
    S, d = 64, 32
    q = torch.zeros(1, 1, S, d, device="cuda"); k = torch.zeros(1, 1, S, d, device="cuda")
    perm = torch.randperm(S, device="cuda")
    for i in range(S):                     # query i matches key perm[i] through a unique pair of +-16 entries
        q[0, 0, i, i % d] = 16.0 * (1 if i < d else -1)
        k[0, 0, perm[i], i % d] = 16.0 * (1 if i < d else -1)
    v = (torch.arange(S * d, device="cuda").reshape(1, 1, S, d) % 61).float()
    out = FusedShortAttention.apply(q, k, v, "softmax", None, 0.0)
    assert torch.allclose(out[0, 0], v[0, 0, perm], atol=0.26)      # bf16 output rounding of values <= 60


@pytest.mark.parametrize("d", [32, 128, 320, 512, 2048])
@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_fused_residual_layernorm(d, drop):
    from mitune.ops.layernorm import residual_layer_norm_reference
    from mitune.ops.layernorm_hip import FusedResidualLayerNorm
    g = torch.Generator(device="cuda").manual_seed(d)
    M = 77
    x = torch.randn(M, d, device="cuda", generator=g, requires_grad=True)
    r = torch.randn(M, d, device="cuda", generator=g, requires_grad=True)
    w = torch.randn(d, device="cuda", generator=g, requires_grad=True)
    b = torch.randn(d, device="cuda", generator=g, requires_grad=True)
    keep = torch.bernoulli(torch.full((M, d), 1 - drop, device="cuda"), generator=g) if drop > 0 else None
    dy = torch.randn(M, d, device="cuda", generator=g)
    y = FusedResidualLayerNorm.apply(x, r, w, b, 1e-5, keep, drop)
    y.backward(dy)
    got = [y.detach()] + [t.grad.clone() for t in (x, r, w, b)]
    for t in (x, r, w, b):
        t.grad = None
    yr = residual_layer_norm_reference(x, r, w, b, 1e-5, keep, drop)
    yr.backward(dy)
    exp = [yr.detach()] + [t.grad for t in (x, r, w, b)]
    for name, a, e in zip(("y", "dx", "dres", "dw", "db"), got, exp):
        torch.testing.assert_close(a, e, rtol=2e-4, atol=2e-4, msg=lambda m, n=name: f"{n}: {m}")


def test_transformer_models_run_on_the_fused_kernels():
    from mitune.models.transformer import SimpleTransformerModel, TransformerModel
    torch.manual_seed(0)
    x = torch.randn(8, 96, 10)
    for attn in ("scaled_dot_product", "linear_attention"):
        m = TransformerModel(10, 4, 2, 64, 128, 0.1, 96, attn, 3, False, 1.0, False, 0.05)
        ref = m.eval()(x)
        mg = m.cuda()
        out = mg.eval()(x.cuda())
        assert torch.allclose(out.cpu(), ref, atol=3e-2, rtol=3e-2)
        mg.train()
        loss = mg(x.cuda()).pow(2).mean()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mg.parameters())
    s = SimpleTransformerModel(10, 32, 4, 2, 0.1).cuda()
    s(torch.randn(4, 50, 10, device="cuda")).sum().backward()


@pytest.mark.parametrize("dtype,level", [(torch.float32, 0), (torch.float32, 2), (torch.bfloat16, 2)])    # the level only concerns f32 operands
@pytest.mark.parametrize("kind", ["softmax", "relu"])
@pytest.mark.parametrize("B,S,h,dk,dv,drop", [(4, 50, 4, 16, 16, 0.0), (3, 65, 8, 24, 24, 0.2), (2, 96, 2, 64, 32, 0.1),
                                              (5, 17, 2, 8, 40, 0.0),
                                              # head widths off the 8-element grid (element-wise staging) ...
                                              (3, 96, 16, 3, 12, 0.1), (2, 50, 32, 2, 2, 0.0), (2, 96, 32, 5, 10, 0.0),
                                              (2, 33, 3, 6, 20, 0.2), (2, 96, 4, 2, 4, 0.0),
                                              # ... and wider than 64 (256-byte image rows, S <= 96)
                                              (2, 96, 2, 96, 96, 0.1), (2, 64, 2, 128, 128, 0.0), (3, 50, 2, 40, 80, 0.0)])
def test_strided_self_attention_on_projection_outputs(B, S, h, dk, dv, drop, kind, dtype, level):
    """The kernels read Q and K in place from one [B, S, 2*h*dk] tensor and V / O as [B, S, h*dv].  Level 0 sends f32
    operands through the matrix form too (what heads wider than 64 elements always get), level 2 is the default."""
    from mitune.ops import lib as L
    from mitune.ops.attention import attention_reference
    from mitune.ops.attention_hip import FusedSelfAttention
    g = torch.Generator(device="cuda").manual_seed(3)
    qk = torch.randn(B, S, 2 * h * dk, device="cuda", generator=g).to(dtype).requires_grad_()
    v = torch.randn(B, S, h * dv, device="cuda", generator=g).to(dtype).requires_grad_()
    keep = (torch.rand(B, h, S, S, device="cuda", generator=g) >= drop).to(torch.uint8) if drop > 0 else None
    do = torch.randn(B, S, h * dv, device="cuda", generator=g).to(dtype)
    so = L.load()
    was = so.mt_attn_exact_level(level)
    try:
        out = FusedSelfAttention.apply(qk, v, h, kind, keep, drop)
        out.backward(do)
        torch.cuda.synchronize()
    finally:
        so.mt_attn_exact_level(was)
    got = (out.float().detach(), qk.grad.float().clone(), v.grad.float().clone())
    # reference in f32 on the operands the kernel multiplies: rounded to bf16 in the matrix form, as they are in the
    # exact form (f32 operands, heads of at most 64 elements, level 2)
    exact = dtype == torch.float32 and level == 2 and dk <= 64 and dv <= 64
    rnd = (lambda t: t) if exact else (lambda t: t.bfloat16().float())          # noqa: E731
    qr = rnd(qk.detach()).requires_grad_()
    vr = rnd(v.detach()).requires_grad_()
    q, k = qr.chunk(2, dim=-1)
    sp = lambda t, d: t.reshape(B, S, h, d).transpose(1, 2)          # noqa: E731
    ref = attention_reference(sp(q, dk), sp(k, dk), sp(vr, dv), kind, keep, drop).transpose(1, 2).reshape(B, S, h * dv)
    ref.backward(rnd(do))
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-2
    for a, r, name in zip(got, (ref.detach(), qr.grad, vr.grad), ("out", "dqk", "dv")):
        assert torch.allclose(a, r, rtol=tol, atol=tol * max(1.0, r.abs().max().item())), name


@pytest.mark.parametrize("kind", ["softmax", "relu"])
@pytest.mark.parametrize("B,S,h,dk,dv,drop", [(6, 10, 4, 8, 8, 0.0), (4, 10, 2, 32, 32, 0.1), (3, 17, 3, 5, 20, 0.2),
                                              (2, 32, 2, 64, 64, 0.0), (5, 1, 2, 16, 16, 0.0), (2, 31, 8, 24, 12, 0.3)])
def test_short_sequence_f32_attention_is_exact_f32(B, S, h, dk, dv, drop, kind):
    """f32 operands, at most 32 tokens, heads of at most 64 elements: nothing is rounded to bf16.  Against a float64
    reference on the SAME (unrounded) operands the error is f32 summation error only."""
    from mitune.ops.attention import attention_reference
    from mitune.ops.attention_hip import FusedSelfAttention
    g = torch.Generator(device="cuda").manual_seed(21)
    qk = torch.randn(B, S, 2 * h * dk, device="cuda", generator=g).requires_grad_()
    v = torch.randn(B, S, h * dv, device="cuda", generator=g).requires_grad_()
    keep = (torch.rand(B, h, S, S, device="cuda", generator=g) >= drop).to(torch.uint8) if drop > 0 else None
    do = torch.randn(B, S, h * dv, device="cuda", generator=g)
    out = FusedSelfAttention.apply(qk, v, h, kind, keep, drop)
    out.backward(do)
    qr, vr = qk.detach().double().requires_grad_(), v.detach().double().requires_grad_()
    q, k = qr.chunk(2, dim=-1)
    sp = lambda t, d: t.reshape(B, S, h, d).transpose(1, 2)          # noqa: E731
    ref = attention_reference(sp(q, dk), sp(k, dk), sp(vr, dv), kind, keep, drop).transpose(1, 2).reshape(B, S, h * dv)
    ref.backward(do.double())
    for a, r, name in ((out.detach(), ref.detach(), "out"), (qk.grad, qr.grad, "dqk"), (v.grad, vr.grad, "dv")):
        torch.testing.assert_close(a.double(), r, rtol=2e-5, atol=2e-5 * max(1.0, float(r.abs().max())), msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("kind", ["softmax", "relu"])
@pytest.mark.parametrize("B,S,h,dk,dv,drop", [(3, 96, 4, 32, 32, 0.1), (2, 128, 2, 64, 64, 0.0), (2, 33, 3, 5, 20, 0.2),
                                              (4, 50, 8, 16, 8, 0.0), (2, 97, 1, 64, 24, 0.3)])
def test_tiled_exact_f32_attention_for_sequences_of_up_to_128_tokens(B, S, h, dk, dv, drop, kind):
    """Level 2 of the exact-f32 attention (query blocks of 32 against K and V resident in LDS): float64 reference on the
    unrounded operands, f32 summation error only."""
    from mitune.ops import lib as L
    from mitune.ops.attention import attention_reference
    from mitune.ops.attention_hip import FusedSelfAttention
    so = L.load()
    was = so.mt_attn_exact_level(2)
    try:
        g = torch.Generator(device="cuda").manual_seed(23)
        qk = torch.randn(B, S, 2 * h * dk, device="cuda", generator=g).requires_grad_()
        v = torch.randn(B, S, h * dv, device="cuda", generator=g).requires_grad_()
        keep = (torch.rand(B, h, S, S, device="cuda", generator=g) >= drop).to(torch.uint8) if drop > 0 else None
        do = torch.randn(B, S, h * dv, device="cuda", generator=g)
        out = FusedSelfAttention.apply(qk, v, h, kind, keep, drop)
        out.backward(do)
        torch.cuda.synchronize()
    finally:
        so.mt_attn_exact_level(was)
    qr, vr = qk.detach().double().requires_grad_(), v.detach().double().requires_grad_()
    q, k = qr.chunk(2, dim=-1)
    sp = lambda t, d: t.reshape(B, S, h, d).transpose(1, 2)          # noqa: E731
    ref = attention_reference(sp(q, dk), sp(k, dk), sp(vr, dv), kind, keep, drop).transpose(1, 2).reshape(B, S, h * dv)
    ref.backward(do.double())
    for a, r, name in ((out.detach(), ref.detach(), "out"), (qk.grad, qr.grad, "dqk"), (v.grad, vr.grad, "dv")):
        torch.testing.assert_close(a.double(), r, rtol=3e-5, atol=3e-5 * max(1.0, float(r.abs().max())), msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("B,S,h,dk,dv,dtype", [(3, 50, 4, 16, 16, torch.float32), (2, 65, 8, 24, 24, torch.bfloat16),
                                                (2, 20, 2, 8, 16, torch.float32), (1, 128, 4, 32, 64, torch.bfloat16),
                                                (2, 96, 16, 3, 12, torch.float32), (2, 96, 32, 5, 10, torch.bfloat16),
                                                (2, 96, 2, 48, 96, torch.float32)])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_fused_qkv_attention_equals_the_two_operand_form(B, S, h, dk, dv, dtype, p):
    """One [B, S, 2*h*dk + h*dv] projection output read in place, one gradient tensor back: bitwise the values the
    kernels give on separate contiguous qk / v tensors (which are checked against the f32 reference above)."""
    from mitune.ops.attention_hip import FusedQKVAttention, FusedSelfAttention
    g = torch.Generator(device="cuda").manual_seed(11)
    W = 2 * h * dk + h * dv
    qkv = torch.randn(B, S, W, device="cuda", generator=g).to(dtype)
    do = torch.randn(B, S, h * dv, device="cuda", generator=g).to(dtype)
    seed = torch.tensor([0x5DEECE66D], dtype=torch.int64, device="cuda") if p > 0 else None
    a = qkv.clone().requires_grad_()
    out = FusedQKVAttention.apply(a, h, dk, "softmax", p, seed)
    out.backward(do)
    qk = qkv[..., :2 * h * dk].clone().requires_grad_()
    v = qkv[..., 2 * h * dk:].clone().requires_grad_()
    ref = FusedSelfAttention.apply(qk, v, h, "softmax", None, p, seed)
    ref.backward(do)
    assert a.grad.shape == qkv.shape and a.grad.dtype == dtype
    assert torch.equal(out, ref)
    assert torch.equal(a.grad[..., :2 * h * dk], qk.grad)
    assert torch.equal(a.grad[..., 2 * h * dk:], v.grad)


def test_in_kernel_dropout_masks_equal_the_materialised_mask():
    """Seed-driven masks inside the attention and LayerNorm kernels are the masks ``keep_mask`` would write for the same
    seed, in forward and in backward."""
    from mitune.ops import lib as L
    from mitune.ops.attention_hip import FusedSelfAttention
    from mitune.ops.layernorm_hip import FusedResidualLayerNorm
    so = L.load()
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    seed = torch.tensor([0x1234ABCD5678], dtype=torch.int64, device=dev)

    def mask(shape, p):
        out = torch.empty(shape, dtype=torch.uint8, device=dev)
        L.check(so.mt_keep_mask_launch(out.data_ptr(), out.numel(), seed.data_ptr(), p, torch.cuda.current_stream().cuda_stream), "mask")
        return out

    B, S, h, dk, dv, p = 3, 50, 4, 16, 16, 0.25
    for dtype in (torch.float32, torch.bfloat16):
        qk = torch.randn(B, S, 2 * h * dk, device=dev, generator=g).to(dtype)
        v = torch.randn(B, S, h * dv, device=dev, generator=g).to(dtype)
        do = torch.randn(B, S, h * dv, device=dev, generator=g).to(dtype)
        res = []
        for use_seed in (True, False):
            a, b = qk.clone().requires_grad_(), v.clone().requires_grad_()
            keep = None if use_seed else mask((B, h, S, S), p)
            out = FusedSelfAttention.apply(a, b, h, "softmax", keep, p, seed if use_seed else None)
            out.backward(do)
            res.append((out.detach(), a.grad, b.grad))
        for x, y in zip(*res):
            assert torch.equal(x, y)
    x = torch.randn(4, 37, 192, device=dev, generator=g)
    r = torch.randn(4, 37, 192, device=dev, generator=g)
    w = torch.randn(192, device=dev, generator=g)
    bias = torch.randn(192, device=dev, generator=g)
    dy = torch.randn(4, 37, 192, device=dev, generator=g)
    for d_cols in (192, 512):
        xs, rs = (x, r) if d_cols == 192 else (torch.randn(3, 9, 512, device=dev, generator=g), torch.randn(3, 9, 512, device=dev, generator=g))
        ws, bs = (w, bias) if d_cols == 192 else (torch.randn(512, device=dev, generator=g), torch.randn(512, device=dev, generator=g))
        dys = dy if d_cols == 192 else torch.randn(3, 9, 512, device=dev, generator=g)
        res = []
        for use_seed in (True, False):
            a, b = xs.clone().requires_grad_(), rs.clone().requires_grad_()
            keep = None if use_seed else mask(xs.shape, 0.3)
            y = FusedResidualLayerNorm.apply(a, b, ws, bs, 1e-5, keep, 0.3, seed if use_seed else None)
            y.backward(dys)
            res.append((y.detach(), a.grad, b.grad))
        for x1, x2 in zip(*res):
            assert torch.equal(x1, x2)


# ------------------------------------------------------------------ fused small-linear backward
def _producer(y):
    """The autograd node that produced ``y``, looking through the slice that drops padded output columns."""
    fn = y.grad_fn
    while fn is not None and type(fn).__name__.startswith(("Slice", "Alias", "View")):
        fn = fn.next_functions[0][0]
    return fn


@pytest.mark.parametrize("R,in_f,out_f", [(1600, 64, 256), (1600, 256, 64), (1600, 10, 64), (32, 64, 1), (37, 50, 33),
                                          (1600, 128, 384), (3200, 32, 96), (800, 512, 128), (130, 36, 40)])
@pytest.mark.parametrize("bias", [True, False])
@pytest.mark.parametrize("tile", [16, 32])
def test_small_linear_backward_matches_f32_reference(R, in_f, out_f, bias, tile, monkeypatch):
    """One launch gives dX, dW and db of a small f32 layer (dimensions that are not multiples of 4 are padded with zeros
    before it and sliced after); compared with a float64 product of the same operands."""
    from mitune.options import OPTIONS
    from mitune.ops.linear import linear
    monkeypatch.setattr(OPTIONS, "small_linear_tile", tile)
    g = torch.Generator(device="cuda").manual_seed(R + in_f + out_f)
    x = torch.randn(R // 4, 4, in_f, device="cuda", generator=g).requires_grad_() if R % 4 == 0 else \
        torch.randn(R, in_f, device="cuda", generator=g).requires_grad_()
    w = (torch.randn(out_f, in_f, device="cuda", generator=g) * in_f ** -0.5).requires_grad_()
    b = torch.randn(out_f, device="cuda", generator=g).requires_grad_() if bias else None
    y = linear(x, w, b)
    assert "SmallLinear" in type(_producer(y)).__name__
    ref = x.detach().double() @ w.detach().double().t() + (b.detach().double() if bias else 0)
    assert torch.allclose(y.double(), ref, rtol=1e-4, atol=1e-4)
    dy = torch.randn(y.shape, device="cuda", generator=g)
    for rep in range(2):
        x.grad = w.grad = None
        if b is not None:
            b.grad = None
        linear(x, w, b).backward(dy)
        x2, w2, dy2 = x.detach().double().reshape(R, in_f), w.detach().double(), dy.double().reshape(R, out_f)
        assert torch.allclose(x.grad.double().reshape(R, in_f), dy2 @ w2, rtol=1e-4, atol=1e-4)
        assert torch.allclose(w.grad.double(), dy2.t() @ x2, rtol=1e-4, atol=1e-3)
        if b is not None:
            assert torch.allclose(b.grad.double(), dy2.sum(0), rtol=1e-4, atol=1e-3)
    first = w.grad.clone()
    w.grad = None
    linear(x, w, b).backward(dy)
    assert torch.equal(w.grad, first)                     # ordered reduction: bitwise repeatable
    xin = x.detach()                                      # an input that needs no gradient: no dX tiles are launched
    w.grad = None
    linear(xin, w, b).backward(dy)
    assert torch.equal(w.grad, first)


def test_small_layers_serve_training_and_validation_passes_with_the_same_forward():
    from mitune.ops.linear import linear
    x = torch.randn(4, 50, 32, device="cuda", requires_grad=True)
    w = torch.randn(48, 32, device="cuda", requires_grad=True)
    b = torch.randn(48, device="cuda", requires_grad=True)
    y = linear(x, w, b)
    assert y.shape == (4, 50, 48) and "SmallLinear" in type(y.grad_fn).__name__
    ref = (x.detach().double() @ w.detach().double().t() + b.detach().double()).float()
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        z = linear(x, w, b)
    assert z.grad_fn is None and torch.equal(z, y.detach())


def test_operands_the_kernels_cannot_take_raise_instead_of_reaching_a_library_gemm():
    from mitune.ops.linear import linear
    x, w = torch.randn(8, 16, device="cuda", dtype=torch.float64), torch.randn(4, 16, device="cuda", dtype=torch.float64)
    with pytest.raises(TypeError):
        linear(x, w)
    with torch.autocast("cuda", dtype=torch.float16), pytest.raises(TypeError):
        linear(x.float(), w.float())


@pytest.mark.parametrize("rows,in_f,out_f", [(300, 81, 70), (5000, 20, 1), (64, 9, 3)])
def test_unaligned_bf16_layers_are_padded_onto_the_tile_kernel(rows, in_f, out_f):
    from mitune.ops.linear import DenseBF16, linear
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(rows, in_f, device="cuda", generator=g).requires_grad_(True)
    w = (torch.randn(out_f, in_f, device="cuda", generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(out_f, device="cuda", generator=g).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = linear(x, w, b)
    assert y.shape == (rows, out_f)
    if out_f > 1:
        assert isinstance(_producer(y), DenseBF16._backward_cls)
    y.float().sum().backward()
    xr = x.detach().to(torch.bfloat16).double()
    wr = w.detach().to(torch.bfloat16).double() if out_f > 1 else w.detach().double()
    xr = xr if out_f > 1 else x.detach().double()
    torch.testing.assert_close(y.double(), xr @ wr.t() + b.detach().double(), rtol=2e-2, atol=2e-2)
    assert w.grad.shape == (out_f, in_f) and b.grad.shape == (out_f,)
    torch.testing.assert_close(w.grad.double(), xr.sum(0).expand(out_f, in_f), rtol=2e-2, atol=2e-2 * float(xr.sum(0).abs().max()))
    torch.testing.assert_close(b.grad.double(), torch.full((out_f,), float(rows), device="cuda", dtype=torch.float64), rtol=1e-2, atol=1.0)


def test_small_linear_is_not_used_under_autocast_or_for_large_layers():
    from mitune.ops.linear import linear
    x = torch.randn(64, 32, device="cuda", requires_grad=True)
    w = torch.randn(16, 32, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert "SmallLinear" not in type(linear(x, w).grad_fn).__name__
    xl = torch.randn(40000, 32, device="cuda", requires_grad=True)
    assert "SmallLinear" not in type(linear(xl, w).grad_fn).__name__
    wl = torch.randn(1024, 256, device="cuda", requires_grad=True)
    assert "SmallLinear" not in type(linear(torch.randn(64, 256, device="cuda"), wl).grad_fn).__name__
    assert "SmallLinear" in type(linear(x, w).grad_fn).__name__
    with torch.no_grad():
        assert linear(x, w).grad_fn is None


# ---------------------------------------------------------------------------------------------------------------
# Dense layers on the package's own GEMM kernels (no BLAS library call in a training step)
@pytest.mark.parametrize("rows,in_f,out_f", [(66560, 192, 576), (1024, 192, 256), (4096, 256, 192), (300, 64, 72), (66560, 256, 192)])
def test_dense_bf16_matches_library_linear(rows, in_f, out_f):
    from mitune.ops.linear import DenseBF16, linear
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(rows, in_f, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(out_f, in_f, device="cuda", generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(out_f, device="cuda", generator=g).requires_grad_(True)
    dy = torch.randn(rows, out_f, device="cuda", generator=g).to(torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = linear(x.view(rows // 4, 4, in_f) if rows % 4 == 0 else x, w, b)
    assert y.dtype == torch.bfloat16 and isinstance(y.grad_fn, DenseBF16._backward_cls)
    y.backward(dy.view_as(y))
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)      # the kernel multiplies bf16-rounded weights
    br = b.detach().clone().requires_grad_(True)
    yr = xr @ wr.t() + br
    yr.backward(dy.float())
    torch.testing.assert_close(y.float().reshape(rows, out_f), yr, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(x.grad.float(), xr.grad, rtol=2e-2, atol=2e-2)
    scale = float(wr.grad.abs().max())
    torch.testing.assert_close(w.grad, wr.grad, rtol=1e-2, atol=1e-3 * scale)
    torch.testing.assert_close(b.grad, br.grad, rtol=1e-2, atol=1e-3 * float(br.grad.abs().max()))


@pytest.mark.parametrize("rows,in_f,out_f", [(4200, 520, 776), (2048, 1024, 512), (520, 264, 1032)])
def test_eight_wave_256_tiles_equal_the_128_tiles_bitwise(rows, in_f, out_f):
    """A large single GEMM runs on 256 x 256 tiles with eight waves; an element's contraction is summed in the same order
    as on 128-tiles, so forward output, input gradient, weight and bias gradient are bit-identical (edges included: none
    of the three shapes is a multiple of 256 everywhere) -- and match an f32 reference."""
    from mitune.ops import lib as L
    from mitune.ops.linear import linear
    so = L.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    x0 = torch.randn(rows, in_f, device="cuda", generator=g).to(torch.bfloat16)
    w0 = torch.randn(out_f, in_f, device="cuda", generator=g) * 0.05
    b0 = torch.randn(out_f, device="cuda", generator=g)
    dy = torch.randn(rows, out_f, device="cuda", generator=g).to(torch.bfloat16)
    got = {}
    before, sched = so.mt_gemm_p8_min_tiles(-1), so.mt_gemm_p8_option(0, -2)
    try:
        # 128-tiles; 256-tiles with the schedule chosen by mode, with two barriers per K-tile, with one
        for key, limit, option in ((0, 0, -1), (1, 1, -1), (2, 1, 0), (3, 1, 1)):
            so.mt_gemm_p8_min_tiles(limit)
            so.mt_gemm_p8_option(0, option)
            x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = linear(x, w, b)
            y.backward(dy)
            torch.cuda.synchronize()
            got[key] = (y.detach(), x.grad, w.grad, b.grad)
    finally:
        so.mt_gemm_p8_min_tiles(before)
        so.mt_gemm_p8_option(0, sched)
    for key in (1, 2, 3):
        for a, c in zip(got[0], got[key]):
            assert torch.equal(a, c), key
    xr, wr = x0.float().requires_grad_(True), w0.to(torch.bfloat16).float().requires_grad_(True)
    yr = xr @ wr.t() + b0
    yr.backward(dy.float())
    torch.testing.assert_close(got[1][0].float(), yr, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(got[1][1].float(), xr.grad, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(got[1][2], wr.grad, rtol=1e-2, atol=1e-3 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("rows,in_f,out_f", [(40000, 128, 384), (3072, 512, 2048), (3072, 2048, 512), (70000, 64, 64), (40001, 68, 36)])
def test_dense_f32_matches_float64_reference(rows, in_f, out_f):
    from mitune.ops.linear import DenseF32, linear
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(rows, in_f, device="cuda", generator=g).requires_grad_(True)
    w = (torch.randn(out_f, in_f, device="cuda", generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(out_f, device="cuda", generator=g).requires_grad_(True)
    dy = torch.randn(rows, out_f, device="cuda", generator=g)
    y = linear(x, w, b)
    assert isinstance(y.grad_fn, DenseF32._backward_cls)
    y.backward(dy)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = xr @ wr.t() + br
    yr.backward(dy.double())
    torch.testing.assert_close(y.double(), yr, rtol=1e-5, atol=5e-5)          # f32 accumulation over up to 2048 terms
    torch.testing.assert_close(x.grad.double(), xr.grad, rtol=1e-5, atol=5e-5)
    torch.testing.assert_close(w.grad.double(), wr.grad, rtol=1e-5, atol=1e-5 * float(wr.grad.abs().max()))
    torch.testing.assert_close(b.grad.double(), br.grad, rtol=1e-5, atol=1e-5 * float(br.grad.abs().max()))


def test_small_unaligned_f32_layer_stays_on_own_kernels():
    """in_features = 10 (the sample model's input projection): rows are not 16-byte aligned, so the small-layer kernels
    take forward and backward; the result still matches a float64 reference."""
    from mitune.ops.linear import SmallLinear, linear
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(32, 50, 10, device="cuda", generator=g)
    w = (torch.randn(64, 10, device="cuda", generator=g) * 0.3).requires_grad_(True)
    b = torch.randn(64, device="cuda", generator=g).requires_grad_(True)
    y = linear(x, w, b)
    assert isinstance(y.grad_fn, SmallLinear._backward_cls)
    y.sum().backward()
    ref = x.double() @ w.detach().double().t() + b.detach().double()
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(w.grad.double(), x.double().reshape(-1, 10).sum(0).expand(64, 10), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("x_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,d", [(65 * 33, 192), (1000, 64), (37, 256)])
def test_mixed_precision_layer_norm_matches_f32_reference(rows, d, x_dtype):
    """bf16 (or f32) branch input, f32 residual, two outputs (f32 + bf16 copy), two incoming gradients added in-kernel."""
    from mitune.ops.layernorm import residual_layer_norm_mixed
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(rows, d, device="cuda", generator=g).to(x_dtype).requires_grad_(True)
    res = torch.randn(rows, d, device="cuda", generator=g).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(d, device="cuda", generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(d, device="cuda", generator=g)).requires_grad_(True)
    gy = torch.randn(rows, d, device="cuda", generator=g)
    g16 = torch.randn(rows, d, device="cuda", generator=g).to(torch.bfloat16)
    y, y16 = residual_layer_norm_mixed(x, res, w, b, 1e-5, 0.0, want_bf16=True)
    assert y.dtype == torch.float32 and y16.dtype == torch.bfloat16
    ((y * gy).sum() + (y16.float() * g16.float()).sum()).backward()
    xr, rr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, res, w, b))
    yr = torch.nn.functional.layer_norm(rr + xr, (d,), wr, br, 1e-5)
    (yr * (gy.double() + g16.double())).sum().backward()
    torch.testing.assert_close(y.double(), yr, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(y16.double(), yr, rtol=1e-2, atol=1e-2)
    assert x.grad.dtype == x_dtype
    tol = 2e-2 if x_dtype == torch.bfloat16 else 1e-4
    torch.testing.assert_close(x.grad.double(), xr.grad, rtol=tol, atol=tol)
    torch.testing.assert_close(res.grad.double(), rr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(w.grad.double(), wr.grad, rtol=1e-3, atol=1e-3 * float(wr.grad.abs().max()))
    torch.testing.assert_close(b.grad.double(), br.grad, rtol=1e-3, atol=1e-3 * float(br.grad.abs().max()))
    # one output only, dropout on: the mask of the backward pass is the forward's
    x2 = torch.randn(rows, d, device="cuda", generator=g).to(x_dtype).requires_grad_(True)
    zero_res = torch.zeros(rows, d, device="cuda")
    y2, none = residual_layer_norm_mixed(x2, zero_res, torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"), 1e-5, 0.5,
                                         want_bf16=False)
    assert none is None and torch.isfinite(y2).all()
    y2.sum().backward()
    assert torch.isfinite(x2.grad.float()).all()


def test_reglu_dropout_matches_reference_and_reuses_its_mask():
    from mitune.ops.elementwise import reglu_dropout
    g = torch.Generator(device="cuda").manual_seed(6)
    rows, h = 65 * 40, 256
    u = torch.randn(rows, 2 * h, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    dout = torch.randn(rows, h, device="cuda", generator=g).to(torch.bfloat16)
    out = reglu_dropout(u, 0.0, True)
    out.backward(dout)
    ur = u.detach().float().requires_grad_(True)
    a, gg = ur.chunk(2, -1)
    ref = a * torch.relu(gg)
    ref.backward(dout.float())
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(u.grad.float(), ur.grad, rtol=1e-2, atol=1e-2)
    # dropout: about half the positive-gate outputs survive, scaled by 2; the gradient is zero exactly where the output is
    u2 = (torch.rand(rows, 2 * h, device="cuda", generator=g) + 0.5).to(torch.bfloat16).requires_grad_(True)
    o2 = reglu_dropout(u2, 0.5, True)
    kept = (o2 != 0).float().mean().item()
    assert 0.47 < kept < 0.53
    a2, g2 = u2.detach().float().chunk(2, -1)
    torch.testing.assert_close(o2.float()[o2 != 0], (2 * a2 * g2)[o2 != 0], rtol=2e-2, atol=2e-2)
    o2.backward(torch.ones_like(o2))
    da = u2.grad[:, :h]
    assert torch.equal(da != 0, o2 != 0)
    assert torch.equal(reglu_dropout(u2, 0.5, False), reglu_dropout(u2, 0.0, True))       # evaluation: no mask


def test_big_unaligned_f32_layer_is_padded_onto_the_tile_kernels():
    """81 input features at 24 576 rows: too big for the small-layer kernels, rows not 16-byte aligned."""
    from mitune.ops.linear import DenseF32, linear
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(256, 96, 81, device="cuda", generator=g)
    w = (torch.randn(512, 81, device="cuda", generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(512, device="cuda", generator=g).requires_grad_(True)
    y = linear(x, w, b)
    y.backward(torch.ones_like(y))
    ref = x.double() @ w.detach().double().t() + b.detach().double()
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-4)
    assert w.grad.shape == (512, 81)
    torch.testing.assert_close(w.grad.double(), x.double().reshape(-1, 81).sum(0).expand(512, 81), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(b.grad.double(), torch.full((512,), 256.0 * 96, device="cuda", dtype=torch.float64))


@pytest.mark.parametrize("B,S,h,dk,dv", [(3, 65, 8, 24, 24), (2, 96, 2, 64, 32), (2, 33, 3, 6, 20)])
def test_bf16_attention_gradients_written_by_the_kernel_equal_a_cast_of_its_f32_gradients(B, S, h, dk, dv):
    """Mode 2 of the backward launcher (bf16 operands, bf16 gradients) rounds exactly what mode 0 returns in f32."""
    import math
    from mitune.ops import lib as L
    so = L.load()
    g = torch.Generator(device="cuda").manual_seed(31)
    qk = torch.randn(B, S, 2 * h * dk, device="cuda", generator=g).bfloat16()
    v = torch.randn(B, S, h * dv, device="cuda", generator=g).bfloat16()
    do = torch.randn(B, S, h * dv, device="cuda", generator=g).bfloat16()
    out = torch.empty(B, S, h * dv, device="cuda", dtype=torch.bfloat16)
    stat = torch.empty(B * h, S, device="cuda", dtype=torch.float32)
    scale = 1.0 / math.sqrt(dk)
    stream = L.raw_stream(qk.device)
    L.check(so.mt_attn_fwd_strided_launch(qk.data_ptr(), qk.data_ptr() + h * dk * 2, v.data_ptr(), None, out.data_ptr(),
                                          stat.data_ptr(), B * h, S, dk, dv, scale, 1.0, 0, h, 2 * h * dk, h * dv, h * dv,
                                          0, None, 0.0, stream), "fwd")
    grads = {}
    for mode, dt, es in ((0, torch.float32, 4), (2, torch.bfloat16, 2)):
        dqk = torch.zeros(B, S, 2 * h * dk, device="cuda", dtype=dt)
        dv_ = torch.zeros(B, S, h * dv, device="cuda", dtype=dt)
        L.check(so.mt_attn_bwd_strided_launch(qk.data_ptr(), qk.data_ptr() + h * dk * 2, v.data_ptr(), do.data_ptr(), None,
                                              stat.data_ptr(), dqk.data_ptr(), dqk.data_ptr() + h * dk * es, dv_.data_ptr(),
                                              B * h, S, dk, dv, scale, 1.0, 0, h, 2 * h * dk, h * dv, h * dv, mode, None,
                                              0.0, stream), "bwd")
        torch.cuda.synchronize()
        grads[mode] = (dqk, dv_)
    assert torch.equal(grads[0][0].bfloat16(), grads[2][0])
    assert torch.equal(grads[0][1].bfloat16(), grads[2][1])
    assert grads[2][0].float().abs().sum() > 0


@pytest.mark.parametrize("d,mixed", [(64, False), (512, False), (192, True)])
def test_layernorm_backward_over_more_rows_than_one_block_per_group(d, mixed):
    """40 000 rows are 1 250 groups of 32: past 1 024 a block takes several groups and writes one partial of the weight
    and bias gradients.  All three backward kernels (narrow, wide, mixed precision) against a float64 reference."""
    from mitune.ops.layernorm import residual_layer_norm_reference
    from mitune.ops.layernorm_hip import FusedResidualLayerNorm, MixedResidualLayerNorm
    g = torch.Generator(device="cuda").manual_seed(d + 1)
    M = 40000
    x = torch.randn(M, d, device="cuda", generator=g)
    if mixed:
        x = x.bfloat16()
    x.requires_grad_()
    r = torch.randn(M, d, device="cuda", generator=g, requires_grad=True)
    w = torch.randn(d, device="cuda", generator=g, requires_grad=True)
    b = torch.randn(d, device="cuda", generator=g, requires_grad=True)
    dy = torch.randn(M, d, device="cuda", generator=g)
    if mixed:
        y = MixedResidualLayerNorm.apply(x, r, w, b, 1e-5, 0.0, None, False)
    else:
        y = FusedResidualLayerNorm.apply(x, r, w, b, 1e-5, None, 0.0)
    y.backward(dy)
    torch.cuda.synchronize()
    x64, r64, w64, b64 = (t.detach().double().requires_grad_() for t in (x, r, w, b))
    yr = residual_layer_norm_reference(x64, r64, w64, b64, 1e-5, None, 0.0)
    yr.backward(dy.double())
    torch.testing.assert_close(y.detach().double(), yr.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(r.grad.double(), r64.grad, rtol=1e-4, atol=1e-4)
    xtol = 2e-2 if mixed else 1e-4          # the mixed kernel returns the gradient of a bf16 input in bf16
    torch.testing.assert_close(x.grad.double(), x64.grad, rtol=xtol, atol=xtol)
    for got, ref in ((w.grad, w64.grad), (b.grad, b64.grad)):       # sums of 40 000 terms of order one
        torch.testing.assert_close(got.double(), ref, rtol=1e-4, atol=5e-3)


@pytest.mark.parametrize("B,F,d", [(64, 64, 192), (37, 5, 8), (1024, 12, 64)])
@pytest.mark.parametrize("two_outputs", [False, True])
def test_feature_tokens_match_tensor_ops(B, F, d, two_outputs):
    """Tokenizer kernels (forward with the bf16 copy, backward over both outputs) against x * W + b | cls in f32."""
    from mitune.ops.tokens import FeatureTokens, feature_tokens_reference
    g = torch.Generator(device="cuda").manual_seed(B + F)
    x = torch.randn(B, F, device="cuda", generator=g)
    W, b = (torch.randn(F, d, device="cuda", generator=g).requires_grad_(True) for _ in range(2))
    cls = torch.randn(1, 1, d, device="cuda", generator=g).requires_grad_(True)
    g32 = torch.randn(B, F + 1, d, device="cuda", generator=g)
    g16 = torch.randn(B, F + 1, d, device="cuda", generator=g).to(torch.bfloat16)
    if two_outputs:
        out, out16 = FeatureTokens.apply(x, W, b, cls, True)
        torch.autograd.backward([out, out16], [g32, g16])
    else:
        out = FeatureTokens.apply(x, W, b, cls, False)
        out.backward(g32)
    got = [out.detach(), W.grad.clone(), b.grad.clone(), cls.grad.clone()]
    W.grad = b.grad = cls.grad = None
    ref = feature_tokens_reference(x, W, b, cls)
    ref.backward(g32 + g16.float() if two_outputs else g32)
    assert torch.equal(got[0], ref.detach())
    if two_outputs:
        assert torch.equal(out16, ref.detach().to(torch.bfloat16))
    for name, a, e in zip(("dW", "dbias", "dcls"), got[1:], (W.grad, b.grad, cls.grad)):
        assert a.shape == e.shape, name
        assert (a - e).abs().max().item() <= 2e-5 * (e.abs().max().item() + 1.0) * max(1.0, B ** 0.5 / 8), name


@pytest.mark.parametrize("rows,cols", [(694, 384), (1, 7), (33, 16), (100, 130)])
def test_column_sums(rows, cols):
    from mitune.ops import lib as L
    so = L.load()
    src = torch.randn(rows, cols, device="cuda")
    out = torch.empty(cols, device="cuda")
    L.check(so.mt_colsum_launch(src.data_ptr(), out.data_ptr(), rows, cols, L.raw_stream(src.device)), "mt_colsum_launch")
    ref = src.double().sum(0)
    assert (out.double() - ref).abs().max().item() < 1e-4 * max(1.0, rows ** 0.5)
    again = torch.empty(cols, device="cuda")
    L.check(so.mt_colsum_launch(src.data_ptr(), again.data_ptr(), rows, cols, L.raw_stream(src.device)), "mt_colsum_launch")
    assert torch.equal(out, again)


@pytest.mark.parametrize("B,d", [(1024, 192), (7, 16), (130, 512), (33, 100)])
def test_summary_token_head_matches_tensor_ops(B, d):
    from mitune.ops.tokens import ClsHead, cls_head_reference
    g = torch.Generator(device="cuda").manual_seed(B + d)
    x = torch.randn(B, d, device="cuda", generator=g).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(d, device="cuda", generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(d, device="cuda", generator=g)).requires_grad_(True)
    w = torch.randn(1, d, device="cuda", generator=g).requires_grad_(True)
    b0 = torch.randn(1, device="cuda", generator=g).requires_grad_(True)
    dp = torch.randn(B, device="cuda", generator=g)
    leaves = [x, gamma, beta, w, b0]
    out = ClsHead.apply(x, gamma, beta, w, b0, 1e-5)
    got = [out.detach()] + list(torch.autograd.grad(out, leaves, dp))
    ref = cls_head_reference(*[t.double() for t in leaves])
    exp = [ref.detach()] + list(torch.autograd.grad(ref, leaves, dp.double()))
    for name, a, e in zip(("pred", "dx", "dgamma", "dbeta", "dw", "db0"), got, exp):
        assert a.shape == e.shape, name
        assert (a.double() - e).abs().max().item() <= 3e-5 * (e.abs().max().item() + 1.0) * max(1.0, B ** 0.5 / 8), name
