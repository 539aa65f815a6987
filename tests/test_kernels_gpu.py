"""Numerics of every hand-written HIP kernel against a plain PyTorch fp32 reference of the same op.

bf16 operands are generated as bf16-representable values, so the only difference between kernel and
reference is the f32 accumulation order (and the final bf16 rounding of bf16 outputs).
"""
import ctypes as C
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from mitune.ops import lib as L
    return L, L.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _up(arr: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).cuda()


def _tiles(L, dims):
    """dims: list of (group, I, J) -> plain (unbalanced) tile list incl. one padding entry."""
    rows = [(-1, 0, 0)]
    edge = 256 if NSTAGE in (256, 258) else 128        # 258: 256-tiles on eight waves
    for g, i, j in dims:
        for ti in range((i + edge - 1) // edge):
            for tj in range((j + edge - 1) // edge):
                rows.append((g, ti, tj))
    return np.array(rows, dtype=L.TILE_DTYPE)


def _rand_bf16(*shape, scale=1.0, gen=None):
    return (torch.randn(*shape, generator=gen, device="cuda") * scale).to(torch.bfloat16)


NSTAGE = 0          # 0: launcher picks the pipeline depth; the parametrised test below forces both forms


@pytest.fixture(params=[2, 4, 256, 258])
def pipeline_code(request, monkeypatch):
    """The bf16 grouped GEMM tests run once per pipeline form of the launcher (two / four stages on 128-tiles, 256-tiles
    on four / eight waves); every other test in this file runs once."""
    monkeypatch.setattr(sys.modules[__name__], "NSTAGE", request.param)


SHAPES = [(256, 128, 64), (512, 512, 512), (200, 72, 40), (128, 64, 8), (384, 256, 192), (130, 136, 520)]


@pytest.mark.usefixtures("pipeline_code")
@pytest.mark.parametrize("relu", [False, True])
def test_gemm_forward_nt(lib, relu):
    L, so = lib
    g = torch.Generator(device="cuda").manual_seed(1)
    groups = np.zeros(len(SHAPES) + 1, dtype=L.GEMM_GROUP_DTYPE)
    keep, dims = [], []
    for gi, (m, n, k) in enumerate(SHAPES):
        ldk, ldn = (k + 7) // 8 * 8, (n + 7) // 8 * 8
        x = torch.zeros(m, ldk, device="cuda", dtype=torch.bfloat16); x[:, :k] = _rand_bf16(m, k, gen=g)
        w = torch.zeros(n, ldk, device="cuda", dtype=torch.bfloat16); w[:, :k] = _rand_bf16(n, k, scale=0.2, gen=g)
        b = torch.randn(n, device="cuda", generator=g)
        y = torch.full((m, ldn), 7.0, device="cuda", dtype=torch.bfloat16)
        groups[gi] = L.gemm_group(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, m, n, k, ldk, ldk, ldn, 0, 1)
        keep.append((x, w, b, y)); dims.append((gi, m, n))
    groups[len(SHAPES)]["active_steps"] = 0
    tiles = _tiles(L, dims)
    gt, tt = _up(groups), _up(tiles)
    L.check(so.mt_gemm_launch(0, gt.data_ptr(), tt.data_ptr(), len(tiles), 0, int(relu), 0, 0, 1, 0, NSTAGE, _stream()), "gemm")
    torch.cuda.synchronize()
    for (m, n, k), (x, w, b, y) in zip(SHAPES, keep):
        ref = x[:, :k].float() @ w[:, :k].float().t() + b
        if relu:
            ref = ref.relu()
        got = y[:, :n].float()
        torch.testing.assert_close(got, ref, rtol=1e-2, atol=1e-2)
        ldn = y.shape[1]
        if ldn > n:     # pad columns inside the last 8-chunk are written as zeros
            assert torch.all(y[:, n:] == 0)


@pytest.mark.usefixtures("pipeline_code")
def test_gemm_forward_identity_asymmetric(lib):
    """A = I with an asymmetric B: catches a transposed or permuted accumulator write."""
    L, so = lib
    n = 128
    x = torch.eye(n, device="cuda", dtype=torch.bfloat16)
    w = (torch.arange(n * n, device="cuda").reshape(n, n) % 251).float().to(torch.bfloat16)   # W[j][k]
    b = torch.zeros(n, device="cuda")
    y = torch.zeros(n, n, device="cuda", dtype=torch.bfloat16)
    groups = np.zeros(1, dtype=L.GEMM_GROUP_DTYPE)
    groups[0] = L.gemm_group(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, n, n, n, n, n, n, 0, 1)
    tiles = _tiles(L, [(0, n, n)])
    gt, tt = _up(groups), _up(tiles)
    L.check(so.mt_gemm_launch(0, gt.data_ptr(), tt.data_ptr(), len(tiles), 0, 0, 0, 0, 1, 0, NSTAGE, _stream()), "gemm")
    torch.cuda.synchronize()
    assert torch.equal(y.float(), w.float().t())


@pytest.mark.usefixtures("pipeline_code")
@pytest.mark.parametrize("mask", [False, True])
def test_gemm_dgrad_nn(lib, mask):
    L, so = lib
    g = torch.Generator(device="cuda").manual_seed(2)
    groups = np.zeros(len(SHAPES), dtype=L.GEMM_GROUP_DTYPE)
    keep, dims = [], []
    for gi, (m, n, k) in enumerate(SHAPES):      # dX[m,k] = dY[m,n] @ W[n,k]
        ldk, ldn = (k + 7) // 8 * 8, (n + 7) // 8 * 8
        dy = torch.zeros(m, ldn, device="cuda", dtype=torch.bfloat16); dy[:, :n] = _rand_bf16(m, n, gen=g)
        w = torch.zeros(n, ldk, device="cuda", dtype=torch.bfloat16); w[:, :k] = _rand_bf16(n, k, scale=0.2, gen=g)
        h = torch.zeros(m, ldk, device="cuda", dtype=torch.bfloat16); h[:, :k] = _rand_bf16(m, k, gen=g).relu()
        dx = torch.full((m, ldk), 7.0, device="cuda", dtype=torch.bfloat16)
        groups[gi] = L.gemm_group(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), h.data_ptr(), 0, m, k, n, ldn, ldk, ldk, ldk, 1)
        keep.append((dy, w, h, dx)); dims.append((gi, m, k))
    tiles = _tiles(L, dims)
    gt, tt = _up(groups), _up(tiles)
    L.check(so.mt_gemm_launch(1, gt.data_ptr(), tt.data_ptr(), len(tiles), 0, int(mask), 0, 0, 1, 0, NSTAGE, _stream()), "gemm")
    torch.cuda.synchronize()
    for (m, n, k), (dy, w, h, dx) in zip(SHAPES, keep):
        ref = dy[:, :n].float() @ w[:, :k].float()
        if mask:
            ref = ref * (h[:, :k].float() > 0)
        torch.testing.assert_close(dx[:, :k].float(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.usefixtures("pipeline_code")
@pytest.mark.parametrize("ksplits", [1, 3])
def test_gemm_wgrad_tn(lib, ksplits):
    L, so = lib
    g = torch.Generator(device="cuda").manual_seed(3)
    groups = np.zeros(len(SHAPES), dtype=L.GEMM_GROUP_DTYPE)
    keep, dims = [], []
    for gi, (m, n, k) in enumerate(SHAPES):      # dW[n,k] = dY[m,n]^T @ X[m,k];  db[n] = sum_m dY[m,n]
        ldk, ldn = (k + 7) // 8 * 8, (n + 7) // 8 * 8
        dy = torch.zeros(m, ldn, device="cuda", dtype=torch.bfloat16); dy[:, :n] = _rand_bf16(m, n, scale=0.1, gen=g)
        x = torch.zeros(m, ldk, device="cuda", dtype=torch.bfloat16); x[:, :k] = _rand_bf16(m, k, gen=g)
        # one arena per group: [ksplits][dW | db]; the split stride is the same for every group
        arena = torch.full((ksplits, 600 * 600), 5.0, device="cuda")
        dw_all = arena[:, :n * ldk].view(ksplits, n, ldk)
        dw_all.zero_()
        db_all = arena[:, 590 * 600:590 * 600 + n]
        groups[gi] = L.gemm_group(dy.data_ptr(), x.data_ptr(), arena.data_ptr(), 0, db_all.data_ptr(), n, k, m, ldn, ldk, ldk, 0, 1)
        keep.append((dy, x, dw_all, db_all)); dims.append((gi, n, k))
    tiles = _tiles(L, dims)
    if ksplits > 1:
        base = tiles[1:]
        reps = []
        for sp in range(ksplits):
            t = base.copy(); t["ti"] = t["ti"] | (sp << 8); reps.append(t)
        tiles = np.concatenate([tiles[:1]] + reps)
    gt, tt = _up(groups), _up(tiles)
    L.check(so.mt_gemm_launch(2, gt.data_ptr(), tt.data_ptr(), len(tiles), 0, 1, 0, 0, ksplits, 600 * 600, NSTAGE, _stream()),
            "gemm")
    torch.cuda.synchronize()
    for (m, n, k), (dy, x, dw_all, db_all) in zip(SHAPES, keep):
        dw, db = dw_all.sum(0), db_all.sum(0)
        ref = dy[:, :n].float().t() @ x[:, :k].float()
        torch.testing.assert_close(dw[:, :k], ref, rtol=2e-3, atol=2e-3)
        torch.testing.assert_close(db, dy[:, :n].float().sum(0), rtol=2e-3, atol=2e-3)
        if dw.shape[1] > k:
            assert torch.all(dw[:, k:] == 0)


def test_gather_matches_numpy_permutation(lib):
    L, so = lib
    from mitune.packed.permutation import half_bits, permute, round_keys
    n, F, Fpad, B = 5000, 10, 16, 96
    X = torch.randn(n, F, device="cuda"); Y = torch.randn(n, device="cuda")
    slots = np.zeros(3, dtype=L.GATHER_DTYPE)
    outs = []
    for s in range(3):
        x0 = torch.full((B, Fpad), 9.0, device="cuda", dtype=torch.bfloat16); y = torch.zeros(B, device="cuda")
        keys = round_keys(11, 100 + s, 2)
        slots[s] = (x0.data_ptr(), y.data_ptr(), keys, B - 8 * s, 50)
        outs.append((x0, y, keys))
    st = _up(slots)
    ids = torch.arange(3, dtype=torch.int32, device="cuda")
    step = 7
    L.check(so.mt_gather_launch(st.data_ptr(), ids.data_ptr(), 3, B, X.data_ptr(), Y.data_ptr(), n, F, Fpad,
                                half_bits(n), step, _stream()), "gather")
    torch.cuda.synchronize()
    for s, (x0, y, keys) in enumerate(outs):
        b = B - 8 * s
        idx = torch.from_numpy(permute(np.arange(step * b, step * b + b, dtype=np.uint32), n, keys)).cuda()
        assert torch.equal(x0[:b, :F], X[idx].to(torch.bfloat16))
        assert torch.all(x0[:b, F:] == 0)
        assert torch.equal(y[:b], Y[idx])


@pytest.mark.parametrize("loss,code", [("mse", 0), ("mae", 1), ("huber", 2), ("mape", 3)])
def test_head_loss(lib, loss, code):
    L, so = lib
    from mitune.train.losses import elementwise_loss
    g = torch.Generator(device="cuda").manual_seed(4)
    cfgs = [(256, 512), (100, 64), (96, 200)]
    slots = np.zeros(len(cfgs), dtype=L.HEAD_DTYPE)
    keep = []
    for s, (B, W) in enumerate(cfgs):
        ld = (W + 7) // 8 * 8
        H = torch.zeros(B, ld, device="cuda", dtype=torch.bfloat16); H[:, :W] = _rand_bf16(B, W, gen=g).relu()
        dH = torch.zeros(B, ld, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(W, device="cuda", generator=g) * 0.1; b = torch.randn(1, device="cuda", generator=g)
        gw = torch.zeros(W, device="cuda"); gb = torch.zeros(1, device="cuda")
        y = torch.randn(B, device="cuda", generator=g) + 2.0
        acc = torch.full((1,), 1.0, device="cuda"); met = torch.zeros(3, device="cuda", dtype=torch.float64)
        slots[s] = (H.data_ptr(), dH.data_ptr(), w.data_ptr(), b.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                    y.data_ptr(), acc.data_ptr(), met.data_ptr(), B, W, ld, code, 1, 0)
        keep.append((H, dH, w, b, gw, gb, y, acc, met))
    st = _up(slots)
    ids = torch.arange(len(cfgs), dtype=torch.int32, device="cuda")
    scratch = torch.zeros(int(so.mt_head_scratch_floats(len(cfgs), 256, 512)), device="cuda")
    L.check(so.mt_head_loss_launch(st.data_ptr(), ids.data_ptr(), len(cfgs), 256, 512, 0, scratch.data_ptr(), _stream()),
            "head_loss")
    L.check(so.mt_head_eval_launch(st.data_ptr(), ids.data_ptr(), len(cfgs), 512, 0, 0, 256, scratch.data_ptr(),
                                   _stream()), "head_eval")
    torch.cuda.synchronize()
    for (B, W), (H, dH, w, b, gw, gb, y, acc, met) in zip(cfgs, keep):
        h = H[:, :W].float().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = h @ wr + br
        l = elementwise_loss(loss, out, y).mean()
        l.backward()
        torch.testing.assert_close(acc, torch.tensor([1.0 + l.item()], device="cuda"), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(gw, wr.grad, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(gb, br.grad, rtol=1e-3, atol=1e-4)
        ref_dh = (h.grad * (h > 0)).detach()
        torch.testing.assert_close(dH[:, :W].float(), ref_dh, rtol=1e-2, atol=1e-5)
        d = (out - y).detach()
        ref_m = torch.stack([elementwise_loss(loss, out, y).sum(), (d * d).sum(), (d.abs() / (y.abs() + 1e-8)).sum()])
        torch.testing.assert_close(met.float(), ref_m.detach().float(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("name,code", [("adam", 0), ("adamw", 1), ("sgd", 2), ("rmsprop", 3)])
@pytest.mark.parametrize("clip,gsplits", [(0.0, 1), (0.5, 1), (0.5, 3)])
def test_fused_optimizer_matches_torch(lib, name, code, clip, gsplits):
    L, so = lib
    from mitune.train.optim import build_optimizer
    g = torch.Generator(device="cuda").manual_seed(5)
    sizes, stride = [5000, 333, 2048], 6144
    T = len(sizes)
    P = torch.zeros(T * stride, device="cuda"); M = torch.zeros_like(P); V = torch.zeros_like(P)
    G = torch.zeros(gsplits * T * stride, device="cuda")        # gradient slabs, summed by the kernel
    Pb = torch.zeros(T * stride, device="cuda", dtype=torch.bfloat16)
    hp = np.zeros(T, dtype=L.OPT_DTYPE)
    refs = []
    for s, n in enumerate(sizes):
        p0 = torch.randn(n, device="cuda", generator=g)
        P[s * stride:s * stride + n] = p0
        lr, wd, mom = 1e-2 * (s + 1), 1e-2, 0.9
        hp[s] = (lr, 0.9, 0.999, 1e-8, wd, mom, clip, code, 0, 0, n, 100, 0, 0, 0, 0)
        pr = p0.clone().requires_grad_(True)
        refs.append((pr, build_optimizer(name, [pr], lr, wd, mom)))
    nb = (max(sizes) + 2047) // 2048
    partials = torch.zeros(T * nb, device="cuda")
    ht = _up(hp)
    ids = torch.arange(T, dtype=torch.int32, device="cuda")
    for step in range(5):
        for s, n in enumerate(sizes):
            gr = torch.randn(n, device="cuda", generator=g)
            parts = torch.randn(gsplits, n, device="cuda", generator=g)
            parts[0] = gr - parts[1:].sum(0)
            gr = parts[0].clone()
            for sp in range(1, gsplits):            # the kernel adds the slabs in this order
                gr = gr + parts[sp]
            for sp in range(gsplits):
                G[(sp * T + s) * stride:(sp * T + s) * stride + n] = parts[sp]
            pr, opt = refs[s]
            pr.grad = gr.clone()
            if clip > 0:
                torch.nn.utils.clip_grad_norm_([pr], clip)
            opt.step()
        L.check(so.mt_optim_launch(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), Pb.data_ptr(), stride,
                                   ht.data_ptr(), ids.data_ptr(), T, max(sizes), partials.data_ptr(), nb, int(clip > 0),
                                   step, gsplits, T * stride, _stream()),
                "optim")
    torch.cuda.synchronize()
    for s, n in enumerate(sizes):
        got = P[s * stride:s * stride + n]
        torch.testing.assert_close(got, refs[s][0].detach(), rtol=2e-5, atol=2e-6)
        assert torch.equal(Pb[s * stride:s * stride + n], got.to(torch.bfloat16))
        assert torch.all(P[s * stride + n:(s + 1) * stride] == 0)


# ---------------------------------------------------------------------------------------------------------------
# Exact-f32 grouped GEMM (v_mfma_f32_16x16x4_f32): same descriptors, f32 operands and outputs, 128- and 64-tiles.
# Random f32 operands; the reference is a float64 product, so the tolerance only has to cover f32 accumulation.
def _tiles_edge(L, dims, edge):
    rows = [(-1, 0, 0)]
    for g, i, j in dims:
        for ti in range((i + edge - 1) // edge):
            for tj in range((j + edge - 1) // edge):
                rows.append((g, ti, tj))
    return np.array(rows, dtype=L.TILE_DTYPE)


F32_SHAPES = [(256, 128, 64), (512, 512, 512), (200, 72, 40), (128, 64, 8), (384, 256, 192), (130, 136, 520), (96, 42, 20)]


def _pad8(n):
    return (n + 7) // 8 * 8


# launch codes of the f32 GEMM: the tile edge
F32_FORMS = [128, 64, 32]


def _tile_of(code):
    return code


@pytest.mark.parametrize("edge", F32_FORMS)
@pytest.mark.parametrize("relu", [False, True])
def test_gemm_f32_forward_nt(lib, relu, edge):
    L, so = lib
    g = torch.Generator(device="cuda").manual_seed(11)
    groups = np.zeros(len(F32_SHAPES) + 1, dtype=L.GEMM_GROUP_DTYPE)
    keep, dims = [], []
    for gi, (m, n, k) in enumerate(F32_SHAPES):
        ldk, ldn = _pad8(k), _pad8(n)
        x = torch.zeros(m, ldk, device="cuda"); x[:, :k] = torch.randn(m, k, device="cuda", generator=g)
        w = torch.zeros(n, ldk, device="cuda"); w[:, :k] = torch.randn(n, k, device="cuda", generator=g) * 0.2
        b = torch.randn(n, device="cuda", generator=g)
        y = torch.full((m, ldn), 7.0, device="cuda")
        groups[gi] = L.gemm_group(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, m, n, k, ldk, ldk, ldn, 0, 1)
        keep.append((x, w, b, y)); dims.append((gi, m, n))
    groups[len(F32_SHAPES)]["active_steps"] = 0
    tiles = _tiles_edge(L, dims, _tile_of(edge))
    gt, tt = _up(groups), _up(tiles)
    L.check(so.mt_gemm_f32_launch(0, gt.data_ptr(), tt.data_ptr(), len(tiles), 0, int(relu), 0, 0, 1, 0, edge, _stream()),
            "gemm_f32")
    torch.cuda.synchronize()
    for (m, n, k), (x, w, b, y) in zip(F32_SHAPES, keep):
        ref = (x[:, :k].double() @ w[:, :k].double().t() + b.double())
        if relu:
            ref = ref.relu()
        torch.testing.assert_close(y[:, :n].double(), ref, rtol=2e-6, atol=2e-5)
        rounded = (n + 3) // 4 * 4          # pad columns inside the last 4-chunk are written as zeros
        assert torch.all(y[:, n:rounded] == 0)
        assert torch.all(y[:, rounded:] == 7.0)


@pytest.mark.parametrize("edge", F32_FORMS)
def test_gemm_f32_forward_identity_asymmetric(lib, edge):
    L, so = lib
    n = 128
    x = torch.eye(n, device="cuda")
    w = (torch.arange(n * n, device="cuda").reshape(n, n) % 251).float() + 0.125      # W[j][k]
    b = torch.zeros(n, device="cuda")
    y = torch.zeros(n, n, device="cuda")
    groups = np.zeros(1, dtype=L.GEMM_GROUP_DTYPE)
    groups[0] = L.gemm_group(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), 0, n, n, n, n, n, n, 0, 1)
    tiles = _tiles_edge(L, [(0, n, n)], _tile_of(edge))
    gt, tt = _up(groups), _up(tiles)
    L.check(so.mt_gemm_f32_launch(0, gt.data_ptr(), tt.data_ptr(), len(tiles), 0, 0, 0, 0, 1, 0, edge, _stream()), "gemm_f32")
    torch.cuda.synchronize()
    assert torch.equal(y, w.t())          # exact: every product is x * 1 or x * 0


@pytest.mark.parametrize("edge", F32_FORMS)
@pytest.mark.parametrize("mask", [False, True])
def test_gemm_f32_dgrad_nn(lib, mask, edge):
    L, so = lib
    g = torch.Generator(device="cuda").manual_seed(12)
    groups = np.zeros(len(F32_SHAPES), dtype=L.GEMM_GROUP_DTYPE)
    keep, dims = [], []
    for gi, (m, n, k) in enumerate(F32_SHAPES):      # dX[m,k] = dY[m,n] @ W[n,k]
        ldk, ldn = _pad8(k), _pad8(n)
        dy = torch.zeros(m, ldn, device="cuda"); dy[:, :n] = torch.randn(m, n, device="cuda", generator=g)
        w = torch.zeros(n, ldk, device="cuda"); w[:, :k] = torch.randn(n, k, device="cuda", generator=g) * 0.2
        h = torch.zeros(m, ldk, device="cuda"); h[:, :k] = torch.randn(m, k, device="cuda", generator=g).relu()
        dx = torch.full((m, ldk), 7.0, device="cuda")
        groups[gi] = L.gemm_group(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), h.data_ptr(), 0, m, k, n, ldn, ldk, ldk, ldk, 1)
        keep.append((dy, w, h, dx)); dims.append((gi, m, k))
    tiles = _tiles_edge(L, dims, _tile_of(edge))
    gt, tt = _up(groups), _up(tiles)
    L.check(so.mt_gemm_f32_launch(1, gt.data_ptr(), tt.data_ptr(), len(tiles), 0, int(mask), 0, 0, 1, 0, edge, _stream()),
            "gemm_f32")
    torch.cuda.synchronize()
    for (m, n, k), (dy, w, h, dx) in zip(F32_SHAPES, keep):
        ref = dy[:, :n].double() @ w[:, :k].double()
        if mask:
            ref = ref * (h[:, :k] > 0)
        torch.testing.assert_close(dx[:, :k].double(), ref, rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("edge", F32_FORMS)
@pytest.mark.parametrize("ksplits", [1, 3])
def test_gemm_f32_wgrad_tn(lib, ksplits, edge):
    L, so = lib
    g = torch.Generator(device="cuda").manual_seed(13)
    groups = np.zeros(len(F32_SHAPES), dtype=L.GEMM_GROUP_DTYPE)
    keep, dims = [], []
    for gi, (m, n, k) in enumerate(F32_SHAPES):      # dW[n,k] = dY[m,n]^T @ X[m,k];  db[n] = sum_m dY[m,n]
        ldk, ldn = _pad8(k), _pad8(n)
        dy = torch.zeros(m, ldn, device="cuda"); dy[:, :n] = torch.randn(m, n, device="cuda", generator=g) * 0.1
        x = torch.zeros(m, ldk, device="cuda"); x[:, :k] = torch.randn(m, k, device="cuda", generator=g)
        arena = torch.full((ksplits, 600 * 600), 5.0, device="cuda")
        dw_all = arena[:, :n * ldk].view(ksplits, n, ldk)
        dw_all.zero_()
        db_all = arena[:, 590 * 600:590 * 600 + n]
        groups[gi] = L.gemm_group(dy.data_ptr(), x.data_ptr(), arena.data_ptr(), 0, db_all.data_ptr(), n, k, m, ldn, ldk, ldk, 0, 1)
        keep.append((dy, x, dw_all, db_all)); dims.append((gi, n, k))
    tiles = _tiles_edge(L, dims, _tile_of(edge))
    if ksplits > 1:
        base = tiles[1:]
        reps = []
        for sp in range(ksplits):
            t = base.copy(); t["ti"] = t["ti"] | (sp << 8); reps.append(t)
        tiles = np.concatenate([tiles[:1]] + reps)
    gt, tt = _up(groups), _up(tiles)
    L.check(so.mt_gemm_f32_launch(2, gt.data_ptr(), tt.data_ptr(), len(tiles), 0, 1, 0, 0, ksplits, 600 * 600, edge, _stream()),
            "gemm_f32")
    torch.cuda.synchronize()
    for (m, n, k), (dy, x, dw_all, db_all) in zip(F32_SHAPES, keep):
        dw, db = dw_all.double().sum(0), db_all.double().sum(0)
        ref = dy[:, :n].double().t() @ x[:, :k].double()
        torch.testing.assert_close(dw[:, :k], ref, rtol=2e-6, atol=2e-5)
        torch.testing.assert_close(db, dy[:, :n].double().sum(0), rtol=2e-6, atol=2e-5)
        if dw.shape[1] > k:
            assert torch.all(dw[:, k:] == 0)


def test_gather_f32_rows(lib):
    L, so = lib
    from mitune.packed.permutation import half_bits, permute, round_keys
    n, F, Fpad, B = 5000, 10, 16, 96
    X = torch.randn(n, F, device="cuda"); Y = torch.randn(n, device="cuda")
    slots = np.zeros(2, dtype=L.GATHER_DTYPE)
    outs = []
    for s in range(2):
        x0 = torch.full((B, Fpad), 9.0, device="cuda"); y = torch.zeros(B, device="cuda")
        keys = round_keys(11, 100 + s, 2)
        slots[s] = (x0.data_ptr(), y.data_ptr(), keys, B - 8 * s, 50)
        outs.append((x0, y, keys))
    st = _up(slots)
    ids = torch.arange(2, dtype=torch.int32, device="cuda")
    L.check(so.mt_gather_f32_launch(st.data_ptr(), ids.data_ptr(), 2, B, X.data_ptr(), Y.data_ptr(), n, F, Fpad,
                                    half_bits(n), 7, _stream()), "gather_f32")
    torch.cuda.synchronize()
    for s, (x0, y, keys) in enumerate(outs):
        b = B - 8 * s
        idx = torch.from_numpy(permute(np.arange(7 * b, 7 * b + b, dtype=np.uint32), n, keys)).cuda()
        assert torch.equal(x0[:b, :F], X[idx])
        assert torch.all(x0[:b, F:] == 0)
        assert torch.equal(y[:b], Y[idx])


def test_head_loss_f32_activations(lib):
    L, so = lib
    g = torch.Generator(device="cuda").manual_seed(14)
    cfgs = [(256, 512), (100, 64), (96, 200)]
    slots = np.zeros(len(cfgs), dtype=L.HEAD_DTYPE)
    keep = []
    for s, (B, W) in enumerate(cfgs):
        ld = _pad8(W)
        H = torch.zeros(B, ld, device="cuda"); H[:, :W] = torch.randn(B, W, device="cuda", generator=g).relu()
        dH = torch.zeros(B, ld, device="cuda")
        w = torch.randn(W, device="cuda", generator=g) * 0.1; b = torch.randn(1, device="cuda", generator=g)
        gw = torch.zeros(W, device="cuda"); gb = torch.zeros(1, device="cuda")
        y = torch.randn(B, device="cuda", generator=g) + 2.0
        acc = torch.full((1,), 1.0, device="cuda"); met = torch.zeros(3, device="cuda", dtype=torch.float64)
        slots[s] = (H.data_ptr(), dH.data_ptr(), w.data_ptr(), b.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                    y.data_ptr(), acc.data_ptr(), met.data_ptr(), B, W, ld, 0, 1, 0)
        keep.append((H, dH, w, b, gw, gb, y, acc, met))
    st = _up(slots)
    ids = torch.arange(len(cfgs), dtype=torch.int32, device="cuda")
    scratch = torch.zeros(int(so.mt_head_scratch_floats(len(cfgs), 256, 512)), device="cuda")
    L.check(so.mt_head_loss_f32_launch(st.data_ptr(), ids.data_ptr(), len(cfgs), 256, 512, 0, scratch.data_ptr(), 1,
                                       _stream()), "head_loss_f32")
    L.check(so.mt_head_eval_f32_launch(st.data_ptr(), ids.data_ptr(), len(cfgs), 512, 0, 0, 256, scratch.data_ptr(),
                                       _stream()), "head_eval_f32")
    torch.cuda.synchronize()
    for (B, W), (H, dH, w, b, gw, gb, y, acc, met) in zip(cfgs, keep):
        h = H[:, :W].clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = h @ wr + br
        l = ((out - y) ** 2).mean()
        l.backward()
        torch.testing.assert_close(acc, torch.tensor([1.0 + l.item()], device="cuda"), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(gw, wr.grad, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(gb, br.grad, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(dH[:, :W], (h.grad * (h > 0)).detach(), rtol=1e-5, atol=1e-7)
        d = (out - y).detach()
        torch.testing.assert_close(met[1].float(), (d * d).sum(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("f32", [False, True])
def test_head_loss_wide_layers(lib, f32):
    """Heads wider than 2048 columns (data-parallel trials) take the two-pass form of the head/loss kernels."""
    L, so = lib
    g = torch.Generator(device="cuda").manual_seed(24)
    cfgs = [(200, 8192), (64, 4096)]
    dt = torch.float32 if f32 else torch.bfloat16
    slots = np.zeros(len(cfgs), dtype=L.HEAD_DTYPE)
    keep = []
    for s, (B, W) in enumerate(cfgs):
        H = torch.randn(B, W, device="cuda", generator=g).relu().to(dt)
        dH = torch.zeros(B, W, device="cuda", dtype=dt)
        w = torch.randn(W, device="cuda", generator=g) * 0.02; b = torch.randn(1, device="cuda", generator=g)
        gw = torch.zeros(W, device="cuda"); gb = torch.zeros(1, device="cuda")
        y = torch.randn(B, device="cuda", generator=g)
        acc = torch.zeros(1, device="cuda"); met = torch.zeros(3, device="cuda", dtype=torch.float64)
        slots[s] = (H.data_ptr(), dH.data_ptr(), w.data_ptr(), b.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                    y.data_ptr(), acc.data_ptr(), met.data_ptr(), B, W, W, 0, 1, 0)
        keep.append((H, dH, w, b, gw, gb, y, acc, met))
    st = _up(slots)
    ids = torch.arange(len(cfgs), dtype=torch.int32, device="cuda")
    scratch = torch.zeros(int(so.mt_head_scratch_floats(len(cfgs), 256, 8192)), device="cuda")
    if f32:
        L.check(so.mt_head_loss_f32_launch(st.data_ptr(), ids.data_ptr(), len(cfgs), 256, 8192, 0, scratch.data_ptr(), 1,
                                           _stream()), "head_loss_f32")
        L.check(so.mt_head_eval_f32_launch(st.data_ptr(), ids.data_ptr(), len(cfgs), 8192, 0, 0, 256, scratch.data_ptr(),
                                           _stream()), "head_eval_f32")
    else:
        L.check(so.mt_head_loss_launch(st.data_ptr(), ids.data_ptr(), len(cfgs), 256, 8192, 0, scratch.data_ptr(), _stream()),
                "head_loss")
        L.check(so.mt_head_eval_launch(st.data_ptr(), ids.data_ptr(), len(cfgs), 8192, 0, 0, 256, scratch.data_ptr(),
                                       _stream()), "head_eval")
    torch.cuda.synchronize()
    for (B, W), (H, dH, w, b, gw, gb, y, acc, met) in zip(cfgs, keep):
        h = H.float().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = h @ wr + br
        l = ((out - y) ** 2).mean()
        l.backward()
        torch.testing.assert_close(acc, torch.tensor([l.item()], device="cuda"), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(gw, wr.grad, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(gb, br.grad, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(dH.float(), (h.grad * (h > 0)).detach(), rtol=1e-2, atol=1e-6)
        d = (out - y).detach()
        torch.testing.assert_close(met[1].float(), (d * d).sum(), rtol=1e-4, atol=1e-3)
