"""GPU parity of the individual HIP kernels (through the C ABI) against the CPU oracle's arithmetic
(torch fp32 ATen ops, the same calls oracle/restate.py makes).  Tolerances: fp32 1e-4 relative,
bf16 2e-2 relative (BASELINE.json north_star)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def _dev():
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _ref_dwconv_ln(x_bcl, w, b, lw, lb):
    c = x_bcl.shape[1]
    y = F.conv1d(x_bcl, w, b, padding=3, groups=c).transpose(1, 2)
    return F.layer_norm(y, (c,), lw, lb, 1e-5)


K1_SHAPES = [(3, 64, 250), (2, 128, 125), (2, 256, 62), (2, 768, 31), (2, 1600, 31), (1, 200, 31), (2, 2048, 31),
             (5, 64, 7), (1, 64, 1), (9, 128, 33)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", K1_SHAPES)
@pytest.mark.parametrize("layout", ["BLC", "BCL"])
def test_dwconv7_ln_fwd(shape, dtype, layout):
    from multimodal_amd import ops, _lib as L
    B, C, Ln = shape
    x = _rand(B, C, Ln, seed=1)
    w, b = _rand(C, 1, 7, seed=2, scale=0.5), _rand(C, seed=3, scale=0.1)
    lw, lb = 1 + _rand(C, seed=4, scale=0.2), _rand(C, seed=5, scale=0.1)
    xq = x.to(dtype).float()                       # the kernel sees the rounded input
    ref = _ref_dwconv_ln(xq, w, b, lw, lb)
    xd = x.to(dtype).to(_dev())
    if layout == "BLC":
        xin, lay = xd.transpose(1, 2).contiguous(), L.LAYOUT_BLC
    else:
        xin, lay = xd.contiguous(), L.LAYOUT_BCL
    y, mean, rstd = ops.dwconv7_ln_fwd(xin, w.to(_dev()), b.to(_dev()), lw.to(_dev()), lb.to(_dev()), layout=lay,
                                       save_stats=True)
    torch.cuda.synchronize()
    assert y.shape == (B, Ln, C) and y.dtype == dtype
    assert rel_err(y.float().cpu(), ref) < TOL[dtype]
    conv = F.conv1d(xq, w, b, padding=3, groups=C).transpose(1, 2)
    assert rel_err(mean.cpu(), conv.mean(-1).reshape(-1)) < 1e-4
    assert rel_err(rstd.cpu(), (conv.var(-1, unbiased=False) + 1e-5).rsqrt().reshape(-1)) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", K1_SHAPES)
def test_dwconv7_ln_bwd(shape, dtype):
    from multimodal_amd import ops
    B, C, Ln = shape
    x = _rand(B, C, Ln, seed=1).to(dtype).float().requires_grad_(True)
    w = _rand(C, 1, 7, seed=2, scale=0.5).requires_grad_(True)
    b = _rand(C, seed=3, scale=0.1).requires_grad_(True)
    lw = (1 + _rand(C, seed=4, scale=0.2)).requires_grad_(True)
    lb = _rand(C, seed=5, scale=0.1).requires_grad_(True)
    dy = _rand(B, Ln, C, seed=6).to(dtype).float()
    dres = _rand(B, Ln, C, seed=7).to(dtype).float()
    ref = _ref_dwconv_ln(x, w, b, lw, lb)
    ref.backward(dy)
    d = _dev()
    xin = x.detach().transpose(1, 2).contiguous().to(dtype).to(d)
    y, mean, rstd = ops.dwconv7_ln_fwd(xin, w.detach().to(d), b.detach().to(d), lw.detach().to(d), lb.detach().to(d),
                                       save_stats=True)
    dx, dw, db, dlw, dlb = ops.dwconv7_ln_bwd(xin, dy.to(dtype).to(d), dres.to(dtype).to(d), w.detach().to(d),
                                              b.detach().to(d), lw.detach().to(d), mean, rstd)
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert rel_err(dx.float().cpu(), x.grad.transpose(1, 2) + dres) < tol
    ptol = 2e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(dw.cpu(), w.grad) < ptol
    assert rel_err(db.cpu(), b.grad) < ptol
    assert rel_err(dlw.cpu(), lw.grad) < ptol
    assert rel_err(dlb.cpu(), lb.grad) < ptol


def test_dwconv7_ln_rejects_bad_args():
    from multimodal_amd import ops
    x = torch.zeros(2, 31, 64)
    with pytest.raises(RuntimeError):
        ops.dwconv7_ln_fwd(x, torch.zeros(64, 1, 7), torch.zeros(64), torch.zeros(64), torch.zeros(64))   # CPU tensor


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Lin", [(3, 1000), (1, 4), (2, 1004)])
def test_stem_ln(B, Lin, dtype):
    from multimodal_amd import ops
    x = ((_rand(B, Lin, seed=1) + 1) * 0.5).requires_grad_(True)
    w = _rand(64, 1, 4, seed=2).requires_grad_(True)
    b = _rand(64, seed=3, scale=0.1).requires_grad_(True)
    lw = (1 + _rand(64, seed=4, scale=0.2)).requires_grad_(True)
    lb = _rand(64, seed=5, scale=0.1).requires_grad_(True)
    ref = F.layer_norm(F.conv1d(x[:, None], w, b, stride=4).transpose(1, 2), (64,), lw, lb, 1e-5)
    d = _dev()
    y, mean, rstd = ops.stem_ln_fwd(x.detach().to(d), w.detach().to(d), b.detach().to(d), lw.detach().to(d),
                                    lb.detach().to(d), dtype, save_stats=True)
    assert rel_err(y.float().cpu(), ref.detach()) < TOL[dtype]
    dy = _rand(*ref.shape, seed=9).to(dtype).float()
    ref.backward(dy)
    dx, dw, db, dlw, dlb = ops.stem_ln_bwd(x.detach().to(d), dy.to(dtype).to(d), w.detach().to(d), b.detach().to(d),
                                           lw.detach().to(d), mean, rstd, need_dx=True)
    torch.cuda.synchronize()
    ptol = 2e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(dx.cpu(), x.grad) < ptol
    assert rel_err(dw.cpu(), w.grad) < ptol
    assert rel_err(db.cpu(), b.grad) < ptol
    assert rel_err(dlw.cpu(), lw.grad) < ptol
    assert rel_err(dlb.cpu(), lb.grad) < ptol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("G,Lin,Lout,C", [(1, 37, 37, 64), (3, 125, 124, 128), (2, 62, 62, 256), (2, 31, 30, 1600),
                                          (1, 9, 9, 200), (2, 5, 5, 2048), (1, 6, 6, 50), (1, 3, 3, 768), (2, 7, 7, 4096),
                                          (1, 5, 4, 1028), (1, 1, 1, 516)])
def test_layernorm(G, Lin, Lout, C, dtype):
    from multimodal_amd import ops
    x = _rand(G, Lin, C, seed=1).to(dtype).float().requires_grad_(True)
    w = (1 + _rand(C, seed=2, scale=0.2)).requires_grad_(True)
    b = _rand(C, seed=3, scale=0.1).requires_grad_(True)
    ref = F.layer_norm(x[:, :Lout], (C,), w, b, 1e-5)
    d = _dev()
    xd = x.detach().to(dtype).to(d)
    drop = Lout if Lout != Lin else None
    y, mean, rstd = ops.layernorm_fwd(xd, w.detach().to(d), b.detach().to(d), save_stats=True, drop_tail_to=drop)
    assert rel_err(y.float().cpu().reshape(ref.shape), ref.detach()) < TOL[dtype]
    dy = _rand(G, Lout, C, seed=4).to(dtype).float()
    dres = _rand(G, Lin, C, seed=5).to(dtype).float()
    ref.backward(dy)
    dx, dw, db = ops.layernorm_bwd(xd, dy.to(dtype).to(d), dres.to(dtype).to(d), w.detach().to(d), mean, rstd,
                                   drop_tail_to=drop)
    torch.cuda.synchronize()
    assert rel_err(dx.float().cpu(), x.grad + dres) < TOL[dtype]
    ptol = 2e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(dw.cpu(), w.grad) < ptol
    assert rel_err(db.cpu(), b.grad) < ptol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(315, 1600), (2600, 768), (7, 2048), (1, 516), (530, 4096)])
def test_layernorm_bwd_with_bias_gradient_sums(rows, C, dtype):
    """mm_layernorm_bwd_sums == mm_layernorm_bwd + the column sums of its residual input and of its dx output."""
    from multimodal_amd import ops
    d = _dev()
    x = _rand(rows, C, seed=1).to(dtype).to(d)
    w = (1 + _rand(C, seed=2, scale=0.2)).to(d)
    b = _rand(C, seed=3, scale=0.1).to(d)
    _, mean, rstd = ops.layernorm_fwd(x, w, b, save_stats=True)
    dy = _rand(rows, C, seed=4).to(dtype).to(d)
    dres = _rand(rows, C, seed=5).to(dtype).to(d)
    dx0, dw0, db0 = ops.layernorm_bwd(x, dy, dres, w, mean, rstd)
    dx1, dw1, db1, rsum, xsum = ops.layernorm_bwd_sums(x, dy, dres, w, mean, rstd)
    assert torch.equal(dx0, dx1) and torch.equal(dw0, dw1) and torch.equal(db0, db1)
    assert rel_err(rsum.cpu(), dres.float().sum(0).cpu()) < 1e-5
    # the kernel sums dx before it is rounded to the storage dtype
    assert rel_err(xsum.cpu(), dx0.float().sum(0).cpu()) < (1e-5 if dtype == torch.float32 else 5e-3)
    assert not ops.layernorm_bwd_sums_ok(512) and not ops.layernorm_bwd_sums_ok(1030)
    with pytest.raises(RuntimeError):
        ops.layernorm_bwd_sums(x[:, :256].contiguous(), dy[:, :256].contiguous(), dres[:, :256].contiguous(), w[:256],
                               mean, rstd)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(315, 6400), (1234, 776), (3, 8), (10080, 256)])
def test_relu2_bwd_with_bias_gradient_sum(M, N, dtype):
    from multimodal_amd import ops
    d = _dev()
    h = _rand(M, N, seed=2, scale=2.0).to(dtype).to(d)
    da = _rand(M, N, seed=3).to(dtype).to(d)
    ref = ops.relu2_bwd_(h, da.clone())
    got, s = ops.relu2_bwd_colsum_(h, da.clone())
    assert torch.equal(got, ref)
    assert rel_err(s.cpu(), ref.float().sum(0).cpu()) < 1e-5
    if dtype == torch.bfloat16:
        with pytest.raises(RuntimeError):
            ops.relu2_bwd_colsum_(h[:, :4].contiguous(), da[:, :4].contiguous())      # rows not 16-byte multiples


GEMM_SHAPES = [(992, 6400, 1600), (8000, 256, 64), (100, 591, 768), (77, 130, 250), (16, 16, 8), (1, 1, 1),
               (300, 200, 200), (513, 129, 1030), (1984, 512, 128)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("layout", ["nt", "nn", "tt"])
def test_gemm_layouts(M, N, K, dtype, layout):
    """nt: A[M,K] B[N,K] (forward linear); nn: B stored [K,N] (dX = dY.W); tt: both K-major (dW = dY^T.X)."""
    from multimodal_amd import ops
    a = _rand(M, K, seed=1).to(dtype)
    b = _rand(N, K, seed=2).to(dtype)
    ref = a.double() @ b.double().t()
    d = _dev()
    ad = a.to(d) if layout != "tt" else a.t().contiguous().to(d)
    bd = b.to(d) if layout == "nt" else b.t().contiguous().to(d)
    out = ops.gemm(ad, bd, M, N, K, a_kmajor=(layout == "tt"), b_kmajor=(layout != "nt"), out_dtype=torch.float32)
    torch.cuda.synchronize()
    err = (out.double().cpu() - ref).abs().max().item()
    scale = (a.double().abs() @ b.double().abs().t()).max().item()
    assert err <= 2e-6 * scale + 1e-12, (err, scale)          # f32 accumulate of exactly representable products


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    from multimodal_amd import ops
    M, N, K, Lr = 62 * 3, 256, 64, 62
    d = _dev()
    a = _rand(M, K, seed=1).to(dtype); w = _rand(N, K, seed=2, scale=0.3).to(dtype)
    bias = _rand(N, seed=3, scale=0.2)
    af, wf = a.float(), w.float()
    # pwconv1 + GELU with saved pre-activation (convnext.py:41-42)
    pre = torch.empty(M, N, dtype=dtype, device=d)
    out = ops.gemm(a.to(d), w.to(d), M, N, K, bias=bias.to(d), pre=pre, act=ops.ACT_GELU)
    h = af @ wf.t() + bias
    assert rel_err(pre.float().cpu(), h) < TOL[dtype]
    assert rel_err(out.float().cpu(), F.gelu(h)) < TOL[dtype]
    # relu^2 (transformer_decoder.py:161)
    out = ops.gemm(a.to(d), w.to(d), M, N, K, bias=bias.to(d), act=ops.ACT_RELU2)
    assert rel_err(out.float().cpu(), F.relu(h).square()) < TOL[dtype]
    # pwconv2 + LayerScale + DropPath factor + residual (convnext.py:43-47)
    gamma = _rand(N, seed=4); rs = torch.tensor([1.0, 0.0, 1.25]); res = _rand(M, N, seed=5).to(dtype)
    out = ops.gemm(a.to(d), w.to(d), M, N, K, bias=bias.to(d), gamma=gamma.to(d), rowscale=rs.to(d), rows_per_scale=Lr,
                   res=res.to(d))
    ref = res.float() + rs.repeat_interleave(Lr)[:, None] * (gamma * h)
    assert rel_err(out.float().cpu(), ref) < TOL[dtype]
    # backward epilogue: dX * GELU'(pre)
    hp = _rand(M, N, seed=6, scale=2.0).to(dtype)
    out = ops.gemm(a.to(d), w.to(d), M, N, K, mul=hp.to(d), act=ops.ACT_GELU)
    hh = hp.float().requires_grad_(True)
    F.gelu(hh).sum().backward()
    assert rel_err(out.float().cpu(), (af @ wf.t()) * hh.grad) < TOL[dtype]
    # split-K (weight gradients): deterministic two-stage
    o1 = ops.gemm(a.t().contiguous().to(d), w.t().contiguous().to(d), M, N, K, a_kmajor=True, b_kmajor=True,
                  out_dtype=torch.float32, split_k=1)
    big_a = _rand(5000, 96, seed=7).to(dtype); big_b = _rand(5000, 80, seed=8).to(dtype)
    o2 = ops.gemm(big_a.to(d), big_b.to(d), 96, 80, 5000, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, split_k=8)
    o3 = ops.gemm(big_a.to(d), big_b.to(d), 96, 80, 5000, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, split_k=8)
    torch.cuda.synchronize()
    assert rel_err(o1.cpu(), af @ wf.t()) < 1e-5
    assert rel_err(o2.cpu(), big_a.float().t() @ big_b.float()) < 1e-5
    assert torch.equal(o2, o3)                                 # bitwise reproducible


@pytest.mark.parametrize("M,N,K", [(2100, 2052, 192), (4096, 1024, 64), (1111, 4100, 320), (2304, 2304, 1600),
                                   (10080, 1600, 128), (13056, 1596, 64), (10080, 1280, 128)])   # the last three pick 224x320, 256x320, 224x256 tiles
def test_gemm_large_tile(M, N, K):
    """The 256x256 global->LDS-DMA kernel (bf16, K-contiguous operands, K % 64 == 0): ragged M / N edges, every
    epilogue, both output types; compared with the exact product of the bf16 operands."""
    from multimodal_amd import ops
    d = _dev()
    a = _rand(M, K, seed=1).to(torch.bfloat16); w = _rand(N, K, seed=2, scale=0.3).to(torch.bfloat16)
    ref = (a.double() @ w.double().t())
    scale = (a.double().abs() @ w.double().abs().t()).max().item()
    out = ops.gemm(a.to(d), w.to(d), M, N, K, out_dtype=torch.float32)
    assert (out.double().cpu() - ref).abs().max().item() <= 2e-6 * scale
    bias = _rand(N, seed=3, scale=0.2)
    h = ref.float() + bias
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=d)
    res = _rand(M, N, seed=5).to(torch.bfloat16)
    out = ops.gemm(a.to(d), w.to(d), M, N, K, bias=bias.to(d), pre=pre, act=ops.ACT_RELU2)
    assert rel_err(pre.float().cpu(), h) < TOL[torch.bfloat16]
    assert rel_err(out.float().cpu(), F.relu(h).square()) < TOL[torch.bfloat16]
    out = ops.gemm(a.to(d), w.to(d), M, N, K, bias=bias.to(d), res=res.to(d))
    assert rel_err(out.float().cpu(), h + res.float()) < TOL[torch.bfloat16]
    hp = _rand(M, N, seed=6, scale=2.0).to(torch.bfloat16)
    out = ops.gemm(a.to(d), w.to(d), M, N, K, mul=hp.to(d), act=ops.ACT_RELU2)
    assert rel_err(out.float().cpu(), ref.float() * 2 * F.relu(hp.float())) < TOL[torch.bfloat16]


@pytest.mark.parametrize("M,N,K", [(1600, 1600, 10080), (1288, 648, 1100), (512, 2568, 2048), (6400, 1600, 4128)])
def test_gemm_large_tile_weight_gradient_form(M, N, K, monkeypatch):
    """Both operands K-major (dW = dY^T X) on the large-tile kernel: K not a multiple of 64 (zero page), ragged M / N
    (multiples of 8), every tile shape, with and without its K split, float32 and bf16 results; exact products of bf16 operands."""
    from multimodal_amd import ops
    d = _dev()
    a = _rand(K, M, seed=1).to(torch.bfloat16); b = _rand(K, N, seed=2, scale=0.3).to(torch.bfloat16)
    ref = a.double().t() @ b.double()
    scale = (a.double().abs().t() @ b.double().abs()).max().item()
    ad, bd = a.to(d), b.to(d)
    out = ops.gemm(ad, bd, M, N, K, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    assert (out.double().cpu() - ref).abs().max().item() <= 4e-6 * scale
    out2 = ops.gemm(ad, bd, M, N, K, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    assert torch.equal(out, out2)                                   # fixed-order slab sums: deterministic
    out16 = ops.gemm(ad, bd, M, N, K, a_kmajor=True, b_kmajor=True)
    assert rel_err(out16.float().cpu(), ref.float()) < TOL[torch.bfloat16]
    # a strided column slice as the A operand (the per-projection weight gradients of a fused q/k/v gradient tensor)
    wide = torch.zeros(K, M + 64, dtype=torch.bfloat16, device=d)
    wide[:, 32: 32 + M] = ad
    out3 = ops.gemm(wide[:, 32: 32 + M], bd, M, N, K, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    assert (out3.double().cpu() - ref).abs().max().item() <= 4e-6 * scale


def test_cast_transpose_and_relu2():
    from multimodal_amd import ops
    d = _dev()
    w = _rand(130, 77, seed=1).to(d)
    for dt in (torch.bfloat16, torch.float32):
        out = torch.zeros(77, 3 * 130, dtype=dt, device=d)
        ops.cast_transpose(w, out[:, 130:260])
        assert torch.equal(out[:, 130:260], w.t().to(dt)) and out[:, :130].abs().sum() == 0 and out[:, 260:].abs().sum() == 0
        w8 = _rand(200, 132, seed=4).to(d)                          # R % 8 == 0, C % 4 == 0: the 16-byte form, ragged tiles
        out8 = torch.zeros(132, 2 * 200 + 8, dtype=dt, device=d)
        ops.cast_transpose(w8, out8[:, 200:400])
        assert torch.equal(out8[:, 200:400], w8.t().to(dt)) and out8[:, :200].abs().sum() == 0 and out8[:, 400:].abs().sum() == 0
        h = _rand(37, 101, seed=2, scale=2.0).to(dt).to(d)
        a = ops.relu2_fwd(h)
        assert torch.equal(a, torch.relu(h.float()).square().to(dt))
        da = _rand(37, 101, seed=3).to(dt).to(d)
        ref = (da.float() * 2 * torch.relu(h.float())).to(dt)
        assert torch.equal(ops.relu2_bwd_(h, da.clone()), ref)


def test_colsum():
    from multimodal_amd import ops
    for dtype in (torch.float32, torch.bfloat16):
        x = _rand(1234, 777, seed=1).to(dtype)
        s = ops.colsum(x.to(_dev()), 1234, 777)
        assert rel_err(s.cpu(), x.float().sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_scale", [True, False])
@pytest.mark.parametrize("shape", [(32, 250, 64), (32, 125, 128), (32, 62, 256), (32, 31, 1600), (3, 13, 72), (1, 5, 2048), (2, 7, 36)])
def test_layerscale_bwd_all_stage_shapes(shape, with_scale, dtype):
    """LayerScale + DropPath backward (models/convnext.py:44-47 through autograd): dz = dout * gamma * s[row // L],
    dgamma = sum dout * s * z, db2 = sum dz -- on every encoder stage shape (bf16: the 8-channels-per-lane kernel of round 3),
    a ragged shape, the widest row it takes, and C % 8 != 0 (the scalar kernel)."""
    from multimodal_amd import ops
    B, Ln, C = shape
    M = B * Ln
    dev = torch.device("cuda:0")
    dout = _rand(M, C, seed=1).to(dtype)
    z = _rand(M, C, seed=2).to(dtype)
    gamma = _rand(C, seed=3) * 0.5
    s = (torch.rand(B, generator=torch.Generator().manual_seed(4)) > 0.3).float() / 0.7 if with_scale else None
    d32 = dout.float() * (s.repeat_interleave(Ln)[:, None] if with_scale else 1.0)
    ref_dz = d32 * gamma
    ref_dg = (d32 * z.float()).sum(0)
    ref_db = ref_dz.to(dtype).float().sum(0) if dtype == torch.float32 else ref_dz.sum(0)
    dz, dg, db = ops.layerscale_bwd(dout.to(dev), z.to(dev), gamma.to(dev), s.to(dev) if with_scale else None, Ln)
    torch.cuda.synchronize()
    assert rel_err(dz.float().cpu(), ref_dz) < TOL[dtype]
    assert rel_err(dg.cpu(), ref_dg) < 1e-4, (dg.cpu() - ref_dg).abs().max()
    assert rel_err(db.cpu(), ref_db) < 1e-4
