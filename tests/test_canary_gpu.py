"""Sanitizer-style checks (GPU AddressSanitizer is not available on this pool):
 * canary-padded buffers: every tensor the op wrappers allocate for a kernel's output or workspace sits between two
   4 KiB guard zones filled with a byte pattern; after the launch the guards must be intact (an out-of-bounds store of
   a tail tile, a halo row or a mis-sized workspace shows up here, at shapes that are NOT multiples of the tile sizes);
 * determinism: the same launch twice returns bit-identical results for the kernels that reduce across workgroups
   (K1b, K3b, K4, LayerNorm backward, split-K GEMM, column sums) -- no atomics, fixed-order reductions."""
import contextlib
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GUARD = 4096
PATTERN = 0xA5


class _GuardedTorch(types.ModuleType):
    """Stands in for the `torch` module inside multimodal_amd.ops: allocation functions hand out the middle of a larger
    byte buffer whose head and tail are guard zones; everything else is forwarded."""

    def __init__(self):
        super().__init__("torch_guarded")
        self.bufs = []

    def __getattr__(self, k):
        return getattr(torch, k)

    def _alloc(self, shape, dtype, device, fill=None):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        pad = (-nbytes) % 256
        raw = torch.full((GUARD + nbytes + pad + GUARD,), PATTERN, dtype=torch.uint8, device=device)
        self.bufs.append((raw, nbytes))
        t = raw[GUARD:GUARD + nbytes].view(dtype).view(shape)
        if fill is not None:
            t.fill_(fill)
        return t

    def empty(self, *shape, dtype=None, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = shape[0]
        return self._alloc(shape, dtype or torch.float32, device)

    def empty_like(self, t, **kw):
        return self._alloc(t.shape, t.dtype, t.device)

    def zeros_like(self, t, **kw):
        return self._alloc(t.shape, t.dtype, t.device, fill=0)

    def check(self):
        torch.cuda.synchronize()
        for raw, nbytes in self.bufs:
            head, tail = raw[:GUARD], raw[GUARD + nbytes:]
            assert bool((head == PATTERN).all()), f"store BEFORE a {nbytes}-byte buffer"
            assert bool((tail == PATTERN).all()), f"store PAST the end of a {nbytes}-byte buffer"
        return len(self.bufs)


@contextlib.contextmanager
def canary():
    from multimodal_amd import ops
    g = _GuardedTorch()
    real, ops.torch = ops.torch, g
    saved_ws = dict(ops._ws_cache)
    ops._ws_cache.clear()                      # workspaces are re-allocated inside the guards, at their exact size
    real_ws = ops.workspace

    def exact_ws(nbytes, device, tag="ws"):
        return g._alloc((max(int(nbytes), 1),), torch.uint8, device)
    ops.workspace = exact_ws
    try:
        yield g
        assert g.check() > 0
    finally:
        ops.torch, ops.workspace = real, real_ws
        ops._ws_cache.clear()
        ops._ws_cache.update(saved_ws)


def _r(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dtype).to(DEV)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,Ln", [(3, 64, 250), (2, 128, 125), (3, 256, 62), (2, 1600, 31), (1, 2048, 31), (2, 768, 7), (5, 64, 1)])
def test_k1_forward_backward_stay_in_bounds(dtype, B, C, Ln):
    from multimodal_amd import ops
    x = _r(B, Ln, C, dtype=dtype)
    w, b, lw, lb = _r(C, 1, 7, seed=1), _r(C, seed=2), _r(C, seed=3) + 1.5, _r(C, seed=4)
    with canary():
        y, mean, rstd = ops.dwconv7_ln_fwd(x, w, b, lw, lb, save_stats=True)
        ops.dwconv7_ln_bwd(x, _r(B, Ln, C, dtype=dtype, seed=5), _r(B, Ln, C, dtype=dtype, seed=6), w, b, lw, mean, rstd)
    xc = x.transpose(1, 2).contiguous()
    with canary():
        from multimodal_amd import _lib as L
        ops.dwconv7_ln_fwd(xc, w, b, lw, lb, layout=L.LAYOUT_BCL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(993, 1600), (77, 64), (10, 2048), (315, 200), (5, 768)])
def test_layernorm_stays_in_bounds(dtype, rows, C):
    from multimodal_amd import ops
    x = _r(rows, C, dtype=dtype)
    w, b = _r(C, seed=1) + 1.5, _r(C, seed=2)
    with canary():
        y, mean, rstd = ops.layernorm_fwd(x, w, b, save_stats=True)
        ops.layernorm_bwd(x, _r(rows, C, dtype=dtype, seed=3), _r(rows, C, dtype=dtype, seed=4), w, mean, rstd)
        if ops.layernorm_bwd_sums_ok(C):
            ops.layernorm_bwd_sums(x, _r(rows, C, dtype=dtype, seed=3), _r(rows, C, dtype=dtype, seed=4), w, mean, rstd)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(993, 1600, 6400), (130, 70, 200), (10080, 1600, 1600), (62, 4 * 256, 256), (1, 591, 1024)])
def test_gemm_family_stays_in_bounds(dtype, M, N, K):
    from multimodal_amd import ops
    if dtype == torch.float32 and M * N * K > 3e9:
        pytest.skip("exact-f32 GEMM at this size adds nothing to the bounds check")
    a, w = _r(M, K, dtype=dtype, scale=0.1), _r(N, K, dtype=dtype, seed=1, scale=0.1)
    bias = _r(N, seed=2)
    res = _r(M, N, dtype=dtype, seed=3)
    with canary():
        ops.gemm(a, w, M, N, K, bias=bias, res=res)
        pre = ops.torch.empty((M, N), dtype=dtype, device=DEV)
        ops.gemm(a, w, M, N, K, bias=bias, pre=pre, act=ops.ACT_GELU)
        ops.gemm(a, w.t().contiguous(), M, N, K, b_kmajor=True)
        ops.gemm(a, a, K, K, M, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, split_k=4) if K <= 256 else None
        ops.colsum(res, M, N)
        if (N * res.element_size()) % 16 == 0:
            ops.relu2_bwd_colsum_(res.clone(), _r(M, N, dtype=dtype, seed=4))
        ops.relu2_fwd(res)
        ops.argmax_rows(res.float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,d,S,P,sm", [(2, 16, 100, 315, 221, True), (3, 8, 128, 238, 159, False), (1, 4, 192, 77, 40, True),
                                          (1, 4, 512, 45, 36, True), (2, 4, 32, 19, 5, True), (2, 2, 100, 1, 0, True)])
def test_attention_stays_in_bounds(dtype, B, H, d, S, P, sm):
    from multimodal_amd import ops
    E = H * d
    rot = int(d * 0.5)
    qkv = _r(B, S, 3, E, dtype=dtype)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, device=DEV).float() / rot))
    cos_t, sin_t = ops.rope_tables(inv, S + 8)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    with canary():
        o, lse = ops.prefix_attn_fwd(q, k, v, H, rot, P, cos_t, sin_t, sm, save_lse=True)        # in-kernel RoPE form
        ops.prefix_attn_bwd(q, k, v, o, _r(B, S, E, dtype=dtype, seed=1), lse, H, rot, P, cos_t, sin_t, sm)
        o, lse = ops.prefix_attn_fwd(q, k, v, H, 0, P, cos_t, sin_t, sm, save_lse=True)          # pre-rotated form
        ops.prefix_attn_bwd(q, k, v, o, _r(B, S, E, dtype=dtype, seed=1), lse, H, 0, P, cos_t, sin_t, sm)
        Smax = S + 3
        kc = ops.torch.empty((B, H, Smax, d), dtype=dtype, device=DEV)
        vc = ops.torch.empty((B, H, Smax, d), dtype=dtype, device=DEV)
        ops.kv_cache_fill(k, v, kc, vc, H, rot, cos_t, sin_t)
        step = _r(B, 3 * E, dtype=dtype, seed=2)
        for pos in (S, Smax - 1):
            ops.decode_attn(step[:, :E], step[:, E:2 * E], step[:, 2 * E:], kc, vc, H, rot, pos, cos_t, sin_t, sm)


# ------------------------------------------------------------------------------------------------ determinism
def _twice(fn):
    a = fn()
    torch.cuda.synchronize()
    b = fn()
    torch.cuda.synchronize()
    a = a if isinstance(a, (tuple, list)) else (a,)
    b = b if isinstance(b, (tuple, list)) else (b,)
    for i, (x, y) in enumerate(zip(a, b)):
        if x is not None:
            assert torch.equal(x, y), f"output {i} differs between two identical launches"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_same_input_twice_is_bit_equal(dtype):
    from multimodal_amd import ops
    # K1b at a shape with many blocks per parameter (cross-block reduction of dw, db, dln_w, dln_b)
    B, C, Ln = 64, 128, 125
    x = _r(B, Ln, C, dtype=dtype)
    w, b, lw, lb = _r(C, 1, 7, seed=1), _r(C, seed=2), _r(C, seed=3) + 1.5, _r(C, seed=4)
    y, mean, rstd = ops.dwconv7_ln_fwd(x, w, b, lw, lb, save_stats=True)
    dy, dres = _r(B, Ln, C, dtype=dtype, seed=5), _r(B, Ln, C, dtype=dtype, seed=6)
    _twice(lambda: tuple(t.clone() for t in ops.dwconv7_ln_bwd(x, dy, dres, w, b, lw, mean, rstd)))
    for Cw in (1600,):
        xw = _r(8, 31, Cw, dtype=dtype)
        ww, bw, lww, lbw = _r(Cw, 1, 7, seed=1), _r(Cw, seed=2), _r(Cw, seed=3) + 1.5, _r(Cw, seed=4)
        _, mw, rw = ops.dwconv7_ln_fwd(xw, ww, bw, lww, lbw, save_stats=True)
        dyw = _r(8, 31, Cw, dtype=dtype, seed=5)
        _twice(lambda: tuple(t.clone() for t in ops.dwconv7_ln_bwd(xw, dyw, None, ww, bw, lww, mw, rw)))
    # K3b at the config head shape
    Bq, H, d, S, P = 4, 16, 100, 315, 221
    E = H * d
    qkv = _r(Bq, S, 3, E, dtype=dtype)
    inv = 1.0 / (10000 ** (torch.arange(0, 50, 2, device=DEV).float() / 50))
    cos_t, sin_t = ops.rope_tables(inv, S)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    for sm in (True, False):
        o, lse = ops.prefix_attn_fwd(q, k, v, H, 0, P, cos_t, sin_t, sm, save_lse=True)
        do = _r(Bq, S, E, dtype=dtype, seed=7)
        _twice(lambda: ops.prefix_attn_bwd(q, k, v, o, do, lse, H, 0, P, cos_t, sin_t, sm).clone())
    # K4 (cache append + attend): same step on two copies of the same cache
    Smax = 64
    kc0, vc0 = _r(8, H, Smax, d, dtype=dtype, seed=8), _r(8, H, Smax, d, dtype=dtype, seed=9)
    step = _r(8, 3 * E, dtype=dtype, seed=10)

    def k4():
        kc, vc = kc0.clone(), vc0.clone()
        o = ops.decode_attn(step[:, :E], step[:, E:2 * E], step[:, 2 * E:], kc, vc, H, 50, 40, cos_t, sin_t, True)
        return o.clone(), kc, vc
    _twice(k4)
    # LayerNorm backward (wide and narrow rows), split-K GEMM, column sums
    for rows, Cc in ((2016, 1600), (4000, 256)):
        xl, dyl, dr = _r(rows, Cc, dtype=dtype), _r(rows, Cc, dtype=dtype, seed=1), _r(rows, Cc, dtype=dtype, seed=2)
        wl, bl = _r(Cc, seed=3) + 1.5, _r(Cc, seed=4)
        _, ml, rl = ops.layernorm_fwd(xl, wl, bl, save_stats=True)
        _twice(lambda: tuple(t.clone() for t in ops.layernorm_bwd(xl, dyl, dr, wl, ml, rl)))
    a, bm = _r(4000, 256, dtype=dtype, scale=0.1), _r(4000, 1024, dtype=dtype, seed=1, scale=0.1)
    _twice(lambda: ops.gemm(a, bm, 256, 1024, 4000, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, split_k=8).clone())
    _twice(lambda: ops.colsum(bm, 4000, 1024).clone())
