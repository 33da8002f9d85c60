"""mm_colsum_noalloc (the entry point that allocates nothing: its partial sums live in a device array built into
libmmhip.so) against mm_colsum, through the C ABI: bit-identical sums on two calls in a row, the second smaller,
plus the size limit.  Collected last (file name) so a failure here cannot hide another test under `pytest -x`."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(x):
    from multimodal_amd import _lib as L
    M, N = x.shape
    lib = L.lib()
    out_ref = torch.full((N,), float("nan"), dtype=torch.float32, device=x.device)
    out_new = torch.full((N,), float("nan"), dtype=torch.float32, device=x.device)
    nbytes = lib.mm_colsum_workspace(M, N)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.mm_colsum(x.data_ptr(), out_ref.data_ptr(), ws.data_ptr(), nbytes, M, N, x.stride(0), L.dtype_code(x), st)
    assert rc == 0, lib.mm_last_error()
    rc = lib.mm_colsum_noalloc(x.data_ptr(), out_new.data_ptr(), M, N, x.stride(0), L.dtype_code(x), st)
    assert rc == 0, lib.mm_last_error()
    torch.cuda.synchronize()
    return out_ref, out_new


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_noalloc_colsum_equals_colsum_twice_second_smaller(dtype):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    # first call: the decoder's bias-gradient shape (10080 rows x 4*1600); second: smaller in both dimensions, odd N
    for M, N in ((10080, 6400), (993, 1601), (31, 64)):
        x = (torch.rand(M, N, generator=g) * 2 - 1).to(dtype).to(dev)
        ref, new = _both(x)
        assert torch.equal(ref, new), f"{dtype} {M}x{N}: max diff {(ref - new).abs().max().item()}"
        want = x.double().sum(0)
        assert float((new.double() - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max()))


def test_noalloc_colsum_allocates_nothing_and_rejects_oversize():
    from multimodal_amd import _lib as L
    from multimodal_amd import ops
    dev = torch.device("cuda:0")
    x = torch.rand(4096, 1600, device=dev)
    out = torch.empty(1600, device=dev)
    ops.colsum_noalloc(x, out)                        # first use on this device resolves the array's address
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    n0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    for _ in range(3):
        ops.colsum_noalloc(x, out)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == n0          # torch's allocator not used
    assert free1 == free0                                                       # nor the driver's
    assert torch.equal(out, ops.colsum(x, 4096, 1600))
    big = torch.empty(8, 8200, device=dev)
    rc = L.lib().mm_colsum_noalloc(big.data_ptr(), out.data_ptr(), 8, 8200, 8200, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == -1 and b"8192" in L.lib().mm_last_error()


def test_noalloc_colsum_under_graph_capture():
    """The caller it exists for: a captured region, where allocation is not allowed."""
    from multimodal_amd import ops
    dev = torch.device("cuda:0")
    x = torch.rand(2048, 768, device=dev)
    out = torch.zeros(768, device=dev)
    ops.colsum_noalloc(x, out)                        # address lookup outside the capture
    want = ops.colsum(x, 2048, 768)
    out.zero_()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        g.capture_begin()
        try:
            ops.colsum_noalloc(x, out)
        finally:
            g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
