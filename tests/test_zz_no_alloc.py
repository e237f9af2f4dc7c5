"""The allocation-free GEMM entry point (otter_gemm_static_ws: split-K scratch in a static device array) against
otter_gemm_ws with a caller-provided workspace: bit-identical outputs, on two calls in a row (the second one smaller,
so stale partial products of the first call sit in the scratch), plus the size limit and the no-allocation property."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from otter_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _problem(dev, dt, m, n, k, seed):
    from otter_amd import ops
    a = (_randn(seed, m, k) * 0.5).to(dev).to(dt)
    w = (_randn(seed + 1, n, k) / math.sqrt(k)).to(dev)
    bias = _randn(seed + 2, n).to(dev)
    odt = dt
    res = _randn(seed + 3, m, n).to(dev).to(odt)
    return a, ops.pack_weight(w, dt), bias, res


def _out(dev, dt, m, n):
    return torch.full((m, n), float("nan"), dtype=dt, device=dev)


def _run(name, dev, dt, a, wp, bias, res, m, n, k, epi, ws=None, out=None):
    """One call through the C ABI into `out` (allocated by the test, not by the library)."""
    from otter_amd import _lib, ops
    if out is None:
        out = _out(dev, dt, m, n)
    args = [ops._dt(dt, "t"), ops._ptr(a), k, ops._ptr(wp), wp.shape[0], ops._ptr(bias), ops._ptr(res if epi & 8 else None), n,
            None, None, ops._ptr(out), n, m, n, k, epi]
    if name == "otter_gemm_ws":
        args += [ops._ptr(ws), ws.numel()]
    args.append(ops._stream())
    _lib.call(name, *args)
    return out


@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
def test_static_scratch_gemm_matches_workspace_gemm_twice(dev, dtype):
    from otter_amd import _lib
    lib = _lib.load()
    dt = DTYPES[dtype]
    # (M, N, K): first a split-K problem that fills a large part of the scratch, then a smaller one; then the remainder
    # rows of a big 16-bit problem (whole 256-row tiles + 1 row split-K)
    shapes = [(128, 1024, 1024), (64, 512, 512)]
    if dtype != "f32":
        shapes.append((8193, 1024, 1024))
    first = True
    for epi in (0, 1, 8):                                     # none, QuickGELU, residual
        for i, (m, n, k) in enumerate(shapes):
            a, wp, bias, res = _problem(dev, dt, m, n, k, 100 + 10 * i)
            ws = torch.empty(lib.otter_gemm_workspace_floats(n), dtype=torch.float32, device=dev)
            ref = _run("otter_gemm_ws", dev, dt, a, wp, bias, res, m, n, k, epi, ws)
            got = _out(dev, dt, m, n)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info(dev)[0]     # hipMemGetInfo: sees every allocation on the device, torch's or not
            _run("otter_gemm_static_ws", dev, dt, a, wp, bias, res, m, n, k, epi, out=got)
            torch.cuda.synchronize()
            if not first:                                     # (the first call on a device may load the code object lazily)
                assert torch.cuda.mem_get_info(dev)[0] == free_before, "the call changed the device's free memory"
            first = False
            assert torch.isfinite(got.float()).all()
            assert torch.equal(got, ref), (dtype, epi, m, n, k, float((got.float() - ref.float()).abs().max()))
    # and against fp32 torch on the same rounded operands (tolerance: one rounding of the 16-bit output)
    m, n, k = shapes[0]
    a, wp, bias, res = _problem(dev, dt, m, n, k, 100)
    got = _run("otter_gemm_static_ws", dev, dt, a, wp, bias, res, m, n, k, 0)
    want = a.float() @ wp[:n].float().t() + bias
    tol = {"bf16": 2e-2, "f16": 2e-3, "f32": 1e-4}[dtype]
    torch.testing.assert_close(got.float(), want, rtol=tol, atol=tol)


def test_static_scratch_gemm_under_graph_capture_after_init(dev):
    """After otter_gemm_static_ws_init() (the per-device symbol look-up, outside any capture) the entry point runs under
    HIP graph capture: a captured split-K product replays to the bits of the eager call, on fresh inputs."""
    from otter_amd import _lib
    lib = _lib.load()
    assert lib.otter_gemm_static_ws_init() == 0 and lib.otter_gemm_static_ws_init() == 0       # idempotent
    dt = torch.float16
    m, n, k = 128, 1024, 1024
    a, wp, bias, res = _problem(dev, dt, m, n, k, 300)
    a_static, out = a.clone(), _out(dev, dt, m, n)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        _run("otter_gemm_static_ws", dev, dt, a_static, wp, bias, res, m, n, k, 1, out=out)     # once eagerly: LDS opt-ins exist before capture
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(dev)[0]
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        _run("otter_gemm_static_ws", dev, dt, a_static, wp, bias, res, m, n, k, 1, out=out)
    for seed in (301, 302):
        a2, _, _, _ = _problem(dev, dt, m, n, k, seed)
        a_static.copy_(a2)
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        ws = torch.empty(lib.otter_gemm_workspace_floats(n), dtype=torch.float32, device=dev)
        ref = _run("otter_gemm_ws", dev, dt, a2, wp, bias, res, m, n, k, 1, ws)
        assert torch.equal(out, ref)
    del ws, ref
    # a stream of another device than the current one is refused (only checkable with two GPUs)
    if torch.cuda.device_count() > 1:
        other = torch.cuda.Stream(device=torch.device("cuda", 1))
        rc = lib.otter_gemm_static_ws(1, ctypes.c_void_p(a.data_ptr()), k, ctypes.c_void_p(wp.data_ptr()), wp.shape[0], None, None, 0, None, None,
                                      ctypes.c_void_p(out.data_ptr()), n, m, n, k, 0, ctypes.c_void_p(other.cuda_stream))
        assert rc != 0 and b"belongs to device" in lib.otter_last_error()
    assert free_before >= 0


def test_static_scratch_gemm_rejects_oversize(dev):
    from otter_amd import _lib
    lib = _lib.load()
    nmax = lib.otter_gemm_static_ws_max_n()
    assert nmax == 4096
    buf = torch.zeros(1 << 20, dtype=torch.float16, device=dev)
    p = ctypes.c_void_p(buf.data_ptr())
    rc = lib.otter_gemm_static_ws(1, p, 64, p, 8192, None, None, 0, None, None, p, 8192, 4, nmax + 8, 64, 0, None)
    assert rc == 2 and b"exceeds the static scratch" in lib.otter_last_error()
