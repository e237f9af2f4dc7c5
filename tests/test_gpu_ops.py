"""Per-kernel parity on the GPU: each C-ABI entry point against a plain fp32 torch / oracle computation of the
same op on the same (bf16-rounded) inputs.  Tolerances are written next to each check."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import otter_ref as R  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from otter_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _randn(seed, *shape):
    return torch.randn(*shape, generator=_g(seed))


def _act(name, x):
    if name == "quickgelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return torch.nn.functional.gelu(x)
    if name == "relu":
        return torch.relu(x)
    return x


ACTS = {"none": 0, "quickgelu": 1, "gelu": 2, "relu": 3}


DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


@pytest.mark.parametrize("m,n,k", [(300, 256, 128), (1000, 3072, 1024), (129, 120, 1536), (5, 768, 704), (2056, 1024, 4096),
                                   (1, 128, 64), (64, 2304, 768), (1500, 1000, 512), (1024, 256, 64)])
@pytest.mark.parametrize("act", ["none", "quickgelu", "gelu", "relu"])
@pytest.mark.parametrize("out_f32", [False, True])
@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
def test_gemm(dev, m, n, k, act, out_f32, dtype):
    from otter_amd import ops
    dt = DTYPES[dtype]
    if dtype == "f32" and not out_f32:
        pytest.skip("fp32 operands always write fp32")
    a = _randn(1, m, k).to(dev).to(dt)
    w = (_randn(2, n, k) / math.sqrt(k)).to(dev)
    bias = _randn(3, n).to(dev)
    wp = ops.pack_weight(w, dt)
    kp = wp.shape[1]
    if kp != k:
        a_p = torch.zeros(m, kp, dtype=dt, device=dev)
        a_p[:, :k] = a
    else:
        a_p = a
    out = ops.gemm(a_p, wp, n, bias, act=ACTS[act], out_f32=out_f32)
    ref = _act(act, a.double() @ w.to(dt).double().t() + bias.double()).float()
    assert out.shape == (m, n)
    if dtype == "f32":
        # exact fp32 fma chain in a different summation order than the fp64 reference: ~1e-6 relative to sum |a||w|
        tol = dict(rtol=2e-5, atol=2e-5)
    elif out_f32:
        # fp32 accumulation of exact 16-bit products: summation order only
        tol = dict(rtol=2e-3, atol=2e-3)
    else:
        tol = dict(rtol=1e-2, atol=1e-2) if dtype == "bf16" else dict(rtol=2e-3, atol=2e-3)   # output rounding 2^-9 / 2^-12
    torch.testing.assert_close(out.float(), ref, **tol)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("m,n,k", [(300, 256, 128), (129, 120, 1536), (1000, 3072, 1024)])
def test_gemm_128_tile_kernel(dev, m, n, k, dtype):
    """The 128x128 kernel on shapes the dispatcher would give to the 64x64 / 256x256 kernels: same results."""
    from otter_amd import ops
    dt = DTYPES[dtype]
    a = _randn(11, m, k).to(dev).to(dt)
    w = (_randn(12, n, k) / math.sqrt(k)).to(dev)
    bias = _randn(13, n).to(dev)
    wp = ops.pack_weight(w, dt)
    res = _randn(14, m, n).to(dev)
    for kw in (dict(act=ACTS["gelu"]), dict(residual=res)):
        ref = ops.gemm(a, wp, n, bias, split_k=False, **kw)
        out = ops.gemm(a, wp, n, bias, force_tile_128=True, **kw)
        assert torch.equal(out, ref)              # same products, same fp32 accumulation order along K (one-pass kernels)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("m,n,k", [(300, 256, 128), (32896 // 8, 1024, 1024), (130, 64, 64)])
def test_gemm_residual_in_operand_type(dev, m, n, k, dtype):
    """16-bit residual stream: out = round16(a @ w^T + bias + residual), in place, on all three kernels."""
    from otter_amd import ops
    dt = DTYPES[dtype]
    a = _randn(31, m, k).to(dev).to(dt)
    w = (_randn(32, n, k) / math.sqrt(k)).to(dev)
    bias = _randn(33, n).to(dev)
    wp = ops.pack_weight(w, dt)
    x = (_randn(34, m, n) * 2).to(dev).to(dt)
    ref = a.float() @ w.to(dt).float().t() + bias + x.float()
    for kw in ({}, dict(force_tile_128=True)):
        xs = x.clone()
        out = ops.gemm(a, wp, n, bias, residual=xs, out=xs, **kw)
        assert out.dtype == dt and out.data_ptr() == xs.data_ptr()
        tol = 2e-2 if dtype == "bf16" else 3e-3                          # one rounding of |out| <~ 8 to the 16-bit type
        torch.testing.assert_close(out.float(), ref, rtol=tol, atol=tol)
    with pytest.raises(ValueError):
        ops.gemm(a, wp, n, bias, residual=x.to(torch.float16 if dt == torch.bfloat16 else torch.bfloat16))


def test_gemm_256_erf_gelu_epilogue_accuracy(dev):
    """The 256x256 kernel's erf-GELU epilogue is a branch-free erfc approximation (OpenCLIP tower MLPs): through an
    identity weight and an fp32 output it must match nn.GELU() to 1e-6 * max(1, |x|) over the whole fp16-exact grid."""
    from otter_amd import ops
    m, n = 32768, 256
    x = torch.linspace(-9.0, 9.0, m * n, device=dev).view(m, n).half()
    wp = ops.pack_weight(torch.eye(n, device=dev), torch.float16)
    out = ops.gemm(x, wp, n, None, act=ACTS["gelu"], out_f32=True)
    ref = torch.nn.functional.gelu(x.double())
    err = (out.double() - ref).abs() / x.double().abs().clamp_min(1.0)
    assert err.max().item() <= 1e-6, err.max().item()
    # deep in the left tail the result keeps its sign and magnitude (no cancellation against 1)
    tail = x < -5
    assert (out[tail] <= 0).all() and torch.allclose(out[tail].double(), ref[tail], rtol=0.05, atol=1e-12)


@pytest.mark.parametrize("m,n,k", [(32768 // 4, 1024, 256), (300, 256, 128), (70, 64, 64)])
def test_gemm_fp16_output_saturates(dev, m, n, k):
    """fp16 outputs beyond +-65504 clamp to the largest finite value (never an infinity) on all three kernels, with and
    without the 16-bit residual; values inside the range are untouched by the clamp."""
    from otter_amd import ops
    a = torch.ones(m, k, device=dev, dtype=torch.float16)
    w = torch.zeros(n, k)
    w[:, 0] = torch.linspace(-1.0, 1.0, n)                          # a row's products sum to w[j, 0]
    bias = torch.zeros(n)
    bias[0::4], bias[1::4], bias[2::4] = 7.0e4, -7.0e4, 65504.0      # over, under, exactly the limit; the rest small
    w, bias = w.to(dev), bias.to(dev)
    wp = ops.pack_weight(w, torch.float16)
    x = (_randn(35, m, n) * 4).to(dev).to(torch.float16)
    for res in (None, x):
        for kw in ({}, dict(force_tile_128=True)):
            out = ops.gemm(a, wp, n, bias, residual=None if res is None else res.clone(), **kw)
            ref = a.float() @ w.to(torch.float16).float().t() + bias + (0 if res is None else res.float())
            want = ref.clamp(-65504.0, 65504.0).to(torch.float16)
            assert torch.isfinite(out.float()).all()
            assert int((ref.abs() > 65504).sum()) > m * n // 8       # the case really overflows
            big = ref.abs() > 6.0e4
            assert torch.equal(out[big], want[big])
            torch.testing.assert_close(out.float(), want.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("m,n,k,act", [(32896 // 4, 1024, 1024, "none"), (300, 256, 128, "quickgelu"), (70, 64, 64, "none")])
def test_gemm_with_folded_layernorm(dev, m, n, k, act, dtype):
    """LayerNorm folded into the GEMM epilogue == LayerNorm in fp32 followed by the GEMM, on all three kernels
    (256x256 tiles + 64x64 remainder rows, 128x128, 64x64)."""
    from otter_amd import ops
    dt = DTYPES[dtype]
    x = (_randn(41, m, k) * 3 + 0.7).to(dev).to(dt)                      # the residual stream: rows with a common offset
    w = (_randn(42, n, k) / math.sqrt(k)).to(dev)
    bias = _randn(43, n).to(dev)
    gamma = (_randn(44, k) * 0.3 + 1).to(dev)
    beta = (_randn(45, k) * 0.2).to(dev)
    wp, bf, cs = ops.fold_layernorm(w, bias, gamma, beta, dt, dev)
    stats = ops.row_stats(x)
    xf = x.float()
    torch.testing.assert_close(stats[:, 0], xf.mean(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stats[:, 1], (xf.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-5, atol=1e-5)
    ref = torch.nn.functional.layer_norm(xf, (k,), gamma, beta, 1e-5) @ w.t() + bias
    if act == "quickgelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    for kw in ({}, dict(force_tile_128=True)):
        out = ops.gemm(x, wp, n, bf, act=ACTS[act], ln_stats=stats, ln_colsum=cs, **kw)
        tol = 3e-2 if dtype == "bf16" else 4e-3                          # rounding of W * gamma and of the output
        torch.testing.assert_close(out.float(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("m,n,k", [(16384, 1280, 256),      # 320 tiles on 256 CUs: workgroups with two tiles beside workgroups with one
                                   (32768, 1000, 320),      # N not a multiple of 256 (lanes beyond N), odd number of K-steps
                                   (33024, 768, 1024),      # 129 x 3 = 387 tiles: an uneven second round
                                   (8192, 1024, 256)])      # 128 tiles: every workgroup's stream ends inside its first tile
def test_gemm_256_tile_stream_shapes(dev, m, n, k):
    """The 256x256 kernel's load stream runs across a workgroup's tiles: shapes where the tile counts per workgroup differ,
    the stream ends at different points and the column tiles are partial -- plain, folded LayerNorm + QuickGELU, in-place
    16-bit residual and fp32 output, each against a GPU fp32 product of the same 16-bit operands."""
    from otter_amd import ops
    dt = torch.float16
    x = (_randn(51, m, k) * 2 + 0.3).to(dev).to(dt)
    w = (_randn(52, n, k) / math.sqrt(k)).to(dev)
    bias = _randn(53, n).to(dev)
    wp = ops.pack_weight(w, dt)
    prod = x.float() @ wp[:n, :k].float().t() + bias
    out = ops.gemm(x, wp, n, bias)
    torch.testing.assert_close(out.float(), prod, rtol=2e-3, atol=2e-3)
    out32 = ops.gemm(x, wp, n, bias, act=ACTS["relu"], out_f32=True)
    torch.testing.assert_close(out32, torch.relu(prod), rtol=1e-4, atol=1e-4)
    res = (_randn(54, m, n) * 2).to(dev).to(dt)
    xs = res.clone()
    ops.gemm(x, wp, n, bias, residual=xs, out=xs)
    torch.testing.assert_close(xs.float(), prod + res.float(), rtol=3e-3, atol=3e-3)
    gamma = (_randn(55, k) * 0.3 + 1).to(dev)
    beta = (_randn(56, k) * 0.2).to(dev)
    wf, bfold, cs = ops.fold_layernorm(w, bias, gamma, beta, dt, dev)
    ref = torch.nn.functional.layer_norm(x.float(), (k,), gamma, beta, 1e-5) @ w.t() + bias
    ref = ref * torch.sigmoid(1.702 * ref)
    out = ops.gemm(x, wf, n, bfold, act=ACTS["quickgelu"], ln_stats=ops.row_stats(x), ln_colsum=cs)
    torch.testing.assert_close(out.float(), ref, rtol=4e-3, atol=4e-3)
    # the same launches again on the same buffers: nothing left over from the previous launch's stream
    out2 = ops.gemm(x, wf, n, bfold, act=ACTS["quickgelu"], ln_stats=ops.row_stats(x), ln_colsum=cs)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("m,n,k", [(128, 1024, 4096), (32896 // 2, 1024, 1024), (100, 3072, 1024), (70, 64, 512)])
def test_gemm_split_k_matches_one_pass(dev, m, n, k, dtype):
    """Few-tile 16-bit problems (and the remainder rows of big ones) are multiplied split-K through a scratch buffer:
    same epilogues, results equal to the one-pass kernel up to the fp32 summation order."""
    from otter_amd import ops, _lib
    dt = DTYPES[dtype]
    a = (_randn(51, m, k) * 0.5).to(dev).to(dt)
    w = (_randn(52, n, k) / math.sqrt(k)).to(dev)
    bias = _randn(53, n).to(dev)
    wp = ops.pack_weight(w, dt)
    x = _randn(54, m, n).to(dev).to(dt)
    lib = _lib.load()
    def one_pass(act, residual):
        out = torch.empty(m, n, dtype=dt, device=dev)
        epi = act | (8 if residual is not None else 0)
        _lib.call("otter_gemm", ops._dt(dt, "t"), ops._ptr(a), k, ops._ptr(wp), wp.shape[0], ops._ptr(bias), ops._ptr(residual), n,
                  ops._ptr(out), n, m, n, k, epi, ops._stream())
        return out
    for act, residual in ((ACTS["none"], None), (ACTS["quickgelu"], None), (ACTS["none"], x)):
        got = ops.gemm(a, wp, n, bias, act=act, residual=residual)          # the wrapper passes the scratch buffer
        ref = one_pass(act, residual)
        tol = 2e-2 if dtype == "bf16" else 2e-3                           # one 16-bit rounding apart at most
        torch.testing.assert_close(got.float(), ref.float(), rtol=tol, atol=tol)
    assert lib.otter_gemm_workspace_floats(n) >= 8 * 255 * n


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("m,n,k", [(32896, 1024, 1024),      # out_proj of a 128-image step: 128 big row tiles + 128 split-K remainder rows
                                   (32896 + 64, 1024, 4096),  # c_proj: K = 4096, 192 remainder rows
                                   (33024, 768, 1024),        # whole tiles only
                                   (514, 1024, 1024),         # a single observation: no big tiles
                                   (32896, 264, 512)])        # a width that leaves the second vector of most lanes empty
def test_gemm_residual_stats_matches_the_two_launches(dev, dtype, m, n, k):
    """x += a W^T + bias and the row statistics of the result from one entry point: bit for bit the residual product followed by
    otter_row_stats, whether the remainder rows' second pass is fused with the statistics or not."""
    from otter_amd import ops
    dt = DTYPES[dtype]
    a = (_randn(40, m, k) * 0.5).to(dev).to(dt)
    wp = ops.pack_weight(_randn(41, n, k).to(dev) / math.sqrt(k), dt)
    bias = _randn(42, n).to(dev)
    x0 = (_randn(43, m, n) * 3).to(dev).to(dt)
    x0[:, 5] *= 40.0                                   # an outlier channel, as the real streams have
    want = ops.gemm(a, wp, n, bias, residual=x0.clone())
    want_stats = ops.row_stats(want)
    x = x0.clone()
    stats = ops.gemm_residual_stats_(a, wp, n, bias, x)
    assert torch.equal(x, want)
    assert torch.equal(stats, want_stats)
    # and in a row-strided view of a wider stream
    wide = torch.zeros(m, n + 16, dtype=dt, device=dev)
    wide[:, :n] = x0
    stats2 = ops.gemm_residual_stats_(a, wp, n, bias, wide[:, :n])
    assert torch.equal(wide[:, :n], want) and torch.equal(stats2, want_stats) and torch.all(wide[:, n:] == 0)


@pytest.mark.parametrize("m", [777, 1300])          # 128-row tiles only / 256-row tiles plus a 128-row-tile remainder
def test_gemm_residual_strided_inplace(dev, m):
    from otter_amd import ops
    n, k = 1024, 1024
    big_a = torch.zeros(m, k + 64, dtype=torch.bfloat16, device=dev)
    a = big_a[:, :k]
    a.copy_(_randn(4, m, k))
    w = (_randn(5, n, k) / math.sqrt(k)).to(dev)
    bias = _randn(6, n).to(dev)
    res = _randn(7, m, n).to(dev)
    ref = a.float() @ w.bfloat16().float().t() + bias + res
    out = ops.gemm(a, ops.pack_weight(w), n, bias, residual=res, out=res)      # in place on the residual stream (bf16 operands)
    assert out.data_ptr() == res.data_ptr()
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-3)
    # rows of an output with a larger leading dimension stay untouched outside [0, n)
    wide = torch.full((m, n + 8), 7.0, device=dev)
    ops.gemm(a, ops.pack_weight(w), n, bias, out_f32=True, out=wide[:, :n])
    assert torch.all(wide[:, n:] == 7.0)


def test_gemm_rejects_bad_arguments(dev):
    from otter_amd import ops
    a = torch.zeros(4, 100, dtype=torch.bfloat16, device=dev)     # K not a multiple of 64
    w = torch.zeros(256, 100, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ops.OtterHipError):
        ops.gemm(a, w, 128)
    with pytest.raises(ops.OtterHipError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(256, 64, dtype=torch.bfloat16, device=dev), 128)   # CPU tensor: no fallback
    with pytest.raises(TypeError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.float16, device=dev), torch.zeros(256, 64, dtype=torch.bfloat16, device=dev), 128)
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16, device=dev), torch.zeros(128, 64, dtype=torch.bfloat16, device=dev), 128)   # unpadded weight


@pytest.mark.parametrize("rows,width", [(37, 1024), (1, 768), (1000, 768), (9, 128), (5, 4096), (3, 64)])
@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
def test_layernorm(dev, rows, width, dtype):
    from otter_amd import ops
    x = (_randn(8, rows, width) * 3 + 1).to(dev)
    g = (_randn(9, width) * 0.2 + 1).to(dev)
    b = _randn(10, width).to(dev)
    y = ops.layernorm(x, g, b, out_dtype=DTYPES[dtype])
    ref = torch.nn.functional.layer_norm(x, (width,), g, b, 1e-5)
    tol = {"f32": 1e-5, "bf16": 8e-3, "f16": 1e-3}[dtype]                 # rounding of the output type
    torch.testing.assert_close(y.float(), ref, rtol=tol, atol=tol)
    y2 = ops.layernorm(x, None, None, out_dtype=torch.float32)
    torch.testing.assert_close(y2, torch.nn.functional.layer_norm(x, (width,)), rtol=1e-5, atol=1e-5)
    if dtype != "f32":                                                     # 16-bit input (residual stream): statistics in fp32
        x16 = x.to(DTYPES[dtype])
        y3 = ops.layernorm(x16, g, b, out_dtype=torch.float32)
        torch.testing.assert_close(y3, torch.nn.functional.layer_norm(x16.float(), (width,), g, b, 1e-5), rtol=1e-5, atol=1e-5)


def test_vit_embed_and_text_embed(dev):
    from otter_amd import ops
    n_img, p, w = 3, 16, 128
    tok = _randn(11, n_img * p, w).to(dev)
    cls = _randn(12, w).to(dev)
    pos = _randn(13, p + 1, w).to(dev)
    g = (_randn(14, w) * 0.1 + 1).to(dev)
    b = _randn(15, w).to(dev)
    x = ops.vit_embed_lnpre(tok, cls, pos, g, b, n_img, p)
    full = torch.cat([cls.expand(n_img, 1, w), tok.view(n_img, p, w)], 1) + pos
    ref = torch.nn.functional.layer_norm(full, (w,), g, b, 1e-5).view(-1, w)
    torch.testing.assert_close(x, ref, rtol=1e-5, atol=1e-5)

    ids = torch.randint(0, 500, (4, 77), generator=_g(16)).int().to(dev)
    table = _randn(17, 500, 64).to(dev)
    tpos = _randn(18, 77, 64).to(dev)
    for take in (15, 77, 1):
        e = ops.text_embed(ids, take, table, tpos)
        ref = (table[ids[:, :take].long()] + tpos[:take]).view(-1, 64)
        torch.testing.assert_close(e, ref, rtol=0, atol=0)


# (res 42 is no multiple of 4: the uint8 path falls back from the LDS-staged kernel to the element-wise one)
@pytest.mark.parametrize("res,ps", [(56, 14), (64, 32), (48, 16), (42, 14), (224, 14)])
@pytest.mark.parametrize("dtype", ["u8", "f32"])
def test_im2col(dev, dtype, res, ps):
    from otter_amd import ops
    n = 3
    if dtype == "u8":
        img = torch.randint(0, 256, (n, 3, res, res), generator=_g(19), dtype=torch.uint8).to(dev)
        ref_img = R.clip_transform(img.cpu()).to(dev)
    else:
        img = _randn(20, n, 3, res, res).to(dev)
        ref_img = img
    kpad = ops.round_up(3 * ps * ps, 64)
    ref = torch.nn.functional.unfold(ref_img, ps, stride=ps).transpose(1, 2).reshape(-1, 3 * ps * ps)
    for dt in (torch.bfloat16, torch.float16):
        out = ops.im2col_patches(img, ps, kpad, R.CLIP_MEAN, R.CLIP_STD, dt)
        assert torch.all(out[:, 3 * ps * ps:] == 0)
        # bit-exact after rounding the fp32 normalised pixel to the 16-bit type
        assert torch.equal(out[:, :3 * ps * ps], ref.to(dt))


def _sdpa_ref(qkv, n_seq, L, heads, hd, causal):
    q, k, v = qkv.float().view(n_seq, L, 3, heads, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=qkv.device).triu(1)
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(n_seq * L, heads * hd)


@pytest.mark.parametrize("n_seq,L,heads,causal", [(3, 257, 16, False), (2, 15, 12, True), (2, 77, 12, True), (5, 17, 2, False),
                                                  (1, 1, 1, True), (2, 65, 2, False), (1, 288, 1, False), (2, 577, 2, False), (1, 592, 1, True), (2, 16, 3, True),
                                                  (2, 40, 2, True), (1, 197, 12, False)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_attn_d64(dev, n_seq, L, heads, causal, dtype):
    from otter_amd import ops
    qkv = _randn(21, n_seq * L, 3 * heads * 64).to(dev).to(DTYPES[dtype])
    out = ops.attn_d64(qkv, n_seq, L, heads, causal)
    ref = _sdpa_ref(qkv, n_seq, L, heads, 64, causal)
    # P is rounded to the 16-bit type before P.V and so is the output: ~2^-8 (bf16) / 2^-11 (fp16) relative on O(1) values
    tol = dict(rtol=2e-2, atol=1.5e-2) if dtype == "bf16" else dict(rtol=3e-3, atol=2e-3)
    torch.testing.assert_close(out.float(), ref, **tol)


def test_attn_d64_peaked_softmax(dev):
    """Large logits (one dominant key per query) must not overflow or lose the row maximum."""
    from otter_amd import ops
    n_seq, L, heads = 1, 257, 2
    qkv = _randn(22, n_seq * L, 3 * heads * 64).to(dev)
    qkv[:, : heads * 64] *= 12.0
    qkv = qkv.bfloat16()
    out = ops.attn_d64(qkv, n_seq, L, heads, False)
    ref = _sdpa_ref(qkv, n_seq, L, heads, 64, False)
    assert torch.isfinite(out.float()).all()
    torch.testing.assert_close(out.float(), ref, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("b,s,nq,pos0,heads,hd", [(3, 12, 12, 0, 8, 96), (3, 12, 1, 0, 8, 96), (3, 12, 1, 7, 8, 96), (2, 32, 32, 0, 8, 96),
                                                  (2, 4, 2, 1, 2, 64)])
def test_attn_policy(dev, b, s, nq, pos0, heads, hd):
    from otter_amd import ops
    dim = heads * hd
    cache = _randn(23, b, s, 3 * dim).to(dev)
    out = ops.attn_policy(cache, nq, pos0, heads)
    q, k, v = cache.view(b, s, 3, heads, hd).permute(2, 0, 3, 1, 4)
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + torch.full((s, s), float("-inf"), device=dev).triu(1)
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(b, s, dim)[:, pos0:pos0 + nq].reshape(b * nq, dim)
    assert out.dtype == torch.float32
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)             # fp32 throughout


@pytest.mark.parametrize("n_obs,cams,P,D,K,t_rep", [(2, 2, 256, 768, 15, 1), (6, 2, 256, 768, 15, 3), (3, 1, 64, 128, 15, 1),
                                                    (2, 2, 16, 128, 1, 2), (1, 2, 256, 768, 16, 1),
                                                    # patch counts that are not whole 16-row tiles: ViT-B/16 (196), ViT-B/32 (49), ...
                                                    (2, 2, 196, 512, 15, 1), (3, 1, 49, 512, 15, 1), (2, 1, 5, 128, 3, 1), (1, 2, 577, 768, 15, 1)])
@pytest.mark.parametrize("temperature", [0.07, 0.005, 150.0])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_text_aware_pool(dev, n_obs, cams, P, D, K, t_rep, temperature, dtype):
    from otter_amd import ops
    dt = DTYPES[dtype]
    patches = (_randn(24, n_obs * cams, P, D) * 1.7).to(dev).to(dt)
    text = (_randn(25, n_obs // t_rep, K, D) * 0.3).to(dev)
    pos = torch.stack([R.sincos_position_embedding(P, D) * (1 + 0.1 * c) for c in range(cams)]).to(dev)
    temp = torch.full((cams,), temperature, device=dev)
    pos_t = torch.stack([ops.pos_emb_pack(pos[c], dt) for c in range(cams)])
    out = ops.text_aware_pool(patches, text, pos_t, temp, n_obs, cams, t_rep)
    pf = patches.float().view(n_obs, cams, P, D)
    pf = pf / pf.norm(dim=-1, keepdim=True)
    tf = (text / text.norm(dim=-1, keepdim=True)).repeat_interleave(t_rep, 0)
    for c in range(cams):
        ref = R.text_aware_visual_extraction(pf[:, c].cpu(), tf.cpu(), torch.tensor(temperature), pos[c].cpu()).to(dev)
        # 16-bit rounding of text tokens, softmax weights and pos_emb: ~2^-8 (bf16) / 2^-11 (fp16) relative to
        # |pos_emb| <= 1.1; the logit error from the rounded text tokens is amplified by 1/temperature (200x at 0.005).
        # Bounds = twice the worst error measured over these cases (bf16: 2.5e-2 at 0.005, 1.7e-3 otherwise; fp16: 2.5e-3, 3.0e-4)
        tol = (4e-2 if temperature < 0.01 else 3.5e-3) if dtype == "bf16" else (5e-3 if temperature < 0.01 else 6e-4)
        torch.testing.assert_close(out[c], ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_text_aware_pool_16bit_edges(dev, dtype):
    """Text tokens in / result out in the patches' 16-bit type: against the fp32-edged launch of the same kernel.  16-bit text
    alone changes the result by what rounding the text rows costs; a 16-bit result is the fp32 result rounded once."""
    from otter_amd import ops
    dt = DTYPES[dtype]
    n_obs, cams, P, D, K = 5, 2, 256, 768, 15
    g = torch.Generator().manual_seed(17)
    patches = torch.randn(n_obs * cams, P + 1, D, generator=g).to(dev).to(dt)
    text = torch.randn(n_obs, K, D, generator=g).to(dev)
    pos = torch.stack([ops.pos_emb_pack(torch.randn(P, D, generator=g).to(dev), dt) for _ in range(cams)])
    temp = torch.tensor([0.07, 0.1], device=dev)
    base = ops.text_aware_pool(patches, text, pos, temp, n_obs, cams, 1, n_patches=P)
    out16 = ops.text_aware_pool(patches, text, pos, temp, n_obs, cams, 1, n_patches=P, out_dtype=dt)
    assert out16.dtype == dt and torch.equal(out16, base.to(dt))              # the same accumulators, rounded once
    t16 = text.to(dt)
    both = ops.text_aware_pool(patches, t16, pos, temp, n_obs, cams, 1, n_patches=P, out_dtype=dt)
    ref16 = ops.text_aware_pool(patches, t16.float(), pos, temp, n_obs, cams, 1, n_patches=P)      # fp32 edges fed the rounded text
    assert torch.equal(both, ref16.to(dt))
    tol = 3e-2 if dtype == "bf16" else 4e-3
    torch.testing.assert_close(both.float(), base, rtol=tol, atol=tol)
    with pytest.raises(TypeError):
        ops.text_aware_pool(patches, text.double(), pos, temp, n_obs, cams, 1, n_patches=P)


@pytest.mark.parametrize("case_name", ["vision_nomask", "text_mask", "small"])
def test_readout_pool(dev, case_name):
    import golden_inputs as GI
    from otter_amd import ops
    from otter_amd.params import pooling_specs, random_state_dict
    from otter_amd.pooling import PoolingWeights
    case = next(c for c in GI.POOL_CASES if c["name"] == case_name)
    sd = random_state_dict(pooling_specs("", case["D_in"], case["D_out"], case["layers"], case["readouts"]), seed=case["seed"])
    x, mask = GI.pool_inputs(case)
    ref = R.attention_pooling(x, sd, "", case["heads"], mask)
    pw = PoolingWeights({k: v.to(dev) for k, v in sd.items()}, "", case["heads"])
    out = torch.zeros(case["B"], case["D_out"] + 8, device=dev)
    pw.forward(x.to(dev).reshape(-1, case["D_in"]), None if mask is None else mask.to(dev), out[:, 4:4 + case["D_out"]], case["B"], case["K"])
    # fp32 MFMA k/v projections + fp32 readout kernel
    torch.testing.assert_close(out[:, 4:4 + case["D_out"]].cpu(), ref, rtol=1e-4, atol=2e-5)
    assert torch.all(out[:, :4] == 0) and torch.all(out[:, 4 + case["D_out"]:] == 0)


def test_readout_pool_multi_matches_single_launches(dev):
    """Three poolings of different widths, key counts and batch sizes in one launch: the same bits as one launch each
    (a workgroup does not know what else is in the grid), an empty problem skipped, a fifth problem refused."""
    import golden_inputs as GI
    from otter_amd import ops
    from otter_amd.params import pooling_specs, random_state_dict
    from otter_amd.pooling import PoolingWeights
    single, multi, problems = [], [], []
    keep = []
    for case in (c for name in ("vision_nomask", "text_mask", "small") for c in GI.POOL_CASES if c["name"] == name):
        sd = random_state_dict(pooling_specs("", case["D_in"], case["D_out"], case["layers"], case["readouts"]), seed=case["seed"])
        x, mask = GI.pool_inputs(case)
        pw = PoolingWeights({k: v.to(dev) for k, v in sd.items()}, "", case["heads"])
        keep.append(pw)
        xd, md = x.to(dev).reshape(-1, case["D_in"]), None if mask is None else mask.to(dev)
        single.append(pw.forward(xd, md, torch.zeros(case["B"], case["D_out"], device=dev), case["B"], case["K"]))
        multi.append(torch.zeros(case["B"], case["D_out"], device=dev))
        problems.append(pw.problem(xd, md, multi[-1], case["B"], case["K"]))
    assert len({(p[7], p[8], p[10]) for p in problems}) == len(problems) == 3          # different (n_obs, nk, d)
    ops.readout_pool_multi(problems)
    for a, b in zip(single, multi):
        assert torch.equal(a, b)
    empty = list(problems[1])
    empty[4], empty[6], empty[7] = empty[4][:0], empty[6][:0], 0
    if empty[5] is not None:
        empty[5] = empty[5][:0]
    for m in multi:
        m.zero_()
    ops.readout_pool_multi([problems[0], tuple(empty), problems[2]])
    assert torch.equal(multi[0], single[0]) and torch.equal(multi[2], single[2]) and torch.all(multi[1] == 0)
    with pytest.raises(ops.OtterHipError):
        ops.readout_pool_multi([problems[0]] * 5)


def test_small_ops(dev):
    from otter_amd import ops
    x = _randn(26, 33, 768).to(dev)
    torch.testing.assert_close(ops.l2norm_rows(x), x / x.norm(dim=-1, keepdim=True), rtol=1e-5, atol=1e-6)
    # fp32 linear chain (proprio MLP shapes)
    h = _randn(27, 19, 10).to(dev)
    w = _randn(28, 256, 10).to(dev)
    b = _randn(29, 256).to(dev)
    y = ops.linear_f32(h, w.t().contiguous(), b, ops.ACT_RELU)
    torch.testing.assert_close(y, torch.relu(h @ w.t() + b), rtol=1e-5, atol=1e-5)
    # row conversion with padding and repetition
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        c = ops.convert_rows(x[:5, :700].contiguous(), dt, cols_pad=704, rep=3)
        assert c.shape == (15, 704) and c.dtype == dt and torch.all(c[:, 700:] == 0)
        assert torch.equal(c[:, :700], x[:5, :700].to(dt).repeat_interleave(3, 0))
    pick = torch.tensor([0, 3, 1], dtype=torch.int32, device=dev)
    src = _randn(30, 12, 64).to(dev)
    sel = ops.convert_rows(src, pick=pick, group=4)
    assert torch.equal(sel, src.view(3, 4, 64)[torch.arange(3), pick.long().cpu()])
    sel = ops.convert_rows(src, pick=pick + 1, group=4, pick_bias=-1)
    assert torch.equal(sel, src.view(3, 4, 64)[torch.arange(3), pick.long().cpu()])
    dst = torch.zeros(10, 40, device=dev)
    ops.repeat_rows(src[:5, :32], dst[:, 4:36], 2)
    assert torch.equal(dst[:, 4:36], src[:5, :32].repeat_interleave(2, 0)) and torch.all(dst[:, :4] == 0)
    # positional embedding add over a window slice
    win = _randn(31, 3, 12, 64).to(dev)
    pe = _randn(32, 12, 64).to(dev)
    torch.testing.assert_close(ops.add_pe(win, pe, 5, 4), (win[:, 4:9] + pe[4:9]).reshape(-1, 64), rtol=0, atol=0)


def test_window_append_matches_reference_cache(dev):
    """otter.py:158-175: concatenate on the time axis, keep the last seq_length entries."""
    from otter_amd import ops
    B, S, dim = 3, 4, 64
    bufs = [torch.zeros(B, S, dim, device=dev), torch.zeros(B, S, dim, device=dev)]
    lens = [torch.tensor([0, 2, 4], dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)]
    hist = [[], [_randn(40, dim), _randn(41, dim)], [_randn(42 + i, dim) for i in range(4)]]
    for b in range(B):
        for j, v in enumerate(hist[b]):
            bufs[0][b, j] = v.to(dev)
    cur = 0
    for step in range(6):
        fresh = _randn(100 + step, B, dim)
        ops.window_append(bufs[cur], fresh.to(dev), bufs[cur ^ 1], lens[cur], lens[cur ^ 1])
        cur ^= 1
        for b in range(B):
            hist[b] = (hist[b] + [fresh[b]])[-S:]
            n = int(lens[cur][b])
            assert n == len(hist[b])
            assert torch.equal(bufs[cur][b, :n].cpu(), torch.stack(hist[b]))
            assert torch.all(bufs[cur][b, n:] == 0)


@pytest.mark.parametrize("h,w,size", [(180, 320, 224), (720, 1280, 224), (300, 200, 224), (224, 224, 224)])
def test_crop_resize_device_is_pil_bicubic_bit_for_bit(dev, h, w, size):
    from PIL import Image
    from otter_amd.vision_tf import crop2square, crop_resize_device, pil_to_uint8_chw
    rng = np.random.default_rng(h + 3 * w)
    frames = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    frames[1, : h // 2] = 255
    got = crop_resize_device(torch.from_numpy(frames).to(dev), size).cpu().numpy()
    for i in range(3):
        want = pil_to_uint8_chw(crop2square(Image.fromarray(frames[i])).resize((size, size), Image.Resampling.BICUBIC)).numpy()
        assert np.array_equal(got[i], want)
    with pytest.raises(ValueError):
        crop_resize_device(torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device=dev), size)


def test_empty_and_limit_cases(dev):
    """Zero-sized problems are no-ops; sizes past a kernel's limit are refused with an error, never run."""
    from otter_amd import ops
    f16 = torch.float16
    w = ops.pack_weight(torch.randn(64, 64), f16, dev)
    assert ops.gemm(torch.zeros(0, 64, dtype=f16, device=dev), w, 64).shape == (0, 64)
    ops.layernorm(torch.zeros(0, 64, device=dev), torch.ones(64, device=dev), torch.zeros(64, device=dev), out_dtype=f16)
    assert ops.attn_d64(torch.zeros(0, 3 * 64, dtype=f16, device=dev), 0, 7, 1, False).shape[0] == 0
    with pytest.raises(ops.OtterHipError):
        ops.attn_d64(torch.zeros(593, 3 * 64, dtype=f16, device=dev), 1, 593, 1, False)          # more than 592 keys
    kv = torch.zeros(1, 65, 3 * 64, device=dev)
    with pytest.raises(ops.OtterHipError):
        ops.attn_policy(kv, 1, 0, 1)                                                              # window > 64
    with pytest.raises(ops.OtterHipError):
        ops.small_attn_fwd(torch.zeros(1, 33, 8, device=dev), torch.zeros(1, 4, 8, device=dev), torch.zeros(1, 4, 8, device=dev), 1, False)
    pos = ops.pos_emb_pack(torch.zeros(16, 128, device=dev), f16)[None].contiguous()
    out = ops.text_aware_pool(torch.zeros(0, 16, 128, dtype=f16, device=dev), torch.zeros(0, 3, 128, device=dev), pos,
                              torch.ones(1, device=dev), 0, 1, 1)
    assert out.shape == (1, 0, 3, 128)
    with pytest.raises(ops.OtterHipError):
        ops.text_aware_pool(torch.zeros(1, 16, 128, dtype=f16, device=dev), torch.zeros(1, 17, 128, device=dev), pos,
                            torch.ones(1, device=dev), 1, 1, 1)                                    # more than 16 text tokens


def test_text_aware_pool_convexity_at_bench_size(dev):
    """Size-independent property at the benchmark's launch size (128 units, 256 patches, 768 wide): with a constant
    positional table c the output is c plus a convex combination of unit vectors, so |out - c| <= 1, and a unit whose
    patches are all equal returns exactly that direction whatever the text is."""
    from otter_amd import ops
    n_obs, cams, P, D, K = 64, 2, 256, 768, 15
    g = torch.Generator().manual_seed(5)
    patches = torch.randn(n_obs * cams, P, D, generator=g)
    patches[7] = patches[7, :1].expand(P, D)                      # unit 7: identical patches
    text = torch.randn(n_obs, K, D, generator=g)
    c = torch.randn(D, generator=g) * 0.3
    pos = torch.stack([ops.pos_emb_pack(c.expand(P, D).contiguous().to(dev), torch.float16)] * cams)
    out = ops.text_aware_pool(patches.to(dev).half(), text.to(dev), pos, torch.tensor([0.07, 0.02], device=dev), n_obs, cams, 1)
    centred = out.cpu() - c.half().float()
    assert float(centred.norm(dim=-1).max()) <= 1.0 + 5e-3
    m, cam = 7 // cams, 7 % cams
    direction = patches[7, 0].half().float()
    direction = direction / direction.norm()
    torch.testing.assert_close(centred[cam, m], direction.expand(K, D), rtol=0, atol=3e-3)


def test_torch_custom_ops_match_the_wrappers(dev):
    """Every ``torch.ops.otter_hip.*`` operator runs the same kernel as its :mod:`otter_amd.ops` wrapper (bit-identical
    results), the mutating forms included."""
    import otter_amd.torch_ops as T
    from otter_amd import ops
    from otter_amd.params import pooling_specs, random_state_dict
    from otter_amd.pooling import PoolingWeights
    ns = torch.ops.otter_hip
    covered = set()

    def op(name):
        covered.add(name)
        return getattr(ns, name)

    dt = torch.float16
    a = _randn(91, 300, 128).to(dev).to(dt)
    wp = ops.pack_weight(_randn(92, 248, 128).to(dev) / 11.0, dt)
    bias = _randn(93, 248).to(dev)
    res = _randn(94, 300, 248).to(dev)
    assert torch.equal(op("gemm")(a, wp, 248, bias, ACTS["quickgelu"], False, None), ops.gemm(a, wp, 248, bias, act=ACTS["quickgelu"]))
    assert torch.equal(ns.gemm(a, wp, 248, bias, 0, False, res), ops.gemm(a, wp, 248, bias, residual=res))
    wf, bf, cs = ops.fold_layernorm(_randn(92, 248, 128).to(dev) / 11.0, bias, _randn(80, 128).to(dev), _randn(81, 128).to(dev), dt, dev)
    st = op("row_stats")(a, 1e-5)
    assert torch.equal(st, ops.row_stats(a))
    assert torch.equal(ns.gemm(a, wf, 248, bf, 0, False, None, st, cs), ops.gemm(a, wf, 248, bf, ln_stats=st, ln_colsum=cs))
    big = torch.zeros(300, 264, dtype=dt, device=dev)
    op("gemm_into")(big[:, 8:256], a, wp, 248, bias, 0, None)                    # a strided view of a wider buffer
    assert torch.equal(big[:, 8:256], ops.gemm(a, wp, 248, bias)) and torch.all(big[:, :8] == 0) and torch.all(big[:, 256:] == 0)
    x_res = res.clone()
    op("gemm_residual_")(x_res, a, wp, 248, bias)
    assert torch.equal(x_res, ops.gemm(a, wp, 248, bias, residual=res))
    x16 = res.to(dt)
    want16 = ops.gemm(a, wp, 248, bias, residual=x16)
    st16 = op("gemm_residual_stats_")(x16, a, wp, 248, bias, 1e-5)
    assert torch.equal(x16, want16) and torch.equal(st16, ops.row_stats(want16))
    x = _randn(95, 37, 768).to(dev)
    g, b = _randn(96, 768).to(dev), _randn(97, 768).to(dev)
    assert torch.equal(op("layernorm")(x, g, b, 1e-5, 1), ops.layernorm(x, g, b, out_dtype=dt))
    imgs = torch.randint(0, 256, (3, 3, 28, 28), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).to(dev)
    mean, std = [0.48, 0.45, 0.40], [0.26, 0.26, 0.27]
    cols = op("im2col_patches")(imgs, 14, 640, mean, std, 1)
    assert torch.equal(cols, ops.im2col_patches(imgs, 14, 640, mean, std, dt))
    tok, cls, vpos = _randn(60, 3 * 4, 64).to(dev), _randn(61, 64).to(dev), _randn(62, 5, 64).to(dev)
    g64, b64 = _randn(63, 64).to(dev), _randn(64, 64).to(dev)
    assert torch.equal(op("vit_embed_lnpre")(tok, cls, vpos, g64, b64, 3, 4, 1e-5, 1), ops.vit_embed_lnpre(tok, cls, vpos, g64, b64, 3, 4, out_dtype=dt))
    ids = torch.randint(0, 100, (2, 77), dtype=torch.int32, generator=torch.Generator().manual_seed(4)).to(dev)
    table, tpos = _randn(65, 100, 64).to(dev), _randn(66, 77, 64).to(dev)
    assert torch.equal(op("text_embed")(ids, 15, table, tpos, 2), ops.text_embed(ids, 15, table, tpos, out_dtype=torch.float32))
    qkv = _randn(98, 2 * 50, 3 * 128).to(dev).to(dt)
    assert torch.equal(op("attn_fwd_d64")(qkv, 2, 50, 2, False), ops.attn_d64(qkv, 2, 50, 2, False))
    cache = _randn(99, 3, 12, 3 * 96).to(dev)
    assert torch.equal(op("attn_policy")(cache, 1, 5, 2), ops.attn_policy(cache, 1, 5, 2))
    # ragged cached step: episodes 2 and 0 at positions 5 and 3 -> the same numbers as the lock-step kernel at those positions
    lens = torch.tensor([4, 9, 6], dtype=torch.int32, device=dev)
    idx = torch.tensor([2, 0], dtype=torch.int32, device=dev)
    fresh = torch.stack([cache[2, 5], cache[0, 3]]).contiguous()
    c1, c2 = cache.clone(), cache.clone()
    c1[2, 5], c1[0, 3], c2[2, 5], c2[0, 3] = 0, 0, 0, 0
    got = op("attn_policy_rows")(fresh, c1, idx, lens, 2)
    assert torch.equal(got, ops.attn_policy_rows(fresh, c2, idx, lens, 2)) and torch.equal(c1, cache) and torch.equal(c2, cache)
    assert torch.equal(got[0], ops.attn_policy(cache, 1, 5, 2)[2]) and torch.equal(got[1], ops.attn_policy(cache, 1, 3, 2)[0])
    pos = _randn(100, 49, 512).to(dev)
    packed = op("pos_emb_pack")(pos, 1)
    assert torch.equal(packed, ops.pos_emb_pack(pos, dt))
    patches = _randn(101, 4, 50, 512).to(dev).to(dt)
    text = _randn(102, 2, 15, 512).to(dev)
    pk = torch.stack([packed, packed])
    temp = torch.full((2,), 0.07, device=dev)
    assert torch.equal(op("text_aware_pool")(patches, text, pk, temp, 2, 2, 1, 49), ops.text_aware_pool(patches, text, pk, temp, 2, 2, 1, n_patches=49))
    assert torch.equal(op("l2norm_rows")(x), ops.l2norm_rows(x))
    # readout: one pooling through the operator against the ctypes-level call
    sd = random_state_dict(pooling_specs("", 128, 64, 2, 4), seed=5)
    pw = PoolingWeights({k: v.to(dev) for k, v in sd.items()}, "", 4)
    xin = _randn(103, 6 * 15, 128).to(dev)
    o1, o2 = torch.zeros(6, 64, device=dev), torch.zeros(6, 64, device=dev)
    pr = pw.problem(xin, None, o1, 6, 15)
    op("readout_pool_multi")(list(pr[0]), [len(pr[0]) // 12], [pr[1]], [pr[2]], [pr[3]], [pr[4]], [None], [o1], [6], [15], [pr[9]], [pr[10]], [pr[11]], [pr[12]])
    ops.readout_pool(pr[0], pr[1], pr[2], pr[3], pr[4], None, o2, 6, 15, pr[9], pr[10], pr[11], pr[12])
    assert torch.equal(o1, o2) and float(o1.abs().sum()) > 0
    xp, wt, bl = _randn(104, 5, 10).to(dev), _randn(105, 10, 256).to(dev), _randn(106, 256).to(dev)
    assert torch.equal(op("linear_f32")(xp, wt, bl, ACTS["relu"]), ops.linear_f32(xp, wt, bl, ACTS["relu"]))
    wide = torch.zeros(5, 300, device=dev)
    op("linear_f32_into")(wide[:, 20:276], xp, wt, bl, 0)
    assert torch.equal(wide[:, 20:276], ops.linear_f32(xp, wt, bl)) and torch.all(wide[:, :20] == 0)
    src = _randn(107, 4, 128).to(dev)
    d1, d2 = torch.zeros(12, 128, device=dev), torch.zeros(12, 128, device=dev)
    op("repeat_rows_into")(d1, src, 3)
    assert torch.equal(d1, ops.repeat_rows(src, d2, 3)) and torch.equal(d1, src.repeat_interleave(3, 0))
    win = _randn(108, 3, 12, 768).to(dev)
    rows = win.view(36, 768)
    assert torch.equal(op("convert_rows")(rows, 2, 0, 1, lens, 12, -1, -1), ops.convert_rows(rows, pick=lens, group=12, pick_bias=-1))
    assert torch.equal(ns.convert_rows(src, 1, 192, 2, None, 1, -1, 0), ops.convert_rows(src, dt, 192, 2))
    pe = _randn(109, 12, 768).to(dev)
    assert torch.equal(op("add_pe")(win, pe, 1, 5), ops.add_pe(win, pe, 1, 5))
    assert torch.equal(op("add_pe_rows")(win, pe, idx, lens), ops.add_pe_rows(win, pe, idx, lens))
    assert torch.equal(ns.add_pe_rows(win, pe, idx, lens), torch.stack([win[2, 5] + pe[5], win[0, 3] + pe[3]]))
    newrow = _randn(110, 3, 768).to(dev)
    w1, w2, l1, l2 = torch.zeros_like(win), torch.zeros_like(win), torch.zeros_like(lens), torch.zeros_like(lens)
    op("window_append")(win, newrow, w1, lens, l1)
    ops.window_append(win, newrow, w2, lens, l2)
    assert torch.equal(w1, w2) and torch.equal(l1, l2) and l1.tolist() == [5, 10, 7]
    delta = (_randn(111, 4, 12, 10) * 0.2).to(dev)
    delta[..., 3:9] += torch.tensor([1.0, 0, 0, 0, 1, 0], device=dev)
    pose = _randn(112, 4, 10).double().to(dev)
    r1, r2 = torch.zeros(4, 3, 3, 10, dtype=torch.float64, device=dev), torch.zeros(4, 3, 3, 10, dtype=torch.float64, device=dev)
    n1, n2 = torch.zeros(4, dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
    got = op("action_step")(delta, pose, r1, n1, 3, False, 0.05, True)
    want = ops.action_step(delta, pose, r2, n2, 3, False, 0.05, True)
    assert all(torch.equal(u, v) for u, v in zip(got, want)) and torch.equal(r1, r2) and n1.tolist() == [1, 1, 1, 1]
    assert covered == set(T.OP_NAMES), set(T.OP_NAMES) - covered


def test_encoder_traces_into_one_graph_of_custom_operators(dev):
    """``OtterDevice.encode`` under ``torch.compile(fullgraph=True)``: Dynamo traces the model code with fake tensors (the
    operators' shape-inference implementations) into ONE graph with no break, every kernel a ``torch.ops.otter_hip`` node;
    running the traced graph gives the eager launches' bits."""
    import golden_inputs as GI
    from otter_amd.otter import OTTER
    from otter_amd.params import otter_specs, random_state_dict
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model = OTTER(mc, sc)
    model.load_state_dict(random_state_dict(otter_specs(mc, sc), seed=case["seed"]), strict=True)
    d = model.to(dev).eval().device_model()
    rng = np.random.default_rng(5)
    n = 3
    images = torch.from_numpy(rng.integers(0, 256, (n * 2, 3, sc.image_size, sc.image_size), dtype=np.uint8)).to(dev)
    text = d.clip.text_tokens_cached(GI._tokens(rng, n, (4, 6)))
    proprio = torch.from_numpy(rng.standard_normal((n, 10)).astype(np.float32)).to(dev)
    eager = d.encode(images, text, proprio, 1)
    graphs = []

    def backend(gm, example_inputs):            # no code generation: keep the graph, run it node by node
        graphs.append(gm)
        return gm.forward

    torch._dynamo.reset()
    compiled = torch.compile(d.encode, backend=backend, fullgraph=True)
    got = compiled(images, text, proprio, 1)
    assert len(graphs) == 1
    targets = [str(node.target) for node in graphs[0].graph.nodes if node.op == "call_function"]
    kernels_in_graph = [t for t in targets if t.startswith("otter_hip.")]
    names = {t.split(".")[1] for t in kernels_in_graph}
    print(f"encode(): one FX graph, {len(kernels_in_graph)} otter_hip kernel nodes of {len(targets)} calls; operators {sorted(names)}")
    assert {"im2col_patches", "gemm", "gemm_residual_stats_", "row_stats", "attn_fwd_d64", "vit_embed_lnpre", "layernorm", "text_aware_pool",
            "readout_pool_multi", "l2norm_rows", "linear_f32", "linear_f32_into"} <= names
    assert len(kernels_in_graph) > 20
    assert torch.equal(got, eager)
    torch._dynamo.reset()


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("rows,width,pad", [(5, 768, 0), (1031, 1024, 64), (64, 8, 8), (3, 520, 0), (130, 2048, 0)])
def test_row_stats_shapes(dev, rows, width, pad, dtype):
    """Row statistics over ragged row counts, widths on both sides of the 16-bit kernel's limits, and strided rows."""
    from otter_amd import ops
    dt = DTYPES[dtype]
    big = torch.zeros(rows, width + pad, dtype=dt, device=dev)
    x = big[:, :width]
    x.copy_((_randn(110, rows, width) * 1.7 - 0.4).to(dt))
    st = ops.row_stats(x)
    xf = x.float()
    torch.testing.assert_close(st[:, 0], xf.mean(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(st[:, 1], (xf.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=2e-5, atol=1e-5)
