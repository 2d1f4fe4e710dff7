"""GPU: every libttvdm kernel (called through the C ABI) against a plain PyTorch fp32 reference of the
same op evaluated on the SAME inputs, in all three storage modes.  Tolerances:
  fp16   rtol = atol = 1e-3   (north_star's rtol; outputs are rounded to fp16, 2^-11, on store; accumulation is fp32)
  bf16   rtol = atol = 1.6e-2 (2^-8 on store; `scale` widens atol where the outputs are O(4))
  fp32   rtol = atol = 2e-5   (TT_F32, the reference-precision mode: exact-fp32 MFMA, only the summation order differs)"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES16 = [torch.float16, torch.bfloat16]
DTYPES = DTYPES16 + [torch.float32]
TOL = {torch.float16: dict(rtol=1e-3, atol=1e-3), torch.bfloat16: dict(rtol=1.6e-2, atol=1.6e-2),
       torch.float32: dict(rtol=2e-5, atol=2e-5)}


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from this_and_that_vdm_amd import ops as o
    return o


def rnd(*shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def close(got, ref, dtype, scale=1.0):
    tol = TOL[dtype]
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=tol["rtol"],
                               atol=tol["atol"] * (scale if dtype == torch.bfloat16 else 1.0))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k", [(300, 72, 40), (1000, 320, 640), (129, 132, 64), (4096, 256, 1280)])
def test_gemm_linear_full_epilogue(ops, dtype, m, n, k):
    a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
    bias = rnd(n, dtype=torch.float32, seed=3)
    rows_per = 50
    rowvec = rnd((m + rows_per - 1) // rows_per, n, dtype=torch.float32, seed=4)
    res, bl = rnd(m, n, dtype=dtype, seed=5), rnd(m, n, dtype=dtype, seed=6)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), acc_scale=0.75, rowvec=rowvec.cuda(), rowvec_rows=rows_per,
                   residual=res.cuda(), blend=bl.cuda(), alpha=0.3)
    ref = (a.float() @ w.float().T + bias) * 0.75 + rowvec.repeat_interleave(rows_per, 0)[:m] + res.float()
    ref = 0.3 * bl.float() + 0.7 * ref
    close(out, ref, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k", [(1000, 320, 320), (6272, 640, 640), (1568, 1280, 1280), (300, 128, 5120)])
def test_gemm_in_place_on_the_residual(ops, dtype, m, n, k):
    """out may alias the residual (the zero-context shortcut updates the live rows of the hidden states in place): every lane
    reads the residual element it overwrites, in the tile kernels and in the split-K reduction alike -- bit-identical to the
    out-of-place call, and rows outside the slice stay untouched."""
    a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
    bias = rnd(n, dtype=torch.float32, seed=3)
    x = rnd(m + 64, n, dtype=dtype, seed=5).cuda()
    ref = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=x[32:32 + m])
    keep = x.clone()
    ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=x[32:32 + m], out=x[32:32 + m])
    assert torch.equal(x[32:32 + m], ref)
    assert torch.equal(x[:32], keep[:32]) and torch.equal(x[32 + m:], keep[32 + m:])


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("m,n,k", [(1300, 2560, 192), (2900, 1408, 64), (9000, 1280, 128)])
def test_gemm_grouped_tile_order(ops, dtype, m, n, k):
    """More than 8 column tiles: the launch order walks groups of tile rows column by column (gemm_kernel.h).  Shapes whose
    last group is ragged (11, 23 and 36 / 71 tile rows against groups of 8 / 4) must still cover every tile exactly once."""
    a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
    out = torch.full((m + 8, n), 3.0, dtype=dtype, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), out=out[:m])
    close(out[:m], a.float() @ w.float().T, dtype, scale=2.0)
    assert (out[m:] == 3.0).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_is_transpose_safe(ops, dtype):
    """A = I with an asymmetric W: catches swapped row/col fragment maps (cdna guide rule 16)."""
    n = 96
    a = torch.eye(64, dtype=dtype)
    w = (torch.arange(n * 64).reshape(n, 64) % 251 - 125).to(dtype) / 64
    out = ops.gemm(a.cuda(), w.cuda())
    torch.testing.assert_close(out.float().cpu(), w.float().T, rtol=0, atol=0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_two_sources_and_f32_out(ops, dtype):
    m, n, k0, k1 = 500, 64, 64, 40
    a0, a1 = rnd(m, k0, dtype=dtype, seed=1), rnd(m, k1, dtype=dtype, seed=2)
    w = rnd(n, k0 + k1, dtype=dtype, seed=3, scale=0.1)
    out = ops.gemm(a0.cuda(), w.cuda(), a1=a1.cuda(), out_f32=True)
    assert out.dtype == torch.float32
    ref = torch.cat([a0, a1], 1).float() @ w.float().T
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_geglu(ops, dtype):
    m, c = 260, 64
    a = rnd(m, c, dtype=dtype, seed=1)
    w = rnd(8 * c, c, dtype=dtype, seed=2, scale=c ** -0.5)        # nn.Linear(c, 8c): rows [value 4c | gate 4c]
    b = rnd(8 * c, dtype=torch.float32, seed=3)
    from this_and_that_vdm_amd.packing import pack_geglu
    wp, bp = pack_geglu(w, b)
    out = ops.gemm(a.cuda(), wp.cuda(), bias=bp.cuda(), geglu=True)
    h = a.float() @ w.float().T + b
    ref = h[:, :4 * c] * F.gelu(h[:, 4 * c:])
    assert out.shape == (m, 4 * c)
    close(out, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("stride,upsample", [(1, 0), (2, 0), (1, 1)])
def test_gemm_conv3x3(ops, dtype, stride, upsample):
    nimg, cin0, cin1, cout, h, w = 3, 32, 24, 40, 10, 14
    x0, x1 = rnd(nimg, cin0, h, w, dtype=dtype, seed=1), rnd(nimg, cin1, h, w, dtype=dtype, seed=2)
    wt = rnd(cout, cin0 + cin1, 3, 3, dtype=dtype, seed=3, scale=0.06)
    bias = rnd(cout, dtype=torch.float32, seed=4)
    from this_and_that_vdm_amd.packing import pack_conv3x3
    wp = pack_conv3x3(wt)
    xin = torch.cat([x0, x1], 1).float()
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, wt.float(), bias, stride=stride, padding=1)
    ho, wo = ref.shape[-2:]
    t0 = x0.permute(0, 2, 3, 1).reshape(-1, cin0).contiguous().cuda()
    t1 = x1.permute(0, 2, 3, 1).reshape(-1, cin1).contiguous().cuda()
    out = ops.gemm(t0, wp.cuda(), a1=t1, mode=1, conv=(nimg, h, w, ho, wo, stride, upsample), bias=bias.cuda())
    close(out, ref.permute(0, 2, 3, 1).reshape(-1, cout), dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("h,w,c0,c1,cout", [(32, 56, 64, 0, 160), (16, 28, 128, 64, 128), (8, 16, 64, 64, 320), (8, 14, 192, 0, 64),
                                            (64, 112, 64, 0, 96)])
def test_conv3x3_fused_groupnorm(ops, dtype, h, w, c0, c1, cout):
    """tt_conv3x3: conv3x3(silu(group_norm(cat(x0, x1)))) + bias + FiLM row + residual, GroupNorm applied while the input
    patch (with halo) is staged in LDS; zero padding AFTER the activation; every tile shape (16x8, 8x14, 8x16)."""
    nimg, frames = 4, 2
    assert ops.conv3x3_supported(h, w, c0, c1, cout, dtype)
    x0 = rnd(nimg, c0, h, w, dtype=dtype, seed=1, scale=2.0) + 0.5
    x1 = rnd(nimg, c1, h, w, dtype=dtype, seed=2) if c1 else None
    c = c0 + c1
    wt = rnd(cout, c, 3, 3, dtype=dtype, seed=3, scale=(9 * c) ** -0.5)
    bias = rnd(cout, dtype=torch.float32, seed=4)
    gamma, beta = rnd(c, dtype=torch.float32, seed=5, scale=0.2) + 1, rnd(c, dtype=torch.float32, seed=6, scale=0.3)
    film = rnd(nimg // frames, cout, dtype=torch.float32, seed=7)
    res = rnd(nimg * h * w, cout, dtype=dtype, seed=8)
    from this_and_that_vdm_amd.packing import pack_conv3x3
    xin = torch.cat([x0, x1], 1).float() if c1 else x0.float()
    act = F.silu(F.group_norm(xin, 32, gamma, beta, eps=1e-5))
    ref = F.conv2d(act, wt.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    ref = ref + film.repeat_interleave(frames * h * w, 0) + res.float()
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().cuda()
    t0, t1 = tok(x0), (tok(x1) if c1 else None)
    st = ops.groupnorm_stats(t0, t1, nimg, h * w, 1, gamma.cuda(), beta.cuda(), 1e-5)
    out = torch.full((nimg * h * w + 8, cout), 7.0, dtype=dtype, device="cuda")       # canary rows behind the output
    ops.conv3x3(t0, t1, pack_conv3x3(wt).cuda(), nimg, h, w, gn=st, silu=True, bias=bias.cuda(), rowvec=film.cuda(),
                rowvec_rows=frames * h * w, residual=res.cuda(), out=out[:nimg * h * w])
    close(out[:nimg * h * w], ref, dtype, scale=2.0)
    assert (out[nimg * h * w:] == 7.0).all()
    # ... and without GroupNorm / epilogue terms (plain conv): equals the implicit-GEMM kernel's result up to rounding
    plain = ops.conv3x3(t0, t1, pack_conv3x3(wt).cuda(), nimg, h, w)
    close(plain, F.conv2d(xin, wt.float(), None, padding=1).permute(0, 2, 3, 1).reshape(-1, cout), dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_temporal_conv(ops, dtype):
    b, f, hw, c = 2, 5, 12, 64
    x = rnd(b, c, f, hw, 1, dtype=dtype, seed=1)
    wt = rnd(c, c, 3, 1, 1, dtype=dtype, seed=2, scale=0.07)
    bias = rnd(c, dtype=torch.float32, seed=3)
    ref = F.conv3d(x.float(), wt.float(), bias, padding=(1, 0, 0))          # [b,c,f,hw,1]
    from this_and_that_vdm_amd.packing import pack_tconv3
    tok = x[..., 0].permute(0, 2, 3, 1).reshape(b * f * hw, c).contiguous().cuda()
    out = ops.gemm(tok, pack_tconv3(wt).cuda(), mode=2, tconv=(f, hw), bias=bias.cuda())
    close(out, ref[..., 0].permute(0, 2, 3, 1).reshape(b * f * hw, c), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_transposed_padded_output(ops, dtype):
    """V^T projection: out[c, seq*hwp + p] with sequences padded 28 -> 32 columns."""
    nseq, hw, hwp, c, k = 3, 28, 32, 64, 48
    x, w = rnd(nseq * hw, k, dtype=dtype, seed=1), rnd(c, k, dtype=dtype, seed=2, scale=0.1)
    out = torch.zeros(c, nseq * hwp, dtype=dtype, device="cuda")
    ops.gemm(w.cuda(), x.cuda(), out=out, out_col_pad=(hw, hwp))
    ref = (w.float() @ x.float().T).reshape(c, nseq, hw)
    got = out.float().cpu().reshape(c, nseq, hwp)
    close(got[:, :, :hw], ref, dtype)
    assert float(got[:, :, hw:].abs().max()) == 0.0


def _sdpa(q, k, v, heads):
    n, lq, c = q.shape
    d = c // heads
    qh, kh, vh = [t.float().view(t.shape[0], -1, heads, d).transpose(1, 2) for t in (q, k, v)]
    return F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(n, lq, c)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("l,heads,d", [(100, 2, 64), (256, 1, 64), (28, 2, 64), (200, 1, 128)])
def test_attention_self(ops, dtype, l, heads, d):
    nseq, c = 3, heads * d
    q, k, v = (rnd(nseq, l, c, dtype=dtype, seed=s) for s in (1, 2, 3))
    lp = (l + 7) // 8 * 8
    vt = torch.zeros(c, nseq * lp, dtype=dtype)
    vt.view(c, nseq, lp)[:, :, :l] = v.permute(2, 0, 1)
    out = torch.empty(nseq * l, c, dtype=dtype, device="cuda")
    ops.attention(q.reshape(-1, c).cuda(), k.reshape(-1, c).cuda(), vt.cuda(), out, nseq=nseq, lq=l, heads=heads, head_dim=d,
                  mask=0, lk=l, k_seq_stride=l, v_seq_stride=lp)
    close(out.view(nseq, l, c), _sdpa(q, k, v, heads), dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("n,k,kind", [(960, 320, "normal"), (2560, 1280, "normal"), (640, 640, "wide"), (37, 200, "wide"), (64, 8, "normal"),
                                      (16, 5120, "wide"), (128, 320, "zeros"), (8, 1000, "tiny")])
def test_zero_sum_round_on_the_device_matches_the_host_version(ops, dtype, n, k, kind):
    """tt_zero_sum_round (one block per row) against packing.zero_sum_round's tensor passes: the packed weights must be identical value for value (= bit for bit except the sign of a zero) --
    wide binade ranges, rows with zeros, all-zero rows, magnitudes near the storage type's subnormals -- and every row must sum to (almost) zero."""
    from this_and_that_vdm_amd import packing
    g = torch.Generator().manual_seed(n * 7919 + k)
    w = torch.randn(n, k, generator=g) * k ** -0.5
    if kind == "wide":
        w = w * torch.exp2(torch.randint(-14, 3, (n, k), generator=g).float())
    if kind == "zeros":
        w[torch.rand(n, k, generator=g) < 0.3] = 0.0
        w[3] = 0.0
    if kind == "tiny":
        w = w * (1e-37 if dtype == torch.bfloat16 else 3e-6)
    w = (w - w.mean(1, keepdim=True)).float()
    keep = packing.DEVICE_ROUNDING
    try:
        packing.DEVICE_ROUNDING = True
        dev = packing.zero_sum_round(w.cuda(), dtype)
        packing.DEVICE_ROUNDING = False
        host = packing.zero_sum_round(w.cuda(), dtype)
        cpu = packing.zero_sum_round(w, dtype)
    finally:
        packing.DEVICE_ROUNDING = keep
    assert dev.dtype == dtype and dev.is_cuda
    # (value equality: for non-zero 16-bit values that is bit equality; a weight that underflows to -0 keeps its sign on the device while the
    # host's `q - 0 * (-u)` turns it into +0 -- the same number)
    assert torch.equal(dev.cpu(), cpu), "device kernel differs from the host algorithm"
    assert torch.equal(host.cpu(), cpu)
    if k >= 64:                                               # (a row of 8 elements has too few candidates to cancel every residual)
        plain = w.to(dtype).double().sum(1).abs().max()
        assert float(dev.double().sum(1).abs().max()) <= max(float(plain) * 1e-2, 1e-30)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("l,heads", [(200, 2), (1792, 5), (448, 3), (28, 4), (65, 1)])
def test_attention_self_with_v_as_rows(ops, dtype, l, heads):
    """TtAttnArgs.v_rows (ABI 10): V read as it leaves a fused Q | K | V projection -- rows = keys, a column slice of a [M, 3C] tensor --
    and transposed on the way out of LDS (ds_read_b64_tr_b16).  Same MFMA operands as the V^T route: the outputs must agree BIT FOR BIT,
    ragged last key tiles included, and match fp32 SDPA."""
    d, nseq = 64, 3
    c = heads * d
    q, k, v = (rnd(nseq, l, c, dtype=dtype, seed=s) for s in (1, 2, 3))
    qkv = torch.cat([q, k, v], dim=2).reshape(nseq * l, 3 * c).cuda()
    lp = (l + 7) // 8 * 8
    vt = torch.zeros(c, nseq * lp, dtype=dtype)
    vt.view(c, nseq, lp)[:, :, :l] = v.permute(2, 0, 1)
    kw = dict(nseq=nseq, lq=l, heads=heads, head_dim=d, mask=0, lk=l, k_seq_stride=l)
    out_t = torch.empty(nseq * l, c, dtype=dtype, device="cuda")
    ops.attention(qkv[:, :c], qkv[:, c:2 * c], vt.cuda(), out_t, v_seq_stride=lp, **kw)
    out_r = torch.empty(nseq * l, c, dtype=dtype, device="cuda")
    ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], out_r, v_seq_stride=l, v_rows=True, **kw)
    close(out_r.view(nseq, l, c), _sdpa(q, k, v, heads), dtype)
    assert torch.equal(out_r, out_t), "row-major V must give the V^T route's output bit for bit"


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("l,heads,d", [(200, 2, 64), (1792, 1, 64), (448, 2, 64), (100, 1, 128)])
def test_attention_fp8(ops, dtype, l, heads, d):
    """BASELINE config 5: spatial self-attention on OCP e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4, unit scales), 16-bit output.
    Checked against fp32 SDPA on the SAME e4m3 values (what the kernel adds on top: P rounded to e4m3, 3 mantissa bits,
    inside fp32 sums) at rtol = atol = 3e-2, and -- printed -- against SDPA on the unquantised 16-bit values."""
    nseq, c = 2, heads * d
    q, k, v = (rnd(nseq, l, c, dtype=dtype, seed=s) for s in (1, 2, 3))
    q8, k8, v8 = (t.to(ops.FP8) for t in (q, k, v))
    lp = (l + 15) // 16 * 16
    vt = torch.zeros(c, nseq * lp, dtype=torch.uint8).view(ops.FP8)
    vt.view(torch.uint8).view(c, nseq, lp)[:, :, :l] = v8.view(torch.uint8).permute(2, 0, 1)
    out = torch.empty(nseq * l, c, dtype=dtype, device="cuda")
    ops.attention(q8.reshape(-1, c).cuda(), k8.reshape(-1, c).cuda(), vt.cuda(), out, nseq=nseq, lq=l, heads=heads, head_dim=d,
                  mask=0, lk=l, k_seq_stride=l, v_seq_stride=lp)
    ref8 = _sdpa(q8.float(), k8.float(), v8.float(), heads)
    torch.testing.assert_close(out.view(nseq, l, c).float().cpu(), ref8, rtol=3e-2, atol=3e-2)
    ref = _sdpa(q, k, v, heads)
    err = (out.view(nseq, l, c).float().cpu() - ref)
    print(f"fp8 attention L={l} d={d} {dtype}: vs SDPA on e4m3 inputs max {float((out.view(nseq, l, c).float().cpu() - ref8).abs().max()):.4f}; "
          f"vs SDPA on 16-bit inputs max {float(err.abs().max()):.4f}, rel-L2 {float(err.norm() / ref.norm()):.4f}")
    assert float(err.norm() / ref.norm()) <= 0.1


@pytest.mark.parametrize("dtype", DTYPES16)
def test_gemm_fp8_output(ops, dtype):
    """tt_gemm out_fp8: e4m3 bytes of the Q | K projection (LayerNorm folded, bias) and of the transposed, padded V^T."""
    from this_and_that_vdm_amd.packing import fold_layernorm, zero_sum_round
    m, c, n = 500, 128, 256
    x = (rnd(m, c, dtype=torch.float32, seed=1) + 0.3).to(dtype)
    w, b = rnd(n, c, dtype=dtype, seed=2, scale=c ** -0.5), rnd(n, dtype=torch.float32, seed=3, scale=0.3)
    g, be = rnd(c, dtype=torch.float32, seed=4, scale=0.2) + 1, rnd(c, dtype=torch.float32, seed=5, scale=0.3)
    wf, bf = fold_layernorm(w.float(), b, g, be)
    out = ops.gemm(x.cuda(), zero_sum_round(wf, dtype).cuda(), bias=bf.cuda(), ln_fold=1, ln_eps=1e-5, out_fp8=True)
    assert out.dtype == ops.FP8 and out.shape == (m, n)
    ref = F.linear(F.layer_norm(x.float(), (c,), g, be, 1e-5), w.float(), b)
    # e4m3: 3 mantissa bits -> half an ulp is 2^-4 relative (subnormal step 2^-9), on top of the 16-bit path's own error
    r8 = 2 ** -4 + 2 * TOL[dtype]["rtol"]
    torch.testing.assert_close(out.float().cpu(), ref, rtol=r8, atol=2 ** -9 + TOL[dtype]["atol"])
    big = ops.gemm((x * 300).to(dtype).cuda(), (w * 50).to(dtype).cuda(), out_fp8=True)          # saturates at +-448, never NaN
    assert torch.isfinite(big.float()).all() and float(big.float().abs().max()) == 448.0
    nseq, hw, hwp = 4, 125, 128
    vt = torch.zeros(n, nseq * hwp, dtype=torch.uint8, device="cuda").view(ops.FP8)
    ops.gemm(w.cuda(), x.cuda(), out=vt, out_col_pad=(hw, hwp), out_fp8=True)
    got = vt.float().cpu().reshape(n, nseq, hwp)
    torch.testing.assert_close(got[:, :, :hw], (w.float() @ x.float().T).reshape(n, nseq, hw), rtol=r8, atol=2 ** -9 + TOL[dtype]["atol"])
    assert float(got[:, :, hw:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_softmax_rescale_branch(ops, dtype):
    """one key in a LATER tile dominates a row: forces the online-softmax rescale (guide rule 26)."""
    l, d = 192, 64
    q, k, v = (rnd(1, l, d, dtype=dtype, seed=s) for s in (1, 2, 3))
    k[0, 150] = (q[0, 7].float() * 4).to(dtype)
    vt = v[0].T.contiguous()
    out = torch.empty(l, d, dtype=dtype, device="cuda")
    ops.attention(q[0].cuda(), k[0].cuda(), vt.cuda(), out, nseq=1, lq=l, heads=1, head_dim=d, mask=0, lk=l,
                  k_seq_stride=l, v_seq_stride=l)
    close(out[None], _sdpa(q, k, v, 1), dtype)

@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("growth", [6.0, 40.0, 400.0])
def test_attention_scores_growing_along_the_keys(ops, dtype, growth):
    """The softmax reference point is lazy and evaluated optimistically (a tile is accepted when its row sums stay below 2^14, the
    maximum is computed only when they do not): keys whose scores keep GROWING along the sequence force the slow path again and
    again, by a little (stays on the fast path with a stale reference), by a lot, and by enough to overflow exp2 (inf caught by
    the same check).  Scaled scores span `growth` * log2(e) / sqrt(d) * |q|^2 octaves over the sequence."""
    l, d = 640, 64
    q, k, v = (rnd(1, l, d, dtype=torch.float32, seed=s) for s in (1, 2, 3))
    ramp = torch.linspace(0.0, 1.0, l)[:, None]
    k = k * 0.3 + ramp * growth * q[0, 5:6] / q[0, 5].norm()          # key j leans towards query 5 by j / l * growth
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    vt = v[0].T.contiguous()
    out = torch.empty(l, d, dtype=dtype, device="cuda")
    ops.attention(q[0].cuda(), k[0].cuda(), vt.cuda(), out, nseq=1, lq=l, heads=1, head_dim=d, mask=0, lk=l,
                  k_seq_stride=l, v_seq_stride=l)
    assert bool(torch.isfinite(out.float()).all())
    close(out[None], _sdpa(q, k, v, 1), dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("l,growth", [(128, 0.0), (192, 0.0), (320, 0.0), (640, 6.0), (640, 40.0), (640, 400.0), (704, 400.0)])
def test_attention_pipelined_self_attention(ops, dtype, l, growth, monkeypatch):
    """attn_pipe_kernel (v_rows, whole 64-key tiles, >= 2 tiles): the softmax of tile t runs beside O += V(t-1) P(t-1), P is carried across
    the loop.  Two / three / five tiles (first-iteration, tail and epilogue paths), even and odd tile counts, a ragged query block, and scores
    growing along the keys (the slow path rescales O AFTER the iteration's PV).  Must match fp32 SDPA and -- same MFMA operands and
    summation order -- the un-pipelined kernel bit for bit (same process: the switch is read once per process, so that side runs the V^T route)."""
    d, heads, nseq = 64, 2, 2
    c = heads * d
    q, k, v = (rnd(nseq, l, c, dtype=torch.float32, seed=s) for s in (1, 2, 3))
    if growth:
        ramp = torch.linspace(0.0, 1.0, l)[None, :, None]
        k = k * 0.3 + ramp * growth * q[:, 5:6] / q[:, 5:6].norm(dim=-1, keepdim=True) / heads ** 0.5
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    qkv = torch.cat([q, k, v], dim=2).reshape(nseq * l, 3 * c).cuda()
    vt = v.permute(2, 0, 1).reshape(c, nseq * l).contiguous()
    kw = dict(nseq=nseq, lq=l, heads=heads, head_dim=d, mask=0, lk=l, k_seq_stride=l, v_seq_stride=l)
    out_t = torch.empty(nseq * l, c, dtype=dtype, device="cuda")
    ops.attention(qkv[:, :c], qkv[:, c:2 * c], vt.cuda(), out_t, **kw)
    out_r = torch.empty(nseq * l, c, dtype=dtype, device="cuda")
    ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], out_r, v_rows=True, **kw)
    assert bool(torch.isfinite(out_r.float()).all())
    close(out_r.view(nseq, l, c), _sdpa(q, k, v, heads), dtype)
    assert torch.equal(out_r, out_t), "the pipelined kernel must give the un-pipelined kernel's output bit for bit"


def test_attention_pipe_switch_off_gives_the_same_bits(ops):
    """TT_ATTN_PIPE=0 (read once per process) keeps attn_kernel for every key count: a child process with the switch off must produce the
    bytes this process gets from attn_pipe_kernel (same seeded inputs; 448 keys = 7 tiles, a ragged query block)."""
    import hashlib
    import os
    import subprocess
    import sys
    code = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
from this_and_that_vdm_amd import ops
g = torch.Generator().manual_seed(7)
nseq, l, heads, d = 2, 448, 3, 64
c = heads * d
qkv = (torch.randn(nseq * l, 3 * c, generator=g) * 1.5).to(torch.bfloat16).cuda()
out = torch.empty(nseq * l, c, dtype=torch.bfloat16, device="cuda")
ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], out, nseq=nseq, lq=l, heads=heads, head_dim=d, mask=0, lk=l, k_seq_stride=l, v_seq_stride=l, v_rows=True)
torch.cuda.synchronize()
print("HASH", hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for pipe in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, TT_ATTN_PIPE=pipe), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        hashes.append([ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("HASH")][0])
    assert hashes[0] == hashes[1]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("s", [1, 5, 78])
def test_attention_cross_spatial_and_temporal(ops, dtype, s):
    b, f, hw, heads, d = 2, 3, 40, 2, 64
    c = heads * d
    sp = (s + 7) // 8 * 8
    q = rnd(b * f, hw, c, dtype=dtype, seed=1)
    kc, vc = rnd(b, s, c, dtype=dtype, seed=2), rnd(b, s, c, dtype=dtype, seed=3)
    kpad = torch.zeros(b, sp, c, dtype=dtype)
    kpad[:, :s] = kc
    vt = torch.zeros(c, b * sp, dtype=dtype)
    vt.view(c, b, sp)[:, :, :s] = vc.permute(2, 0, 1)
    qd, kd, vd = q.reshape(-1, c).cuda(), kpad.reshape(-1, c).cuda(), vt.cuda()
    # spatial: frame n sees the context of batch n // f
    out = torch.empty(b * f * hw, c, dtype=dtype, device="cuda")
    ops.attention(qd, kd, vd, out, nseq=b * f, lq=hw, heads=heads, head_dim=d, mask=1, lk=s, k_seq_stride=sp,
                  v_seq_stride=sp, frames=f, ctx_batches=b)
    ref = _sdpa(q, kc.repeat_interleave(f, 0), vc.repeat_interleave(f, 0), heads)
    close(out.view(b * f, hw, c), ref, dtype)
    # temporal (reference quirk Q3): token (b, p) sees context (b*hw + p) % B
    out2 = torch.empty_like(out)
    ops.attention(qd, kd, vd, out2, nseq=b * f, lq=hw, heads=heads, head_dim=d, mask=2, lk=s, k_seq_stride=sp,
                  v_seq_stride=sp, frames=f, ctx_batches=b)
    sel = (torch.arange(b)[:, None] * hw + torch.arange(hw)[None]) % b                      # [b, hw]
    qt = q.view(b, f, hw, c).permute(0, 2, 1, 3).reshape(b * hw, f, c)
    ref2 = _sdpa(qt, kc[sel.reshape(-1)], vc[sel.reshape(-1)], heads)
    ref2 = ref2.view(b, hw, f, c).permute(0, 2, 1, 3).reshape(b * f, hw, c)
    close(out2.view(b * f, hw, c), ref2, dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("c,heads,b,f,hw,s,strided", [(320, 5, 2, 3, 300, 78, False), (640, 10, 2, 2, 130, 78, False), (1280, 20, 2, 3, 40, 5, False),
                                                     (320, 5, 1, 4, 256, 78, True), (128, 2, 2, 2, 37, 1, False)])
def test_attention_fused_query_projection(ops, dtype, c, heads, b, f, hw, s, strided):
    """tt_attention with the query projection fused in (TtAttnArgs.qx): Q = LN(x) Wq^T computed by the attention blocks themselves
    from LayerNorm-folded, row-permuted weights (packing.permute_q_rows).  Against (a) the separate projection GEMM (ln_fold = 1)
    followed by the attention kernel on the same operands and (b) torch (layer_norm -> linear -> SDPA) in fp32, for the spatial
    (mask 1) and the temporal (mask 2, quirk Q3 pairing) cross-attention; ragged query blocks (300 = 2 x 128 + 44 rows), rows of
    a strided view (the live residue class of the temporal block)."""
    from this_and_that_vdm_amd.packing import fold_layernorm, permute_q_rows, zero_sum_round
    d = 64
    sp = (s + 7) // 8 * 8
    rows = b * f * hw
    big = (rnd(2 * rows if strided else rows, c, dtype=torch.float32, seed=1, scale=1.5) + rnd(2 * rows if strided else rows, 1, dtype=torch.float32, seed=9)).to(dtype).cuda()
    x = big[1::2] if strided else big
    wq = rnd(c, c, dtype=dtype, seed=2, scale=c ** -0.5)
    g, be = rnd(c, dtype=torch.float32, seed=4, scale=0.2) + 1, rnd(c, dtype=torch.float32, seed=5, scale=0.3)
    wf, bf = fold_layernorm(wq.float(), None, g, be)
    wqf = zero_sum_round(wf, dtype)
    kc, vc = rnd(b, s, c, dtype=dtype, seed=6), rnd(b, s, c, dtype=dtype, seed=7)
    kpad = torch.zeros(b, sp, c, dtype=dtype)
    kpad[:, :s] = kc
    vt = torch.zeros(c, b * sp, dtype=dtype)
    vt.view(c, b, sp)[:, :, :s] = vc.permute(2, 0, 1)
    kd, vd = kpad.reshape(-1, c).cuda(), vt.cuda()
    q_ref = F.linear(F.layer_norm(x.float().cpu(), (c,), g, be, 1e-5), wq.float())
    wq_d, bq_d = wqf.cuda(), bf.cuda()
    wq_p, bq_p = permute_q_rows(wq_d), permute_q_rows(bq_d)
    assert torch.equal(permute_q_rows(wq_p), wq_d)                       # an involution
    for mask in (1, 2):
        kw = dict(nseq=b * f, lq=hw, heads=heads, head_dim=d, mask=mask, lk=s, k_seq_stride=sp, v_seq_stride=sp, frames=f, ctx_batches=b)
        q = ops.gemm(x, wq_d, bias=bq_d, ln_fold=1, ln_eps=1e-5)
        two = ops.attention(q, kd, vd, torch.empty(rows, c, dtype=dtype, device="cuda"), **kw)
        one = ops.attention(None, kd, vd, torch.full((rows, c), float("nan"), dtype=dtype, device="cuda"), qx=x, wq=wq_p, bq=bq_p, ln_eps=1e-5, **kw)
        tol = TOL[dtype]
        torch.testing.assert_close(one.float(), two.float(), rtol=tol["rtol"], atol=tol["atol"])
        if mask == 1:
            ref = _sdpa(q_ref.view(b * f, hw, c), kc.float().repeat_interleave(f, 0), vc.float().repeat_interleave(f, 0), heads)
        else:
            sel = (torch.arange(b)[:, None] * hw + torch.arange(hw)[None]) % b
            qt = q_ref.view(b, f, hw, c).permute(0, 2, 1, 3).reshape(b * hw, f, c)
            ref = _sdpa(qt, kc.float()[sel.reshape(-1)], vc.float()[sel.reshape(-1)], heads).view(b, hw, f, c).permute(0, 2, 1, 3).reshape(b * f, hw, c)
        torch.testing.assert_close(one.float().cpu().view(b * f, hw, c), ref, rtol=tol["rtol"] * 2, atol=tol["atol"] * 2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("frames,heads,d,hw", [(14, 3, 64, 19), (4, 1, 64, 19), (25, 2, 64, 19), (3, 2, 128, 19), (16, 5, 64, 19), (1, 2, 64, 7),
                                               (14, 5, 64, 1100)])        # (the last one: 11 000 units -- more than one grid-stride round of the MFMA kernel)
def test_temporal_self_attention(ops, dtype, frames, heads, d, hw):
    """tt_temporal_attention against fp32 SDPA over the frame axis: the matrix-core kernel (16-bit storage, head dimension 64, <= 16 frames:
    16 x 16 x 32 MFMAs, V transposed on the way out of LDS) and the per-lane kernel (everything else)."""
    b, c = 2, heads * d
    qkv = rnd(b * frames * hw, 3 * c, dtype=dtype, seed=1)
    out = torch.empty(b * frames * hw, c, dtype=dtype, device="cuda")
    ops.temporal_attention(qkv.cuda(), out, batch=b, frames=frames, hw=hw, heads=heads, head_dim=d)
    t = qkv.view(b, frames, hw, 3, c).permute(3, 0, 2, 1, 4).reshape(3, b * hw, frames, c)
    ref = _sdpa(t[0], t[1], t[2], heads).view(b, hw, frames, c).permute(0, 2, 1, 3).reshape(-1, c)
    close(out, ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c0,c1,fpg,h,w", [(64, 0, 1, 9, 11), (64, 32, 1, 9, 11), (320, 0, 4, 9, 11), (640, 320, 1, 9, 11),
                                            (1280, 1280, 1, 8, 14), (640, 0, 1, 16, 28),      # one-kernel statistics (<= 640 KiB/image)
                                            (320, 0, 1, 40, 56), (320, 320, 1, 32, 56)])      # partial + finalize kernels
def test_groupnorm(ops, dtype, c0, c1, fpg, h, w):
    """per-image statistics of small images run in one kernel (one block per image), larger ones and the cross-frame
    statistics of the temporal ResBlocks (fpg > 1) in the partial + finalize pair"""
    nimg = 4
    x0 = rnd(nimg, c0, h, w, dtype=dtype, seed=1, scale=3.0) + 1.5
    x1 = rnd(nimg, c1, h, w, dtype=dtype, seed=2) if c1 else None
    c = c0 + c1
    gamma, beta = rnd(c, dtype=torch.float32, seed=3) + 1, rnd(c, dtype=torch.float32, seed=4)
    xin = torch.cat([x0, x1], 1).float() if c1 else x0.float()
    b = nimg // fpg
    x5 = xin.view(b, fpg, c, h, w).permute(0, 2, 1, 3, 4)
    ref = F.silu(F.group_norm(x5, 32, gamma, beta, eps=1e-5)).permute(0, 2, 1, 3, 4).reshape(nimg, c, h, w)
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().cuda()
    t0, t1 = tok(x0), (tok(x1) if c1 else None)
    sc, sh = ops.groupnorm_stats(t0, t1, nimg, h * w, fpg, gamma.cuda(), beta.cuda(), 1e-5)
    y = ops.groupnorm_apply(t0, t1, nimg, h * w, sc, sh, True)
    close(y, ref.permute(0, 2, 3, 1).reshape(-1, c), dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c0,c1,h,w,silu", [(1280, 1280, 8, 14, True), (640, 0, 16, 28, True), (1280, 0, 4, 7, False), (64, 32, 5, 3, True),
                                            (320, 0, 40, 56, True), (640, 320, 32, 56, True), (320, 0, 32, 56, False), (1280, 640, 16, 28, True)])
def test_groupnorm_one_launch(ops, dtype, c0, c1, h, w, silu):
    """ops.groupnorm: statistics + apply in one launch for small per-image problems (tt_groupnorm_small: several blocks per image,
    each recomputing the image's statistics), the stats + apply pair otherwise (last case); equal to each other up to rounding
    and to torch; canary columns behind a wider output stride stay untouched."""
    nimg = 6
    x0 = rnd(nimg, c0, h, w, dtype=dtype, seed=1, scale=3.0) + 1.5
    x1 = rnd(nimg, c1, h, w, dtype=dtype, seed=2) if c1 else None
    c = c0 + c1
    gamma, beta = rnd(c, dtype=torch.float32, seed=3) + 1, rnd(c, dtype=torch.float32, seed=4)
    xin = torch.cat([x0, x1], 1).float() if c1 else x0.float()
    ref = F.group_norm(xin, 32, gamma, beta, eps=1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1).reshape(-1, c)
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().cuda()
    t0, t1 = tok(x0), (tok(x1) if c1 else None)
    y = ops.groupnorm(t0, t1, nimg, h * w, 1, gamma.cuda(), beta.cuda(), 1e-5, silu)
    close(y, ref, dtype, scale=2.0)
    sc, sh = ops.groupnorm_stats(t0, t1, nimg, h * w, 1, gamma.cuda(), beta.cuda(), 1e-5)
    y2 = ops.groupnorm_apply(t0, t1, nimg, h * w, sc, sh, silu)
    close(y, y2.float().cpu(), dtype, scale=2.0)
    small = bool(ops._lib.load().tt_groupnorm_small_supported(h * w, c, ops._code(dtype)))
    # >= 256 rows: one block per (image, group slice), any size; fewer: one block per image up to 640 KiB
    assert small == (256 <= h * w <= 4096 or h * w * c * (4 if dtype == torch.float32 else 2) <= 640 * 1024)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,frames,h,w", [(1280, 14, 4, 7), (2560, 14, 4, 7), (640, 4, 8, 14), (1280, 14, 8, 14)])
def test_groupnorm_cross_frame_one_launch(ops, dtype, c, frames, h, w):
    """ops.groupnorm with frames_per_group > 1 (TemporalResnetBlock: statistics per video over frames x h x w): with the opt-in
    bound GN_CROSS_MAX_ROWS = 512 a video of at most 512 rows is one "image" of the one-launch kernel (the frames of a video are contiguous rows); larger videos keep the
    partial + finalize + apply route (last case: 14 x 112 rows).  Both against torch's group_norm on [B, C, F, h, w]."""
    b = 2
    nimg = b * frames
    x = rnd(nimg, c, h, w, dtype=dtype, seed=1, scale=3.0) + 1.5
    gamma, beta = rnd(c, dtype=torch.float32, seed=3) + 1, rnd(c, dtype=torch.float32, seed=4)
    x5 = x.float().view(b, frames, c, h, w).permute(0, 2, 1, 3, 4)
    ref = F.silu(F.group_norm(x5, 32, gamma, beta, eps=1e-5)).permute(0, 2, 1, 3, 4).reshape(nimg, c, h, w)
    tok = x.permute(0, 2, 3, 1).reshape(-1, c).contiguous().cuda()
    lib = ops._lib.load()
    keep, ops.GN_CROSS_MAX_ROWS = ops.GN_CROSS_MAX_ROWS, 512          # the route is opt-in (measured: not faster in the step)
    try:
        y = ops.groupnorm(tok, None, nimg, h * w, frames, gamma.cuda(), beta.cuda(), 1e-5, True)
    finally:
        ops.GN_CROSS_MAX_ROWS = keep
    close(y, ref.permute(0, 2, 3, 1).reshape(-1, c), dtype, scale=2.0)
    sc, sh = ops.groupnorm_stats(tok, None, nimg, h * w, frames, gamma.cuda(), beta.cuda(), 1e-5)
    y2 = ops.groupnorm_apply(tok, None, nimg, h * w, sc, sh, True)
    close(y, y2.float().cpu(), dtype, scale=2.0)
    one_launch = frames * h * w <= 512 and bool(lib.tt_groupnorm_small_supported(frames * h * w, c, ops._code(dtype)))
    assert one_launch == (frames * h * w <= 512)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c", [64, 320, 1280])
def test_layernorm_and_fused_frame_embedding(ops, dtype, c):
    rows = 37 * 6
    x = rnd(rows, c, dtype=dtype, seed=1, scale=2.0)
    g, b = rnd(c, dtype=torch.float32, seed=2) + 1, rnd(c, dtype=torch.float32, seed=3)
    y = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-5)
    close(y, F.layer_norm(x.float(), (c,), g, b, 1e-5), dtype, scale=2.0)
    emb = rnd(3, c, dtype=torch.float32, seed=4)
    xs, y2 = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-5, rowvec=emb.cuda(), rows_per_vec=37, nvec=3)
    xsum = x.float() + emb.repeat(2, 1).repeat_interleave(37, 0)
    close(xs, xsum, dtype, scale=2.0)
    close(y2, F.layer_norm(xs.float().cpu(), (c,), g, b, 1e-5), dtype, scale=2.0)      # normalises the STORED sum


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("ln,geglu,m,n,k", [(1, 1, 8192 - 37, 6144 - 16, 128), (1, 0, 8192, 6144 - 48, 192), (0, 1, 8192, 6144, 128),
                                            (0, 0, 8192 - 200, 6144, 320),
                                            (1, 1, 3136, 10240, 1280),      # 12 whole tile rows + a ragged row of 64: ragged tiles last, to the idle CUs
                                            (0, 0, 3072 + 130, 10240, 256),   # 130 ragged rows: launched in two parts (12 whole tile rows + 130 rows on the tiled kernel)
                                            (1, 1, 10752, 5120, 640),         # 256x384: 840 tiles, 3.3 rounds (82 %): the GEGLU projections' lower bar
                                            (1, 0, 10752, 4864, 640),         # ... without GEGLU (798 tiles, 78 %): 40 tile rows on the persistent kernel + 512 rows on the tiled one
                                            (1, 1, 2688, 10240, 1280)])       # ... third level: 10 whole tile rows (400 tiles, 78 %) + 128 rows tiled
def test_gemm_persistent_pingpong_kernel(ops, dtype, ln, geglu, m, n, k):
    """gemm_pp.hip: tall-and-wide Linear problems without per-row epilogue operands (>= 460 tiles of 256 x 256: ~1.8 rounds of the 256 CUs) run on
    the persistent ping-pong kernel -- fused LayerNorm statistics, bias, GEGLU or plain 16-bit output, ragged last tile rows and
    columns.  Checked against torch fp32 and against the tiled kernel on the same operands (forced tile configuration)."""
    from this_and_that_vdm_amd.packing import fold_layernorm, pack_geglu, zero_sum_round
    lib = ops._lib.load()
    x = (rnd(m, k, dtype=torch.float32, seed=1, scale=1.5) + rnd(m, 1, dtype=torch.float32, seed=9, scale=1.5)).to(dtype)
    w, b = rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5), rnd(n, dtype=torch.float32, seed=3)
    if ln:
        g, be = rnd(k, dtype=torch.float32, seed=4, scale=0.2) + 1, rnd(k, dtype=torch.float32, seed=5, scale=0.3)
        wf, bf = fold_layernorm(w.float(), b, g, be)
        wq = zero_sum_round(wf, dtype)
        ref = F.linear(F.layer_norm(x.float(), (k,), g, be, 1e-5), w.float(), b)
    else:
        wq, bf = w, b
        ref = F.linear(x.float(), w.float(), b)
    if geglu:
        ref = ref[:, :n // 2] * F.gelu(ref[:, n // 2:])
        wq, bf = pack_geglu(wq, bf)
    args = dict(bias=bf.cuda(), geglu=bool(geglu), ln_fold=ln, ln_eps=1e-5)
    xd, wd = x.cuda(), wq.cuda()
    out = ops.gemm(xd, wd, **args)
    # (the plan query needs the same TtGemmArgs: go through the profiler hook)
    ops.PROFILE = []
    ops.gemm(xd, wd, **args)
    torch.cuda.synchronize()
    name = ops.PROFILE[0][0]
    ops.PROFILE = None
    assert name.startswith("gemm_pp_kernel<"), name
    lib.tt_gemm_set_tile_override(11)
    try:
        tiled = ops.gemm(xd, wd, **args)
    finally:
        lib.tt_gemm_set_tile_override(-1)
    tol = TOL[dtype]
    # vs torch: the limits of test_gemm_fused_layernorm_geglu_and_columns (fp16 with folded weights and value * gelu(gate): 5e-3)
    if dtype == torch.float16 and ln and geglu:
        rt = at = 5e-3
    else:
        k2 = 2.0 if (ln or geglu) else 1.0
        rt, at = tol["rtol"] * k2, tol["atol"] * k2 * (2.0 if dtype == torch.bfloat16 else 1.0)
    torch.testing.assert_close(out.float().cpu(), ref, rtol=rt, atol=at)
    # vs the tiled kernel on the same operands: the same arithmetic up to fp32 summation order / one fused multiply-add
    torch.testing.assert_close(out.float(), tiled.float(), rtol=tol["rtol"], atol=tol["atol"])


def _kernel_name(ops, *a, **kw):
    ops.PROFILE = []
    ops.gemm(*a, **kw)
    torch.cuda.synchronize()
    name = ops.PROFILE[0][0]
    ops.PROFILE = None
    return name


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("variant", ["no_bias", "strided_a", "wide_out", "split_tail_wide_out", "vae_vt"])
def test_gemm_persistent_pingpong_operand_variants(ops, dtype, variant):
    """Reachable variants of the persistent kernel beyond contiguous-A-with-bias: bias == NULL (the strip's bias slot is filled
    by the out-of-range zero fill of the LDS-DMA), a row-strided A view (lda0 != k0: the live-class projections run on
    t[cls::cb]), an output with ldo > n, the two-part launch (whole tile rows on the persistent kernel, ragged rows on the
    tiled kernel) into such an output, and the temporal VAE decoder's V^T projection (m = 512 channels, n = frames * h*w
    tokens >= 120 k, no bias: autoencoder_kl_temporal_decoder.py).  Each against torch fp32 and the tiled kernel."""
    lib = ops._lib.load()
    kw = {}
    if variant == "vae_vt":
        m, n, k = 512, 14 * 8960, 512                                # 14 frames x (80 x 112) tokens
    elif variant == "split_tail_wide_out":
        m, n, k = 3136, 10240, 320
    else:
        m, n, k = 8192 - 56, 6144, 192
    w = rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5).cuda()
    if variant == "strided_a":
        big = rnd(2 * m, k, dtype=dtype, seed=1).cuda()
        x = big[1::2]
        assert x.stride(0) == 2 * k
    else:
        x = rnd(m, k, dtype=dtype, seed=1).cuda()
    bias = None if variant in ("no_bias", "vae_vt") else rnd(n, dtype=torch.float32, seed=3)
    if bias is not None:
        kw["bias"] = bias.cuda()
    out = None
    if variant in ("wide_out", "split_tail_wide_out"):
        store = torch.full((m + 4, n + 72), 5.0, dtype=dtype, device="cuda")
        out = store[:m, 8:8 + n]
        assert out.stride(0) == n + 72
    ref = x.float().cpu() @ w.float().cpu().T + (bias if bias is not None else 0.0)
    name = _kernel_name(ops, x, w, out=out, **kw)
    assert name.startswith("gemm_pp_kernel<"), name
    got = ops.gemm(x, w, out=out, **kw)
    lib.tt_gemm_set_tile_override(11)
    try:
        tiled = ops.gemm(x, w, **kw)
    finally:
        lib.tt_gemm_set_tile_override(-1)
    close(got, ref, dtype, scale=2.0)
    tol = TOL[dtype]
    torch.testing.assert_close(got.float(), tiled.float(), rtol=tol["rtol"], atol=tol["atol"])
    if out is not None:                                              # nothing outside the [m, n] window was written
        assert (store[m:] == 5.0).all() and (store[:, :8] == 5.0).all() and (store[:, 8 + n:] == 5.0).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ratio", [50.0, 300.0])
def test_gemm_fused_layernorm_rows_with_large_row_means(ops, dtype, ratio):
    """The fused LayerNorm takes 1/sigma from ONE pass over the operand fragments, var = E[x^2] - E[x]^2 in fp32.
    16-bit storage (packed dot products on the raw fragments): the cancellation costs ~6e-8 * mean^2 / var * sqrt(K) of the
    output -- inside the usual per-kernel tolerance up to |row mean| = 50 sigma (the hidden states of the transformer blocks
    stay below ~5 sigma), and a graceful ~2 % of the output at 300 sigma (measured 3.0e-2 bf16 / 6.6e-2 fp16 absolute).
    fp32 storage (TT_F32 has to meet atol 1e-4): the sums are shifted by the row's first element (ln_stat_shifted), so the
    per-kernel tolerance holds at any row mean (unshifted it was 5e-3 at 50 sigma)."""
    from this_and_that_vdm_amd.packing import fold_layernorm, zero_sum_round
    m, c, n = 700, 320, 256
    x = (rnd(m, c, dtype=torch.float32, seed=1) + ratio * (1.0 + 0.1 * rnd(m, 1, dtype=torch.float32, seed=9))).to(dtype)
    w, b = rnd(n, c, dtype=dtype, seed=2, scale=c ** -0.5), rnd(n, dtype=torch.float32, seed=3)
    g, be = rnd(c, dtype=torch.float32, seed=4, scale=0.2) + 1, rnd(c, dtype=torch.float32, seed=5, scale=0.3)
    wf, bf = fold_layernorm(w.float(), b, g, be)
    out = ops.gemm(x.cuda(), zero_sum_round(wf, dtype).cuda(), bias=bf.cuda(), ln_fold=1, ln_eps=1e-5)
    ref = F.linear(F.layer_norm(x.float(), (c,), g, be, 1e-5), w.float(), b)
    tol = TOL[dtype]
    k = 2.0 * (2.0 if dtype == torch.float16 else 1.0)                  # (folded weights are rounded after the gamma product)
    rt, at = tol["rtol"] * k, tol["atol"] * k
    if dtype == torch.float32:
        # x * W'' with |x| ~ mean: the products cancel to O(1) from O(mean), so the fp32 rounding of the GEMM itself grows
        # linearly with the mean (measured 9.3e-5 at 50 sigma, 5.3e-4 at 300)
        rt, at = rt * ratio / 12, at * ratio / 12
    elif ratio > 50.0:
        rt = at = 5e-2
    torch.testing.assert_close(out.float().cpu(), ref, rtol=rt, atol=at)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,c,n", [(333, 64, 192), (2500, 320, 960), (9000, 640, 1280), (300, 1280, 2560)])
def test_gemm_fused_layernorm_rows(ops, dtype, m, c, n):
    """Linear(LayerNorm(x)) with the LayerNorm folded into the GEMM (tt_gemm ln_fold = 1): x reaches the kernel
    UN-normalised, gamma/beta live in the centred weights / bias (packing.fold_layernorm), 1/sigma comes from the operand
    fragments inside the K loop.  Rows carry a mean offset of ~1 sigma, as hidden states do."""
    from this_and_that_vdm_amd.packing import fold_layernorm, zero_sum_round
    x = (rnd(m, c, dtype=torch.float32, seed=1, scale=1.5) + rnd(m, 1, dtype=torch.float32, seed=9, scale=1.5)).to(dtype)
    w, b = rnd(n, c, dtype=dtype, seed=2, scale=c ** -0.5), rnd(n, dtype=torch.float32, seed=3)
    g, be = rnd(c, dtype=torch.float32, seed=4, scale=0.2) + 1, rnd(c, dtype=torch.float32, seed=5, scale=0.3)
    res = rnd(m, n, dtype=dtype, seed=6)
    wf, bf = fold_layernorm(w.float(), b, g, be)
    out = ops.gemm(x.cuda(), zero_sum_round(wf, dtype).cuda(), bias=bf.cuda(), residual=res.cuda(), ln_fold=1, ln_eps=1e-5)
    ref = F.linear(F.layer_norm(x.float(), (c,), g, be, 1e-5), w.float(), b) + res.float()
    # the folded weights are rounded AFTER the multiplication by gamma: allow their 2^-9 / 2^-12 relative rounding
    close(out, ref, dtype, scale=2.0) if dtype != torch.float16 else torch.testing.assert_close(out.float().cpu(), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_fused_layernorm_geglu_and_columns(ops, dtype):
    from this_and_that_vdm_amd.packing import fold_layernorm, pack_geglu, zero_sum_round
    m, c = 700, 320
    x = (rnd(m, c, dtype=torch.float32, seed=1) + 0.5).to(dtype)
    g, be = rnd(c, dtype=torch.float32, seed=4, scale=0.2) + 1, rnd(c, dtype=torch.float32, seed=5, scale=0.3)
    xn = F.layer_norm(x.float(), (c,), g, be, 1e-5)
    # fp16: the folded weights are rounded AFTER the multiplication by gamma and up to 1.5 ulp away from it (zero-sum
    # rounding); value * gelu(gate) multiplies two such ~1e-3 errors by O(3) operands: rtol = atol = 5e-3
    tol = dict(rtol=5e-3, atol=5e-3) if dtype == torch.float16 else dict(rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * (2 if dtype == torch.bfloat16 else 1))
    # GEGLU feed-forward projection (FeedForward.net[0]) behind norm3
    w, b = rnd(8 * c, c, dtype=dtype, seed=2, scale=c ** -0.5), rnd(8 * c, dtype=torch.float32, seed=3)
    wf, bf = fold_layernorm(w.float(), b, g, be)
    wp, bp = pack_geglu(zero_sum_round(wf, dtype), bf)
    out = ops.gemm(x.cuda(), wp.cuda(), bias=bp.cuda(), geglu=True, ln_fold=1, ln_eps=1e-5)
    h = xn @ w.float().T + b
    torch.testing.assert_close(out.float().cpu(), h[:, :4 * c] * F.gelu(h[:, 4 * c:]), **tol)
    # swapped V^T projection behind norm1: out[ch, token] = (Wv LN(x)^T)[ch, token]; beta is dropped here (it is folded
    # into to_out's bias by the block), sequences padded 175 -> 176 columns
    nseq, hw, hwp = 4, 175, 176
    wv = rnd(c, c, dtype=dtype, seed=7, scale=c ** -0.5)
    wvf, bv = fold_layernorm(wv.float(), None, g, be)
    vt = torch.zeros(c, nseq * hwp, dtype=dtype, device="cuda")
    ops.gemm(zero_sum_round(wvf, dtype).cuda(), x.cuda(), out=vt, out_col_pad=(hw, hwp), ln_fold=2, ln_eps=1e-5)
    ref = (wv.float() @ xn.T - bv[:, None]).reshape(c, nseq, hw)
    got = vt.float().cpu().reshape(c, nseq, hwp)
    torch.testing.assert_close(got[:, :, :hw], ref, **tol)
    assert float(got[:, :, hw:].abs().max()) == 0.0
    # ... and without padding (the non-transposing epilogue path)
    vt2 = ops.gemm(zero_sum_round(wvf, dtype).cuda(), x.cuda(), ln_fold=2, ln_eps=1e-5)
    torch.testing.assert_close(vt2.float().cpu(), wv.float() @ xn.T - bv[:, None], **tol)


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_rowvec(ops, dtype):
    x = rnd(37 * 6, 320, dtype=dtype, seed=1, scale=2.0)
    emb = rnd(3, 320, dtype=torch.float32, seed=4)
    y = ops.add_rowvec(x.cuda(), emb.cuda(), rows_per_vec=37, nvec=3)
    close(y, x.float() + emb.repeat(2, 1).repeat_interleave(37, 0), dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_small_linear_and_timestep_embedding(ops, dtype):
    from oracle.leaves import get_timestep_embedding
    t = torch.tensor([1.63777, -0.4622, 6.0, 200.0, 0.1])
    emb = ops.timestep_embedding(t.cuda(), 320)
    torch.testing.assert_close(emb.cpu(), get_timestep_embedding(t, 320, True, 0), rtol=1e-4, atol=2e-4)
    x = rnd(3, 320, dtype=torch.float32, seed=1)
    w, b = rnd(1280, 320, dtype=dtype, seed=2, scale=0.05), rnd(1280, dtype=torch.float32, seed=3)
    y = ops.small_linear(x.cuda(), w.cuda(), b.cuda(), act_in=True, act_out=True)
    ref = F.silu(F.silu(x) @ w.float().T + b)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-4)
    y2 = ops.small_linear(x.cuda(), w.cuda(), None, out=y.clone(), accumulate=True)
    torch.testing.assert_close(y2.cpu(), ref + x @ w.float().T, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_denoise_glue_matches_oracle_scheduler(ops, dtype):
    from oracle.scheduler import EulerDiscreteScheduler
    f, h, w = 5, 6, 7
    s = EulerDiscreteScheduler()
    s.set_timesteps(25)
    lat = rnd(1, f, 4, h, w, dtype=torch.float32, seed=1, scale=300.0)
    img = rnd(2, f, 4, h, w, dtype=torch.float32, seed=2)
    cond = rnd(f, 4, h, w, dtype=torch.float32, seed=3)
    step = 3
    s._step_index = step
    x = ops.prep_model_input(lat.cuda(), img.cuda(), cond.cuda(), s.sigmas.cuda(), step, 2, f, h, w, 64, dtype)
    t = s.timesteps[step]
    ref_in = torch.cat([s.scale_model_input(torch.cat([lat] * 2), t), img, torch.cat([cond, cond])[None].view(2, f, 4, h, w)], dim=2)
    got = x.float().cpu().view(2, f, h, w, 64)
    close(got[..., :12].permute(0, 1, 4, 2, 3), ref_in, dtype)
    assert float(got[..., 12:].abs().max()) == 0.0
    eps = rnd(2, f, 4, h, w, dtype=torch.float32, seed=4)
    g = torch.linspace(1, 3, f)
    u, c = eps.chunk(2)
    ref = s.step(u + g.view(1, f, 1, 1, 1) * (c - u), t, lat)
    eps_tok = eps.permute(0, 1, 3, 4, 2).reshape(-1, 4).contiguous().cuda()
    lat_d = lat.clone().cuda()
    ops.cfg_euler_step(eps_tok, lat_d, g.cuda(), s.sigmas.cuda(), step, 2, f, h, w)
    torch.testing.assert_close(lat_d.cpu(), ref, rtol=1e-5, atol=1e-3)
    # use_instructpix2pix: batch of 3 = (first-frame, cond, uncond), reference :698-702
    img3 = rnd(3, f, 4, h, w, dtype=torch.float32, seed=5)
    x3 = ops.prep_model_input(lat.cuda(), img3.cuda(), cond.cuda(), s.sigmas.cuda(), step, 3, f, h, w, 64, dtype)
    got3 = x3.float().cpu().view(3, f, h, w, 64)
    s._step_index = step
    ref3 = torch.cat([s.scale_model_input(torch.cat([lat] * 3), t), img3, torch.cat([cond] * 3).view(3, f, 4, h, w)], dim=2)
    close(got3[..., :12].permute(0, 1, 4, 2, 3), ref3, dtype)
    eps3 = rnd(3, f, 4, h, w, dtype=torch.float32, seed=6)
    e1, c3, u3 = eps3.chunk(3)
    s._step_index = step
    ref = s.step(u3 + g.view(1, f, 1, 1, 1) * (c3 - u3) + 7.5 * (c3 - e1), t, lat)
    lat_d = lat.clone().cuda()
    ops.cfg_euler_step(eps3.permute(0, 1, 3, 4, 2).reshape(-1, 4).contiguous().cuda(), lat_d, g.cuda(), s.sigmas.cuda(), step, 3,
                       f, h, w, image_guidance_scale=7.5)
    torch.testing.assert_close(lat_d.cpu(), ref, rtol=1e-5, atol=1e-3)
    with pytest.raises(ValueError):
        ops.cfg_euler_step(eps_tok, lat_d, g.cuda(), s.sigmas.cuda(), step, 3, f, h, w)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_plumbing(ops, dtype):
    x = rnd(3, 40, 5, 7, dtype=torch.float32, seed=1)
    tok = ops.nchw_to_tokens(x.cuda(), dtype, ld=64)
    assert tok.shape == (3 * 35, 64)
    close(tok[:, :40], x.permute(0, 2, 3, 1).reshape(-1, 40), dtype)
    back = ops.tokens_to_nchw(tok, 3, 40, 5, 7, torch.float32)
    close(back, x, dtype)
    a, b = rnd(64, 40, dtype=dtype, seed=2), rnd(64, 40, dtype=dtype, seed=3)
    close(ops.add_scaled(a.cuda(), b.cuda(), 0.5), a.float() + 0.5 * b.float(), dtype)


def test_errors_are_loud(ops):
    a = torch.zeros(8, 12, dtype=torch.float16, device="cuda")       # k not a multiple of 8
    with pytest.raises(RuntimeError, match="tt_gemm"):
        ops.gemm(a, torch.zeros(8, 12, dtype=torch.float16, device="cuda"))
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.gemm(torch.zeros(8, 16, dtype=torch.float16), torch.zeros(8, 16, dtype=torch.float16))


@pytest.mark.parametrize("cfg", list(range(21)))
def test_gemm_every_tile_configuration(ops, cfg):
    """each entry of the tile table in gemm.hip, forced, on a ragged linear and a 2-source conv (bf16)."""
    from this_and_that_vdm_amd import _lib
    from this_and_that_vdm_amd.packing import pack_conv3x3
    lib = _lib.load()
    dtype = torch.bfloat16
    try:
        assert lib.tt_gemm_set_tile_override(cfg) == 0
        m, n, k = 777, 328, 200
        a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
        bias, res = rnd(n, dtype=torch.float32, seed=3), rnd(m, n, dtype=dtype, seed=5)
        out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda())
        close(out, a.float() @ w.float().T + bias + res.float(), dtype, scale=2.0)
        nimg, c0, c1, cout, h, wd = 2, 72, 40, 168, 9, 13
        x0, x1 = rnd(nimg, c0, h, wd, dtype=dtype, seed=1), rnd(nimg, c1, h, wd, dtype=dtype, seed=2)
        wt = rnd(cout, c0 + c1, 3, 3, dtype=dtype, seed=3, scale=0.04)
        ref = F.conv2d(torch.cat([x0, x1], 1).float(), wt.float(), None, padding=1)
        t0 = x0.permute(0, 2, 3, 1).reshape(-1, c0).contiguous().cuda()
        t1 = x1.permute(0, 2, 3, 1).reshape(-1, c1).contiguous().cuda()
        out = ops.gemm(t0, pack_conv3x3(wt).cuda(), a1=t1, mode=1, conv=(nimg, h, wd, h, wd, 1, 0))
        close(out, ref.permute(0, 2, 3, 1).reshape(-1, cout), dtype)
    finally:
        lib.tt_gemm_set_tile_override(-1)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("m", [4101, 50176, 3 * 8192 - 31])
@pytest.mark.parametrize("epi", ["plain", "bias_res", "blend", "rowvec_parity", "rowvec_two_groups", "rowvec_blend"])
def test_gemm_square_320_streaming_kernel(ops, dtype, m, epi):
    """the W-in-registers persistent kernel that serves the 320 x 320 linears of the finest level (M >= 4096)."""
    import ctypes as C
    from this_and_that_vdm_amd import _lib
    k = n = 320
    a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
    bias, res = rnd(n, dtype=torch.float32, seed=3), rnd(m, n, dtype=dtype, seed=5)
    lib = _lib.load()
    try:
        assert lib.tt_gemm_set_streaming_square(1) == 0
        _run_square_320(ops, lib, dtype, m, epi, a, w, bias, res)
    finally:
        lib.tt_gemm_set_streaming_square(2)                  # back to the default: by size


def _run_square_320(ops, lib, dtype, m, epi, a, w, bias, res):
    import ctypes as C
    from this_and_that_vdm_amd import _lib
    k = n = 320
    g = _lib.TtGemmArgs()
    g.m, g.n, g.k0, g.mode = m, n, k, 0
    cfg = (C.c_int32 * 7)()
    assert lib.tt_gemm_plan(C.byref(g), cfg) == 0 and cfg[0] == 32 and cfg[1] == 320, list(cfg)
    out = torch.full((m + 8, n), 7.0, dtype=dtype, device="cuda")
    ref = a.float() @ w.float().T
    if epi.startswith("rowvec"):
        # round 6: a row vector with two distinct rows rides on the residual form -- even / odd rows (the temporal block's output
        # projection, rowvec_rows = 1, rowvec_mod = 2) or two row groups (the zero-context bias per CFG half; a multiple of 32 rows)
        rv = rnd(2, n, dtype=torch.float32, seed=4)
        half = (m // 2 + 31) // 32 * 32
        if epi == "rowvec_parity":
            kw, sel = dict(rowvec_rows=1, rowvec_mod=2), (torch.arange(m) & 1)
        else:
            kw, sel = dict(rowvec_rows=half), (torch.arange(m) >= half).long()
        g.rowvec, g.rowvec_rows, g.rowvec_mod, g.residual, g.ld_rowvec, g.ld_res = 16, kw["rowvec_rows"], kw.get("rowvec_mod", 0), 16, n, n
        assert lib.tt_gemm_plan(C.byref(g), cfg) == 0 and cfg[0] == 32 and cfg[1] == 320, list(cfg)      # still the streaming kernel
        r = res.cuda()
        extra = dict(blend=r, alpha=0.3) if epi == "rowvec_blend" else {}
        ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), acc_scale=0.5, rowvec=rv.cuda(), residual=r, out=out[:m], **kw, **extra)
        ref = (ref + bias) * 0.5 + rv[sel] + res.float()
        if extra:
            ref = 0.3 * res.float() + 0.7 * ref
        close(out[:m], ref, dtype, scale=2.0)
        assert (out[m:] == 7.0).all()
        g3 = _lib.TtGemmArgs()                                   # three groups: not this kernel
        g3.m, g3.n, g3.k0, g3.mode, g3.rowvec, g3.rowvec_rows, g3.residual, g3.ld_rowvec, g3.ld_res = m, n, k, 0, 16, (m // 3 + 31) // 32 * 32, 16, n, n
        assert lib.tt_gemm_plan(C.byref(g3), cfg) == 0 and cfg[0] != 32
        return
    if epi == "plain":
        ops.gemm(a.cuda(), w.cuda(), out=out[:m])
    elif epi == "bias_res":
        ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), acc_scale=0.5, residual=res.cuda(), out=out[:m])
        ref = (ref + bias) * 0.5 + res.float()
    else:
        r = res.cuda()
        ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=r, blend=r, alpha=0.3, out=out[:m])
        ref = 0.3 * res.float() + 0.7 * (ref + bias + res.float())
    close(out[:m], ref, dtype, scale=2.0)
    assert (out[m:] == 7.0).all()
    # strided operands (a column slice of a wider tensor, as the fused QK / context buffers are)
    wide = torch.zeros(m, 2 * k, dtype=dtype, device="cuda")
    wide[:, k:] = a.cuda()
    close(ops.gemm(wide[:, k:], w.cuda()), a.float() @ w.float().T, dtype, scale=2.0)


@pytest.mark.parametrize("n,rowvec", [(160, False), (128, False), (320, True)])
def test_gemm_tall_short_k(ops, n, rowvec):
    """the model's tall, short-K linears (M ~ 1e5, K = 64): ragged last tile, FiLM row vector spanning tile boundaries,
    nothing written behind the last row."""
    dtype = torch.bfloat16
    m, k = 784 * 128 - 37, 64
    a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
    bias, res = rnd(n, dtype=torch.float32, seed=3), rnd(m, n, dtype=dtype, seed=5)
    kw, ref = {}, a.float() @ w.float().T + bias
    if rowvec:
        rows_per = 1792
        rv = rnd((m + rows_per - 1) // rows_per, n, dtype=torch.float32, seed=4)
        kw = dict(rowvec=rv.cuda(), rowvec_rows=rows_per)
        ref = ref + rv.repeat_interleave(rows_per, 0)[:m]
    out = torch.full((m + 8, n), 7.0, dtype=dtype, device="cuda")         # canary rows behind the output
    ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), out=out[:m], **kw)
    close(out[:m], ref + res.float(), dtype, scale=2.0)
    assert (out[m:] == 7.0).all()


@pytest.mark.parametrize("m,n,k,cfg", [(12544, 640, 640, -1), (3136, 1280, 1280, -1), (12544, 2560, 320, -1), (50176, 320, 320, -1),
                                       (3136, 1280, 640, 1), (1500, 640, 1280, 2), (8192, 1024, 512, 13), (4096, 640, 512, 14), (3136, 1280, 5120, 20), (3000, 1280, 1344, 20)])
def test_gemm_is_repeatable_under_load(ops, m, n, k, cfg):
    """the K loop reads LDS with instructions the compiler cannot see and orders them against the LDS-DMA by counted
    waits and barriers only: a misplaced wait would show up as run-to-run differences.  40 launches back to back
    (other launches in flight, operands rotating) must be bit-identical to the first."""
    from this_and_that_vdm_amd import _lib
    lib = _lib.load()
    dtype = torch.bfloat16
    a = [rnd(m, k, dtype=dtype, seed=10 + i).cuda() for i in range(2)]
    w = rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5).cuda()
    res = rnd(m, n, dtype=dtype, seed=5).cuda()
    try:
        lib.tt_gemm_set_tile_override(cfg)
        first = [ops.gemm(a[i], w, residual=res).clone() for i in range(2)]
        close(first[0], a[0].float().cpu() @ w.float().cpu().T + res.float().cpu(), dtype, scale=2.0)
        outs = [ops.gemm(a[i % 2], w, residual=res) for i in range(40)]
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, first[i % 2]), f"launch {i} differs"
    finally:
        lib.tt_gemm_set_tile_override(-1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_split_k_is_used_and_exact(ops, dtype):
    """few tiles + long K -> the planner asks for workspace and splits K; fixed-order slab reduction."""
    import ctypes as C
    from this_and_that_vdm_amd import _lib
    from this_and_that_vdm_amd.packing import pack_conv3x3
    nimg, cin, cout, h, wd = 4, 640, 256, 7, 9                       # M = 252, K = 5760
    x = rnd(nimg, cin, h, wd, dtype=dtype, seed=1)
    wt = rnd(cout, cin, 3, 3, dtype=dtype, seed=2, scale=(9 * cin) ** -0.5)
    bias, res = rnd(cout, dtype=torch.float32, seed=3), rnd(nimg * h * wd, cout, dtype=dtype, seed=4)
    g = _lib.TtGemmArgs()
    g.m, g.n, g.k0, g.mode = nimg * h * wd, cout, cin, 1
    assert _lib.load().tt_gemm_ws_bytes(C.byref(g)) >= 2 * g.m * g.n * 4
    tok = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().cuda()
    args = dict(mode=1, conv=(nimg, h, wd, h, wd, 1, 0), bias=bias.cuda(), residual=res.cuda())
    a = ops.gemm(tok, pack_conv3x3(wt).cuda(), **args)
    b = ops.gemm(tok, pack_conv3x3(wt).cuda(), **args)
    assert torch.equal(a, b), "split-K must be bit-reproducible"
    ref = F.conv2d(x.float(), wt.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + res.float()
    close(a, ref, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k,rows_per,mod", [(1000, 320, 320, 1, 2),      # even / odd rows (the temporal block's two context classes)
                                                (12544 - 3, 640, 640, 1, 2), (3136, 1280, 1280, 1, 2), (784, 1280, 1280, 1, 2),
                                                (300, 128, 5120, 1, 2),      # split-K reduction path
                                                (1000, 320, 64, 1, 3),       # general small period: per-row loads
                                                (1000, 320, 64, 50, 3), (4096, 256, 128, 64, 2)])     # wrapped groups of >= 32 rows
def test_gemm_periodic_row_vector(ops, dtype, m, n, k, rows_per, mod):
    """TtGemmArgs.rowvec_mod: row r takes rowvec[(r / rowvec_rows) % rowvec_mod] -- against torch, with residual, in place and out of place
    bit for bit; the even / odd form must equal two launches on the strided row views (what the temporal block ran before)."""
    a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
    bias, rv = rnd(n, dtype=torch.float32, seed=3), rnd(mod, n, dtype=torch.float32, seed=4)
    res = rnd(m, n, dtype=dtype, seed=5)
    ad, wd, bd, rd = a.cuda(), w.cuda(), bias.cuda(), rv.cuda()
    out = ops.gemm(ad, wd, bias=bd, rowvec=rd, rowvec_rows=rows_per, rowvec_mod=mod, residual=res.cuda())
    idx = (torch.arange(m) // rows_per) % mod
    ref = a.float() @ w.float().T + bias + rv[idx] + res.float()
    close(out, ref, dtype, scale=2.0)
    inplace = res.cuda().clone()
    ops.gemm(ad, wd, bias=bd, rowvec=rd, rowvec_rows=rows_per, rowvec_mod=mod, residual=inplace, out=inplace)
    assert torch.equal(inplace, out)
    if rows_per == 1 and mod == 2 and m % 2 == 0:
        two = res.cuda().clone()
        for cls in range(2):
            ops.gemm(ad[cls::2], wd, bias=(bias + rv[cls]).cuda(), residual=two[cls::2], out=two[cls::2])
        tol = TOL[dtype]
        torch.testing.assert_close(out.float(), two.float(), rtol=tol["rtol"], atol=tol["atol"])
    with pytest.raises(RuntimeError):
        ops.gemm(ad, wd, bias=bd, rowvec=rd[:1], rowvec_rows=rows_per, rowvec_mod=mod)          # fewer vectors than the period


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["conv_l3_per_image", "tconv_l3_cross_frame_film", "conv_l3_cross_frame_res", "conv_l2_per_image", "linear_l3_c640"])
def test_gemm_groupnorm_inside_the_splitk_reduction(ops, dtype, case):
    """TtGemmArgs.gn_out (ABI 9): on the split-K plans of the two coarsest levels the reduction pass also writes SiLU(GroupNorm(out)) --
    per image (28 / 112 rows) and across the frames of a video (392 rows).  The plain output must not change (bit for bit), the
    normalised tensor must match torch group_norm of the STORED output and the statistics-pass kernels within storage rounding, be
    bit-reproducible, and ops.groupnorm must hand it out only for the parameters it was computed with."""
    g = dict(conv_l3_per_image=dict(nimg=28, h=4, w=7, cin=1280, n=1280, mode=1, film=True, fpg=1),
             tconv_l3_cross_frame_film=dict(nimg=28, h=4, w=7, cin=1280, n=1280, mode=2, film=True, fpg=14),
             conv_l3_cross_frame_res=dict(nimg=28, h=4, w=7, cin=1280, n=1280, mode=1, res=True, fpg=14),
             conv_l2_per_image=dict(nimg=28, h=8, w=14, cin=1280, n=1280, mode=1, film=True, fpg=1),
             linear_l3_c640=dict(nimg=28, h=4, w=7, cin=10240, n=640, mode=0, res=True, fpg=1))[case]
    frames, mode, n, fpg = 14, g["mode"], g["n"], g["fpg"]
    hw, k = g["h"] * g["w"], g["cin"]
    nimg = g["nimg"]
    rows = nimg * hw
    taps = {0: 1, 1: 9, 2: 3}[mode]
    a = rnd(rows, k, dtype=dtype, seed=1).cuda()
    w = rnd(n, taps * k, dtype=dtype, seed=2, scale=(taps * k) ** -0.5).cuda()
    kw = dict(bias=(rnd(n, dtype=torch.float32, seed=3) + 0.5).cuda(), mode=mode)
    if mode == 1:
        kw["conv"] = (nimg, g["h"], g["w"], g["h"], g["w"], 1, 0)
    if mode == 2:
        kw["tconv"] = (frames, hw)
    if g.get("film"):
        kw.update(rowvec=rnd(2, n, dtype=torch.float32, seed=4).cuda(), rowvec_rows=frames * hw)
    if g.get("res"):
        kw["residual"] = rnd(rows, n, dtype=dtype, seed=5).cuda()
    gamma, beta = (rnd(n, dtype=torch.float32, seed=6) * 0.2 + 1).cuda(), (rnd(n, dtype=torch.float32, seed=7) * 0.3).cuda()
    seg = fpg * hw
    plain = ops.gemm(a, w, **kw)
    y_old = ops.groupnorm(plain, None, nimg, hw, fpg, gamma, beta, 1e-5, True)            # the statistics-pass kernels
    out = ops.gemm(a, w, stats=seg, gn=(gamma, beta, 1e-5, True), **kw)
    assert torch.equal(out, plain), "the fused GroupNorm must not change the output"
    fz = getattr(out, "_tt_gn", None)
    assert fz is not None, f"{case}: no split-K reduction pass on this plan"
    y = ops.groupnorm(out, None, nimg, hw, fpg, gamma, beta, 1e-5, True)
    assert y.data_ptr() == fz[0].data_ptr(), "groupnorm() must hand out the producer's result"
    x = out.float()
    ref = F.silu(F.group_norm(x.view(nimg // fpg, seg, n).permute(0, 2, 1), 32, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(rows, n)
    close(y, ref.cpu(), dtype, scale=2.0)
    tol = TOL[dtype]
    torch.testing.assert_close(y.float(), y_old.float(), rtol=tol["rtol"], atol=tol["atol"])
    again = ops.gemm(a, w, stats=seg, gn=(gamma, beta, 1e-5, True), **kw)
    assert torch.equal(again._tt_gn[0], y), "bit-reproducible"
    # other parameters than the ones it was computed with: not handed out (the generic routes run)
    y2 = ops.groupnorm(out, None, nimg, hw, fpg, gamma, beta, 1e-5, False)
    assert y2.data_ptr() != fz[0].data_ptr()
    ref2 = F.group_norm(x.view(nimg // fpg, seg, n).permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1).reshape(rows, n)
    close(y2, ref2.cpu(), dtype, scale=2.0)
    if fpg == 1:
        y3 = ops.groupnorm(out, None, nimg, hw, frames, gamma, beta, 1e-5, True)         # another segment length
        assert y3.data_ptr() != fz[0].data_ptr()
    # an in-place overwrite drops the stale result
    ops.gemm(a, w, out=out, **kw)
    assert not hasattr(out, "_tt_gn")
    # a plan without a reduction pass ignores `gn` (and keeps the statistics route)
    big = ops.gemm(rnd(50176, 64, dtype=dtype, seed=8).cuda(), rnd(320, 64, dtype=dtype, seed=9).cuda(), stats=1792, gn=(gamma[:320].contiguous(), beta[:320].contiguous(), 1e-5, True))
    assert not hasattr(big, "_tt_gn")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["w320_conv", "w320_linear_res", "w320_tconv_blend", "w320h_conv", "tiled_linear", "tiled_tconv", "tiled_small",
                                  "splitk_conv_l3", "splitk_tconv_l3", "splitk_conv_l2", "tiled_wave_rows", "tiled_wave_rows_ragged", "w320h_conv_64"])
def test_gemm_output_statistics_and_groupnorm_from_tiles(ops, dtype, case):
    """TtGemmArgs.stats_out: per (row tile, column) sum / sum of squares of the STORED output, on every route that has the epilogue
    (256 x 320 and 128 x 320 big tiles, the tiled template), against sums over the stored tensor; then tt_groupnorm_tiles (one pass
    from those sums) against torch group_norm on the stored tensor, per-image and cross-frame (a video = `frames` images), and against
    the statistics-pass kernels (ops.groupnorm with TT_GN_TILES off) within storage rounding.  Launches without statistics must give
    the same output bit for bit."""
    frames = 4
    g = dict(w320_conv=dict(nimg=28, h=32, w=56, cin=64, n=320, mode=1), w320_linear_res=dict(rows=50176, k=128, n=320, mode=0, res=True),
             w320_tconv_blend=dict(nimg=28, h=32, w=56, cin=64, n=320, mode=2, blend=True), w320h_conv=dict(nimg=28, h=16, w=28, cin=64, n=640, mode=1),
             tiled_linear=dict(rows=3584, k=256, n=640, mode=0, res=True), tiled_tconv=dict(nimg=8, h=16, w=28, cin=128, n=640, mode=2),
             tiled_small=dict(rows=1024, k=64, n=96, mode=0),
             # the split-K routes of the two coarsest levels: the reduction kernel takes the sums on tiles of the caller's height (28-row images)
             splitk_conv_l3=dict(nimg=28, h=4, w=7, cin=1280, n=1280, mode=1, res=True, per_image=True), splitk_tconv_l3=dict(nimg=28, h=4, w=7, cin=1280, n=1280, mode=2, blend=True),
             splitk_conv_l2=dict(nimg=28, h=8, w=14, cin=1280, n=1280, mode=1),
             # the tiled template with one statistics tile per wave row (32 rows): segments that are no whole number of 128-row tiles -- 448-row
             # images of the second level; 1568-row videos of the third, whose 3136 rows also end in a ragged tile
             w320h_conv_64=dict(nimg=28, h=16, w=28, cin=64, n=640, mode=1, per_image=True),      # 448-row images on 128-row tiles: one statistics tile per 64-row wave row
             tiled_wave_rows=dict(rows=12544, k=128, n=640, mode=0, res=True, per_image=True),
             tiled_wave_rows_ragged=dict(nimg=28, h=8, w=14, cin=128, n=1280, mode=2, blend=True))[case]
    if case.startswith("splitk") or case == "tiled_wave_rows_ragged":
        frames = 14
    if dtype == torch.float32 and (case.startswith("w320") or case.startswith("splitk")):
        pytest.skip("the big-tile kernels and the split-K plans serve 16-bit storage")
    mode, n = g["mode"], g["n"]
    if mode == 0:
        rows, k = g["rows"], g["k"]
        hw = rows // 28 if rows % 28 == 0 else rows // 8
    else:
        hw, k = g["h"] * g["w"], g["cin"]
        rows = g["nimg"] * hw
    taps = {0: 1, 1: 9, 2: 3}[mode]
    a = rnd(rows, k, dtype=dtype, seed=1).cuda()
    w = rnd(n, taps * k, dtype=dtype, seed=2, scale=(taps * k) ** -0.5).cuda()
    bias = (rnd(n, dtype=torch.float32, seed=3) + 0.5).cuda()
    kw = dict(bias=bias, mode=mode)
    if mode == 1:
        kw["conv"] = (g["nimg"], g["h"], g["w"], g["h"], g["w"], 1, 0)
    if mode == 2:
        kw["tconv"] = (frames, hw)
    res = rnd(rows, n, dtype=dtype, seed=5).cuda()
    if g.get("res") or g.get("blend"):
        kw["residual"] = res
    if g.get("blend"):
        kw.update(blend=res, alpha=0.3)
    seg = hw if g.get("per_image") else frames * hw        # the consumer's segment: one image, or the frames of a video
    plain = ops.gemm(a, w, **kw)
    out = ops.gemm(a, w, stats=seg, **kw)
    assert torch.equal(out, plain), "the statistics epilogue must not change the output"
    st = getattr(out, "_tt_stats", None)
    assert st is not None, f"{case}: route without statistics epilogue"
    sbuf, r = st[:2]
    if case == "w320h_conv_64":
        assert r == 64
    elif case.startswith("w320h"):
        assert r == 128
    elif case.startswith("w320"):
        assert r == 256
    elif case.startswith("splitk"):
        assert r == {"splitk_conv_l3": 28, "splitk_tconv_l3": 98, "splitk_conv_l2": 112}[case]     # largest divisor <= 128 of the requested segment
    elif case.startswith("tiled_wave_rows"):
        assert r == 32 or (dtype == torch.float32 and seg % r == 0)          # (the fp32 template has its own tile shapes)
    elif case.startswith("tiled") and dtype != torch.float32:
        assert r in (64, 128)                                 # whole tiles of the planner's tile shape
    x = out.float()
    want = torch.stack([x.view(rows // r, r, n).sum(1), (x * x).view(rows // r, r, n).sum(1)], 1)
    torch.testing.assert_close(sbuf, want, rtol=2e-5, atol=2e-4)
    again = ops.gemm(a, w, stats=seg, **kw)
    assert torch.equal(again._tt_stats[0], sbuf), "tile sums must be bit-reproducible"
    # GroupNorm from the tile sums, per image (seg = hw rows) and across frames (seg = frames * hw rows)
    gamma, beta = (rnd(n, dtype=torch.float32, seed=6) * 0.2 + 1).cuda(), (rnd(n, dtype=torch.float32, seed=7) * 0.3).cuda()
    nimg = rows // hw
    for fpg in (1, frames):
        seg = fpg * hw
        if seg % r:
            continue
        y = ops.groupnorm(out, None, nimg, hw, fpg, gamma, beta, 1e-5, True)
        ref = F.silu(F.group_norm(x.view(nimg // fpg, seg, n).permute(0, 2, 1), 32, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(rows, n)
        close(y, ref.cpu(), dtype, scale=2.0)
        del out._tt_stats                                   # the statistics-pass kernels on the same tensor
        y_old = ops.groupnorm(out, None, nimg, hw, fpg, gamma, beta, 1e-5, True)
        out._tt_stats = st
        tol = TOL[dtype]
        torch.testing.assert_close(y.float(), y_old.float(), rtol=tol["rtol"], atol=tol["atol"])
    # an in-place overwrite drops the stale sums
    ops.gemm(a, w, out=out, **kw)
    assert not hasattr(out, "_tt_stats")
    # ... and so does a write through ANY view of the buffer by another ops.* launch (the write ledger in ops.py): the sums stay attached to
    # the base tensor object, but groupnorm() must not use them -- the statistics pass runs on the values that are there now
    fresh = ops.gemm(a, w, stats=seg, **kw)
    assert hasattr(fresh, "_tt_stats")
    ops.add_scaled(fresh, fresh, 1.0, out=fresh.view(rows, n))          # x <- x + x, written through a VIEW object of the same storage
    assert hasattr(fresh, "_tt_stats"), "the view write cannot see the base object's attribute"
    fpg0 = 1 if hw % r == 0 else frames
    if (fpg0 * hw) % r == 0:
        y_now = ops.groupnorm(fresh, None, nimg, hw, fpg0, gamma, beta, 1e-5, True)
        xf = fresh.float()
        ref_now = F.silu(F.group_norm(xf.view(nimg // fpg0, fpg0 * hw, n).permute(0, 2, 1), 32, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(rows, n)
        close(y_now, ref_now.cpu(), dtype, scale=2.0)
