"""GPU: the "split16" product mode of TT_F32 (tt_gemm_set_f32_split, ABI 11): fp32 storage, every fp32 operand split on the fly
into fp16 hi + lo, three v_mfma_f32_32x32x16_f16 per product block (this_and_that_vdm_amd/csrc/gemm_kernel.h, `mma`).

  * every gather mode / epilogue of tt_gemm in this mode against an fp64 product of the SAME fp32 inputs, at the exact mode's
    tolerance (rtol = atol = 2e-5), and against the exact-fp32 MFMA route (they must agree to ~1e-6 relative and must NOT be
    bit-identical on a long-K problem: the variant really ran);
  * operand ranges: rows with a large mean (pre-LayerNorm hidden states), tiny operands (fp16 denormal lo parts), K tails;
  * the tiny VGL model against the oracle and the reference-produced vectors, every element inside rtol 1e-3 / atol 1e-4;
  * the switch is part of DenoiseLoop's graph key.
The full-size legs (32x56 VGL forward pair + two fused steps, 32x48, the 64x112 block) are parametrised over this mode in
tests/test_full_size_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_ops_gpu import rnd

pytestmark = pytest.mark.gpu
TOL = dict(rtol=2e-5, atol=2e-5)


@pytest.fixture()
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from this_and_that_vdm_amd import ops as o
    was = o.f32_split()
    o.set_f32_split(True)
    yield o
    o.set_f32_split(was)


def f32(*shape, seed, scale=1.0):
    return rnd(*shape, dtype=torch.float32, seed=seed, scale=scale)


def both(ops, fn):
    """fn() under split16 and under the exact-fp32 MFMA"""
    got = fn()
    ops.set_f32_split(False)
    try:
        exact = fn()
    finally:
        ops.set_f32_split(True)
    return got, exact


# (2048 x 2048 and 3000 x 1500: >= 256 tiles of 128 x 128, the NST = 2 configuration the full-size models run on; the others the 64 x 64 one)
@pytest.mark.parametrize("m,n,k", [(300, 72, 40), (1000, 320, 640), (129, 132, 64), (4096, 256, 1280), (50, 1280, 5120), (2048, 2048, 320),
                                   (3000, 1500, 136)])
def test_split16_linear_full_epilogue(ops, m, n, k):
    a, w = f32(m, k, seed=1), f32(n, k, seed=2, scale=k ** -0.5)
    bias, rows_per = f32(n, seed=3), 50
    rowvec = f32((m + rows_per - 1) // rows_per, n, seed=4)
    res, bl = f32(m, n, seed=5), f32(m, n, seed=6)
    run = lambda: ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), acc_scale=0.75, rowvec=rowvec.cuda(), rowvec_rows=rows_per,
                           residual=res.cuda(), blend=bl.cuda(), alpha=0.3)
    got, exact = both(ops, run)
    ref = (a.double() @ w.double().T + bias) * 0.75 + rowvec.repeat_interleave(rows_per, 0)[:m] + res
    ref = 0.3 * bl + 0.7 * ref
    torch.testing.assert_close(got.cpu().double(), ref, **TOL)
    torch.testing.assert_close(got, exact, rtol=1e-5, atol=1e-5)
    if k >= 640:
        assert not torch.equal(got, exact), "split16 and the exact-fp32 MFMA gave identical bits: the split variant did not run"


def test_split16_product_precision(ops):
    """plain products against fp64: relative L2 <= 3e-6 (probe: 1.1e-6 against 6.4e-7 for the exact-fp32 MFMA), also for rows
    with a large mean (|a| ~ 100 around 50), for tiny operands (the hi part a denormal, repaired by the lo part) and for operands far
    beyond fp16's own range (the parts sit on the scales 2^8 and 2^-3: |x| < 2^24)."""
    m, n, k = 512, 256, 1280
    for name, a, lim in (("unit", f32(m, k, seed=1), 3e-6), ("large mean", f32(m, k, seed=2, scale=100.0) + 50.0, 3e-6),
                         ("tiny", f32(m, k, seed=3, scale=1e-3), 3e-6), ("1e-6 (absolute floor 2^-28 per operand)", f32(m, k, seed=5, scale=1e-6), 1e-2),
                         ("beyond fp16's range", f32(m, k, seed=6, scale=2e5), 3e-6), ("up to 2^23", f32(m, k, seed=7, scale=1.5e6), 3e-6)):
        w = f32(n, k, seed=4, scale=0.03)
        got = ops.gemm(a.cuda(), w.cuda()).cpu().double()
        ref = a.double() @ w.double().T
        rel = float((got - ref).norm() / ref.norm())
        print(f"split16 product, {name}: rel-L2 vs fp64 {rel:.3e}")
        assert rel <= lim, (name, rel)


@pytest.mark.parametrize("stride,upsample", [(1, 0), (2, 0), (1, 1)])
def test_split16_conv3x3_two_sources(ops, stride, upsample):
    nimg, cin0, cin1, cout, h, w = 3, 32, 24, 40, 10, 14
    x0, x1 = f32(nimg, cin0, h, w, seed=1), f32(nimg, cin1, h, w, seed=2)
    wt, bias = f32(cout, cin0 + cin1, 3, 3, seed=3, scale=0.06), f32(cout, seed=4)
    from this_and_that_vdm_amd.packing import pack_conv3x3
    xin = torch.cat([x0, x1], 1).double()
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, wt.double(), bias.double(), stride=stride, padding=1)
    ho, wo = ref.shape[-2:]
    t0 = x0.permute(0, 2, 3, 1).reshape(-1, cin0).contiguous().cuda()
    t1 = x1.permute(0, 2, 3, 1).reshape(-1, cin1).contiguous().cuda()
    out = ops.gemm(t0, pack_conv3x3(wt).cuda(), a1=t1, mode=1, conv=(nimg, h, w, ho, wo, stride, upsample), bias=bias.cuda())
    torch.testing.assert_close(out.cpu().double(), ref.permute(0, 2, 3, 1).reshape(-1, cout), **TOL)


@pytest.mark.parametrize("nimg,c,cout,h,w", [(2, 640, 256, 8, 14), (14, 320, 640, 16, 28)])
def test_split16_conv3x3_full_width(ops, nimg, c, cout, h, w):
    """ResBlock convs at the widths of the second / third level: 9 x 640 deep on the 64 x 64 tiles, 6272 x 640 on the 128 x 128 ones"""
    x, wt, bias = f32(nimg, c, h, w, seed=1), f32(cout, c, 3, 3, seed=3, scale=(9 * c) ** -0.5), f32(cout, seed=4)
    from this_and_that_vdm_amd.packing import pack_conv3x3
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    tok = x.permute(0, 2, 3, 1).reshape(-1, c).contiguous().cuda()
    out = ops.gemm(tok, pack_conv3x3(wt).cuda(), mode=1, conv=(nimg, h, w, h, w, 1, 0), bias=bias.cuda())
    torch.testing.assert_close(out.cpu().double(), ref, **TOL)


@pytest.mark.parametrize("b,f,hw,c", [(2, 5, 12, 64), (2, 14, 448, 640)])
def test_split16_temporal_conv(ops, b, f, hw, c):
    x, wt, bias = f32(b, c, f, hw, 1, seed=1), f32(c, c, 3, 1, 1, seed=2, scale=0.07), f32(c, seed=3)
    ref = F.conv3d(x.double(), wt.double(), bias.double(), padding=(1, 0, 0))
    from this_and_that_vdm_amd.packing import pack_tconv3
    tok = x[..., 0].permute(0, 2, 3, 1).reshape(b * f * hw, c).contiguous().cuda()
    out = ops.gemm(tok, pack_tconv3(wt).cuda(), mode=2, tconv=(f, hw), bias=bias.cuda())
    torch.testing.assert_close(out.cpu().double(), ref[..., 0].permute(0, 2, 3, 1).reshape(b * f * hw, c), **TOL)


@pytest.mark.parametrize("m,c,n", [(500, 128, 256), (3000, 320, 960), (132, 1280, 384), (6272, 320, 960)])
def test_split16_layernorm_fold_rows_and_columns(ops, m, c, n):
    """ln_fold 1 (LayerNorm of the A rows: Q | K | V and GEGLU projections) and 2 (of the W rows: the swapped V^T projection),
    the statistics gathered next to the split products; rows with a mean of ~3 sigma."""
    from this_and_that_vdm_amd.packing import fold_layernorm, zero_sum_round
    x = f32(m, c, seed=1, scale=1.5) + f32(m, 1, seed=9, scale=4.0)
    w, b = f32(n, c, seed=2, scale=c ** -0.5), f32(n, seed=3, scale=0.3)
    g, be = f32(c, seed=4, scale=0.2) + 1, f32(c, seed=5, scale=0.3)
    wf, bf = fold_layernorm(w, b, g, be)
    wz = zero_sum_round(wf, torch.float32).cuda()
    ref = F.linear(F.layer_norm(x.double(), (c,), g.double(), be.double(), 1e-5), w.double(), b.double())
    out = ops.gemm(x.cuda(), wz, bias=bf.cuda(), ln_fold=1, ln_eps=1e-5)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-4, atol=1e-4)
    outt = ops.gemm(wz, x.cuda(), ln_fold=2, ln_eps=1e-5)                     # [n, m] = W LN(x)^T without the bias
    torch.testing.assert_close(outt.cpu().double(), (ref - bf.double()).T, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("m,c", [(260, 64), (4096, 320)])
def test_split16_geglu_and_in_place_residual(ops, m, c):
    a, w, b = f32(m, c, seed=1), f32(8 * c, c, seed=2, scale=c ** -0.5), f32(8 * c, seed=3)
    from this_and_that_vdm_amd.packing import pack_geglu
    wp, bp = pack_geglu(w, b)
    out = ops.gemm(a.cuda(), wp.cuda(), bias=bp.cuda(), geglu=True)
    h = a.double() @ w.double().T + b
    torch.testing.assert_close(out.cpu().double(), h[:, :4 * c] * F.gelu(h[:, 4 * c:]), **TOL)
    w2 = f32(c, 4 * c, seed=4, scale=(4 * c) ** -0.5)
    x = f32(m + 64, c, seed=5).cuda()
    ref = ops.gemm(out, w2.cuda(), residual=x[32:32 + m])
    keep = x.clone()
    ops.gemm(out, w2.cuda(), residual=x[32:32 + m], out=x[32:32 + m])
    assert torch.equal(x[32:32 + m], ref) and torch.equal(x[:32], keep[:32]) and torch.equal(x[32 + m:], keep[32 + m:])


def test_split16_is_transpose_safe(ops):
    n = 96
    a = torch.eye(64)
    w = (torch.arange(n * 64).reshape(n, 64) % 251 - 125).float() / 64
    out = ops.gemm(a.cuda(), w.cuda())
    torch.testing.assert_close(out.cpu(), w.T, rtol=0, atol=0)               # hi parts exact, lo parts zero


@torch.no_grad()
def test_split16_tiny_model_meets_the_north_star_tolerance(ops):
    """UNet (VL), GestureNet residuals and UNet (VGL) of the tiny configuration against the oracle AND the vectors the imported
    reference produced (tests/golden/tiny_vgl.npz), every element inside rtol 1e-3 / atol 1e-4."""
    from tests.parity_common import run_tiny_vgl_parity
    stats = run_tiny_vgl_parity(torch.float32, device="cuda:0", strict=True)
    print("tiny VGL, split16:", {k: (v["max_abs"], v["rel_l2"]) for k, v in stats.items()})
    for name, st in stats.items():
        assert st["frac_in_tol"] == 1.0 and st["rel_l2"] <= 2e-5, (name, st)
    stats128 = run_tiny_vgl_parity(torch.float32, device="cuda:0", name="tiny_vl_d128", strict=True)
    assert all(st["frac_in_tol"] == 1.0 for st in stats128.values())


@torch.no_grad()
def test_split16_switch_is_part_of_the_graph_key(ops):
    """a DenoiseLoop whose graph was captured in one product mode must not replay it in the other"""
    from tests.parity_common import build_pair, load_golden
    from this_and_that_vdm_amd.svd.denoise import DenoiseLoop
    from this_and_that_vdm_amd.svd.scheduling_euler_discrete import EulerDiscreteScheduler
    g = load_golden("tiny_vgl")
    p_unet, _, _, _ = build_pair("tiny_vgl", torch.float32, "cuda:0", False)
    sched = EulerDiscreteScheduler()
    sched.set_timesteps(4)
    f, h, w = 4, g["sample"].shape[-2], g["sample"].shape[-1]
    gen = torch.Generator().manual_seed(0)
    args = dict(latents=torch.randn(1, f, 4, h, w, generator=gen) * 700, image_latents=torch.randn(2, f, 4, h, w, generator=gen),
                encoder_hidden_states=g["encoder_hidden_states"][:2], added_time_ids=g["added_time_ids"][:2],
                guidance_scale=torch.linspace(1, 3, f).view(1, f, 1, 1, 1), sigmas=sched.sigmas, timesteps=sched.timesteps)
    loop = DenoiseLoop(p_unet, None, use_graph=True).begin(**args)
    loop.step()
    split = loop.result().clone()
    key_split = loop._key
    ops.set_f32_split(False)
    try:
        loop.begin(**args)
        assert loop._key != key_split and loop._graph is None
        loop.step()
        exact = loop.result().clone()
    finally:
        ops.set_f32_split(True)
    assert not torch.equal(split, exact)
    contrib = lambda z: z - args["latents"].cuda().reshape(z.shape) * float(sched.sigmas[1] / sched.sigmas[0])
    torch.testing.assert_close(contrib(split), contrib(exact), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("m,n,k", [(1000, 320, 640), (2048, 2048, 320), (300, 72, 40)])
def test_split16_presplit_operands_are_bit_identical_to_on_the_fly_conversion(ops, m, n, k):
    """TtGemmArgs.presplit: a packed weight handed over as fp16 (h, l) pairs (packing.presplit_f32, done once per pack) gives the
    bits the kernel's own conversion gives -- as the W operand (every Linear / conv) and as the A operand (the swapped V^T projection,
    whose LayerNorm statistics come from the other operand), also through a row slice of the pre-split matrix."""
    from this_and_that_vdm_amd.packing import PreSplitF32, fold_layernorm, presplit_f32
    a, w, bias = f32(m, k, seed=1, scale=3.0), f32(n, k, seed=2, scale=k ** -0.5), f32(n, seed=3)
    wp = presplit_f32(w.cuda())
    assert isinstance(wp, PreSplitF32) and wp.shape == w.shape and wp.dtype == torch.float32 and isinstance(wp[: n // 2], PreSplitF32)
    ref = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda())
    assert torch.equal(ops.gemm(a.cuda(), wp, bias=bias.cuda()), ref)
    half = (n // 2) // 4 * 4
    assert torch.equal(ops.gemm(a.cuda(), wp[:half], bias=bias[:half].cuda()), ref[:, :half])
    if k % 8 == 0 and m % 4 == 0:
        g, be = f32(k, seed=4, scale=0.2) + 1, f32(k, seed=5, scale=0.3)
        wf, _ = fold_layernorm(w, None, g, be)
        swapped = ops.gemm(wf.cuda(), a.cuda(), ln_fold=2, ln_eps=1e-5)
        assert torch.equal(ops.gemm(presplit_f32(wf.cuda()), a.cuda(), ln_fold=2, ln_eps=1e-5), swapped)
    with pytest.raises(RuntimeError, match="pre-split"):                       # the statistics cannot come from a pre-split operand
        ops.gemm(a.cuda(), wp, ln_fold=2, ln_eps=1e-5)
    ops.set_f32_split(False)
    try:
        with pytest.raises(RuntimeError, match="outside the split16 mode"):
            ops.gemm(a.cuda(), wp)
    finally:
        ops.set_f32_split(True)


@torch.no_grad()
def test_split16_models_pack_their_weights_pre_split(ops):
    """prepare() under split16 hands every packed GEMM weight over pre-split (and repacks when the mode is toggled)."""
    from tests.parity_common import build_pair
    from this_and_that_vdm_amd.packing import PreSplitF32
    p_unet, p_cn, _, _ = build_pair("tiny_vgl", torch.float32, "cuda:0", True)
    p_unet.prepare(); p_cn.prepare()
    blk = p_unet.down_blocks[0]
    res, tfm = blk.resnets[0].spatial_res_block, blk.attentions[0]
    for t in (res.w1, res.w2, tfm.w_in, tfm.transformer_blocks[0].wqkv, tfm.transformer_blocks[0].wqk, tfm.transformer_blocks[0].wv,
              tfm.transformer_blocks[0].ff.wg, tfm.transformer_blocks[0].q2.w, p_unet._w_in, p_unet._k_w, p_cn._zero[0][0], p_cn._zero_mid[0]):
        assert isinstance(t, PreSplitF32), type(t)
    gen = p_unet._pack_gen
    ops.set_f32_split(False)
    try:
        p_unet.prepare()
        assert p_unet._pack_gen == gen + 1 and not isinstance(p_unet.down_blocks[0].resnets[0].spatial_res_block.w1, PreSplitF32)
    finally:
        ops.set_f32_split(True)


# ---- the TT_F32 legs of the loop / VAE / pipeline tests, re-run with split-fp16 products (same assertions: every element inside the
# north-star tolerance).  The VAE decoder's weights are not pre-split (its pack path is its own): its launches convert both operands on the fly.
def test_split16_fused_loop_matches_oracle_loop(ops):
    from tests import test_denoise_loop_gpu as L
    L.test_fused_loop_matches_oracle_loop(True, torch.float32)
    L.test_fused_loop_matches_oracle_loop(False, torch.float32)


def test_split16_vae_decoder_matches_oracle(ops):
    from tests import test_vae_decoder_gpu as V
    V.test_decoder_matches_oracle(torch.float32, 1e-4)
    V.test_decoder_at_the_shipped_widths_matches_oracle(torch.float32, 1e-4)


def test_split16_pipeline_call_through_the_native_vae(ops):
    from tests import test_pipeline_gpu as P
    P.test_vgl_pipeline_call_produces_frames_through_the_native_vae()


# ---- tt_attention on fp32 storage follows the same switch (head dimension 64): both products as three fp16 MFMAs on split operands
@pytest.mark.parametrize("l,heads,d", [(100, 2, 64), (256, 1, 64), (28, 2, 64), (1000, 3, 64), (200, 1, 128)])
def test_split16_attention_self(ops, l, heads, d):
    from tests import test_ops_gpu as T
    T.test_attention_self(ops, torch.float32, l, heads, d)           # same assertion: rtol = atol = 2e-5 against SDPA in fp32


def test_split16_attention_other_paths(ops):
    """cross-attention (spatial / temporal pairing, 1 / 5 / 78 keys: ragged and masked tiles), the online-softmax rescale branch and
    scores that grow along the keys (lazy reference point, overflow recovery) -- the exact mode's tests, run on the split variant."""
    from tests import test_ops_gpu as T
    for s in (1, 5, 78):
        T.test_attention_cross_spatial_and_temporal(ops, torch.float32, s)
    T.test_attention_softmax_rescale_branch(ops, torch.float32)
    for growth in (6.0, 40.0):
        T.test_attention_scores_growing_along_the_keys(ops, torch.float32, growth)
    # growth = 400: raw scores reach several hundred, and a split product carries 2^-22 of its magnitude -- an absolute 1e-4 on such a
    # score, i.e. 1e-4 relative on its probability: finite, on the slow path, inside 2e-4 (the exact-fp32 MFMA holds 2e-5 there)
    l, d = 640, 64
    q, k, v = (f32(1, l, d, seed=s) for s in (1, 2, 3))
    ramp = torch.linspace(0.0, 1.0, l)[:, None]
    k = k * 0.3 + ramp * 400.0 * q[0, 5:6] / q[0, 5].norm()
    out = torch.empty(l, d, dtype=torch.float32, device="cuda")
    ops.attention(q[0].cuda(), k[0].cuda(), v[0].T.contiguous().cuda(), out, nseq=1, lq=l, heads=1, head_dim=d, mask=0, lk=l, k_seq_stride=l, v_seq_stride=l)
    assert bool(torch.isfinite(out).all())
    torch.testing.assert_close(out[None].cpu(), T._sdpa(q, k, v, 1), rtol=2e-4, atol=2e-4)


def test_split16_attention_is_the_split_variant_and_close_to_the_exact_one(ops):
    l, heads, d, nseq = 448, 2, 64, 2
    c = heads * d
    q, k, v = (f32(nseq, l, c, seed=s, scale=1.5) for s in (1, 2, 3))
    vt = v.permute(2, 0, 1).reshape(c, nseq * l).contiguous()
    def run():
        out = torch.empty(nseq * l, c, dtype=torch.float32, device="cuda")
        ops.attention(q.reshape(-1, c).cuda(), k.reshape(-1, c).cuda(), vt.cuda(), out, nseq=nseq, lq=l, heads=heads, head_dim=d, mask=0, lk=l,
                      k_seq_stride=l, v_seq_stride=l)
        return out
    got, exact = both(ops, run)
    assert not torch.equal(got, exact), "identical bits: the split variant did not run"
    torch.testing.assert_close(got, exact, rtol=1e-5, atol=1e-5)
