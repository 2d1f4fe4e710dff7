"""GPU: gemm_w320.hip -- the 256 x 320 x 64 big-tile member of tt_gemm (problems of the finest UNet level: N = 320 t, M ~ 50176)
and its 128 x 320 variant gemm_w320h_kernel (two K halves per slab; the second level: M ~ 12544, and the live rows of the finest)
through the C ABI, every gather mode and epilogue operand, against (a) a plain PyTorch fp32 reference of the same op on the SAME
inputs and (b) the tiled kernel on the same operands (forced tile configuration: the same arithmetic up to fp32 summation order).
Tolerances as in tests/test_ops_gpu.py: fp16 rtol = atol = 1e-3, bf16 1.6e-2 (atol x 2 where outputs are O(4))."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES16 = [torch.float16, torch.bfloat16]
TOL = {torch.float16: dict(rtol=1e-3, atol=1e-3), torch.bfloat16: dict(rtol=1.6e-2, atol=1.6e-2)}


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
    torch.testing.assert_close(got.float().cpu(), ref.float(), rtol=tol["rtol"], atol=tol["atol"] * (scale if dtype == torch.bfloat16 else 1.0))


def both(ops, *a, **kw):
    """(w320 result, tiled-kernel result, kernel name of the default route)"""
    lib = ops._lib.load()
    lib.tt_gemm_set_big_tile(3)                           # the 128-row variant for every gather mode (default: conv3x3 only)
    ops.PROFILE = []
    out = ops.gemm(*a, **kw)
    torch.cuda.synchronize()
    name = ops.PROFILE[0][0]
    ops.PROFILE = None
    lib.tt_gemm_set_big_tile(0)                           # the planner's choice among the tiled kernels: the route these problems took before
    try:
        tiled = ops.gemm(*a, **kw)
    finally:
        lib.tt_gemm_set_big_tile(1)
    return out, tiled, name


def same_as_tiled(out, tiled, dtype):
    tol = TOL[dtype]
    torch.testing.assert_close(out.float(), tiled.float(), rtol=tol["rtol"], atol=tol["atol"])


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("m,n,k", [(46100, 320, 192), (50176, 640, 128), (50176 - 250, 960, 128),
                                   (200704 - 100, 320, 128),                                          # 784 tiles: four rounds of workgroups (64x112 latents)
                                   (12500, 640, 192), (25088 - 60, 320, 128), (12544, 1920, 128)])     # the last three: 128-row tiles
def test_linear_full_epilogue(ops, dtype, m, n, k):
    """bias, scale, row vector (groups of 3000 rows: a fragment row can straddle two), residual and a DISTINCT blend tensor;
    ragged last row tile (46100 = 180 x 256 + 20; 49926 = 195 x 256 + 6), one to three column tiles."""
    a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
    bias = rnd(n, dtype=torch.float32, seed=3)
    rows_per = 3000
    rowvec = rnd((m + rows_per - 1) // rows_per, n, dtype=torch.float32, seed=4)
    res, bl = rnd(m, n, dtype=dtype, seed=5), rnd(m, n, dtype=dtype, seed=6)
    kw = dict(bias=bias.cuda(), acc_scale=0.75, rowvec=rowvec.cuda(), rowvec_rows=rows_per, residual=res.cuda(), blend=bl.cuda(), alpha=0.3)
    out, tiled, name = both(ops, a.cuda(), w.cuda(), **kw)
    assert name.startswith("gemm_w320_kernel<" if m > 40000 else "gemm_w320h_kernel<") and name.endswith(", 0, 0>"), name
    ref = (a.float() @ w.float().T + bias) * 0.75 + rowvec.repeat_interleave(rows_per, 0)[:m] + res.float()
    ref = 0.3 * bl.float() + 0.7 * ref
    close(out, ref, dtype, scale=2.0)
    same_as_tiled(out, tiled, dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
def test_linear_plain_and_bias_only(ops, dtype):
    """no epilogue operand at all (bias == NULL: a 0-byte descriptor returns zeros), then bias only; K = 320 (5 slabs) and 64-deep
    K = 128 (2 slabs: the ring never reaches its steady state)."""
    for m, n, k in ((50176, 320, 320), (47000, 320, 128), (12544, 640, 640), (12000, 640, 128), (25088, 320, 320)):
        a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
        out, tiled, name = both(ops, a.cuda(), w.cuda())
        assert name.startswith("gemm_w320_kernel<" if m > 40000 else "gemm_w320h_kernel<"), name
        close(out, a.float() @ w.float().T, dtype, scale=2.0)
        same_as_tiled(out, tiled, dtype)
        bias = rnd(n, dtype=torch.float32, seed=3)
        out, tiled, _ = both(ops, a.cuda(), w.cuda(), bias=bias.cuda())
        close(out, a.float() @ w.float().T + bias, dtype, scale=2.0)
        same_as_tiled(out, tiled, dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("m,n", [(50176, 320), (12544, 640)])
def test_linear_two_sources_strided_in_place_self_blend(ops, dtype, m, n):
    """the 1x1 shortcut over a skip concat (two channel sources), A rows taken from a strided view, the output written IN PLACE
    over the residual (every lane reads the element it overwrites) with the AlphaBlender source == the residual; rows outside
    the [m, n] window stay untouched."""
    k0, k1 = 128, 64
    big = rnd(2 * m, k0, dtype=dtype, seed=1).cuda()
    a0 = big[0::2]
    a1 = rnd(m, k1, dtype=dtype, seed=7).cuda()
    w = rnd(n, k0 + k1, dtype=dtype, seed=2, scale=(k0 + k1) ** -0.5).cuda()
    bias = rnd(n, dtype=torch.float32, seed=3).cuda()
    x = rnd(m + 64, n + 16, dtype=dtype, seed=5).cuda()
    view = x[32:32 + m, 8:8 + n]
    kw = dict(a1=a1, bias=bias, residual=view, blend=view, alpha=0.4)
    ref_out, tiled, name = both(ops, a0, w, **kw)
    assert name.startswith("gemm_w320_kernel<" if m > 40000 else "gemm_w320h_kernel<"), name
    lin = torch.cat([a0.float().cpu(), a1.float().cpu()], 1) @ w.float().cpu().T + bias.cpu()
    xr = view.float().cpu()
    close(ref_out, 0.4 * xr + 0.6 * (lin + xr), dtype, scale=2.0)
    same_as_tiled(ref_out, tiled, dtype)
    keep = x.clone()
    lib = ops._lib.load()
    lib.tt_gemm_set_big_tile(3)                           # the same route as `ref_out`
    try:
        ops.gemm(a0, w, out=view, **kw)
    finally:
        lib.tt_gemm_set_big_tile(1)
    assert torch.equal(view, ref_out), "in place == out of place, bit for bit"
    assert torch.equal(x[:32], keep[:32]) and torch.equal(x[32 + m:], keep[32 + m:])
    assert torch.equal(x[:, :8], keep[:, :8]) and torch.equal(x[:, 8 + n:], keep[:, 8 + n:])


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("m,n,k", [(50176 - 100, 320, 320), (50176 - 100, 960, 320), (12544 - 50, 640, 640), (12544, 1920, 640), (25088, 320, 320)])
def test_linear_with_fused_layernorm(ops, dtype, m, n, k):
    """ln_fold = 1 (the Q / QKV projections of the L0 transformer blocks): 1/sigma of the A rows from the operand fragments."""
    from this_and_that_vdm_amd.packing import fold_layernorm, zero_sum_round
    x = (rnd(m, k, dtype=torch.float32, seed=1, scale=1.5) + rnd(m, 1, dtype=torch.float32, seed=9, scale=1.5)).to(dtype)
    w, b = rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5), rnd(n, dtype=torch.float32, seed=3)
    g, be = rnd(k, dtype=torch.float32, seed=4, scale=0.2) + 1, rnd(k, dtype=torch.float32, seed=5, scale=0.3)
    wf, bf = fold_layernorm(w.float(), b, g, be)
    wq = zero_sum_round(wf, dtype)
    res = rnd(m, n, dtype=dtype, seed=6)
    ref = F.linear(F.layer_norm(x.float(), (k,), g, be, 1e-5), w.float(), b) + res.float()
    kw = dict(bias=bf.cuda(), ln_fold=1, ln_eps=1e-5, residual=res.cuda())
    out, tiled, name = both(ops, x.cuda(), wq.cuda(), **kw)
    assert name.startswith("gemm_w320_kernel<" if m > 40000 else "gemm_w320h_kernel<") and name.endswith(", 0, 1>"), name
    tol = TOL[dtype]
    torch.testing.assert_close(out.float().cpu(), ref, rtol=tol["rtol"] * 2, atol=tol["atol"] * 2 * (2.0 if dtype == torch.bfloat16 else 1.0))
    same_as_tiled(out, tiled, dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("h,w,c0,c1,cout", [(32, 56, 64, 0, 320), (32, 56, 64, 64, 320), (32, 56, 128, 0, 640),
                                            (16, 28, 64, 64, 640), (16, 28, 320, 0, 640)])      # 16 x 28: the second level, 128-row tiles; C = 320: 5 slabs per tap
def test_conv3x3(ops, dtype, h, w, c0, c1, cout):
    """3x3 / stride 1 / pad 1 conv at the finest level's geometry (28 images of 32 x 56 = 50176 output rows: every tile holds
    image borders, tiles straddle images), one or two channel sources, bias + FiLM row (one vector per 14 frames) + residual."""
    from this_and_that_vdm_amd.packing import pack_conv3x3
    nimg, frames = 28, 14
    x0 = rnd(nimg, c0, h, w, dtype=dtype, seed=1)
    x1 = rnd(nimg, c1, h, w, dtype=dtype, seed=2) if c1 else None
    c = c0 + c1
    wt = rnd(cout, c, 3, 3, dtype=dtype, seed=3, scale=(9 * c) ** -0.5)
    bias = rnd(cout, dtype=torch.float32, seed=4)
    film = rnd(nimg // frames, cout, dtype=torch.float32, seed=7)
    res = rnd(nimg * h * w, cout, dtype=dtype, seed=8)
    xin = torch.cat([x0, x1], 1).float() if c1 else x0.float()
    ref = F.conv2d(xin, wt.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    ref = ref + film.repeat_interleave(frames * h * w, 0) + res.float()
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().cuda()
    kw = dict(a1=tok(x1) if c1 else None, mode=1, conv=(nimg, h, w, h, w, 1, 0), bias=bias.cuda(), rowvec=film.cuda(),
              rowvec_rows=frames * h * w, residual=res.cuda())
    out, tiled, name = both(ops, tok(x0), pack_conv3x3(wt).cuda(), **kw)
    assert name.startswith("gemm_w320_kernel<" if h == 32 else "gemm_w320h_kernel<") and name.endswith(", 1, 0>"), name
    close(out, ref, dtype, scale=2.0)
    same_as_tiled(out, tiled, dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("hw,cout", [(1792, 320), (448, 640)])
def test_temporal_conv(ops, dtype, hw, cout):
    """Conv3d (3,1,1) along the frame axis (2 videos x 14 frames x 1792 pixels = 50176 rows; the first / last frame of each video
    reads zeros for the missing tap) + bias + AlphaBlender with the residual as its source (the temporal ResBlock's epilogue)."""
    from this_and_that_vdm_amd.packing import pack_tconv3
    b, f, c = 2, 14, 64
    x = rnd(b, c, f, hw, 1, dtype=dtype, seed=1)
    wt = rnd(cout, c, 3, 1, 1, dtype=dtype, seed=2, scale=(3 * c) ** -0.5)
    bias = rnd(cout, dtype=torch.float32, seed=3)
    res = rnd(b * f * hw, cout, dtype=dtype, seed=5)
    conv = F.conv3d(x.float(), wt.float(), bias, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(b * f * hw, cout)
    ref = 0.35 * res.float() + 0.65 * (conv + res.float())
    tok = x[..., 0].permute(0, 2, 3, 1).reshape(b * f * hw, c).contiguous().cuda()
    r = res.cuda()
    out, tiled, name = both(ops, tok, pack_tconv3(wt).cuda(), mode=2, tconv=(f, hw), bias=bias.cuda(), residual=r, blend=r, alpha=0.35)
    assert name.startswith("gemm_w320_kernel<" if hw == 1792 else "gemm_w320h_kernel<") and name.endswith(", 2, 0>"), name
    close(out, ref, dtype, scale=2.0)
    same_as_tiled(out, tiled, dtype)


def test_planner_keeps_other_problems_off_the_big_tile(ops):
    """what gemm_w320 does not serve stays where it was: N not a multiple of 320, too few rows for a round of 256-row tiles,
    K not a multiple of 64, stride-2 / upsampling convs, fp32 storage, GEGLU."""
    import ctypes as C
    lib = ops._lib.load()
    dt = torch.bfloat16

    def name_of(a, w, **kw):
        ops.PROFILE = []
        ops.gemm(a, w, **kw)
        torch.cuda.synchronize()
        n = ops.PROFILE[0][0]
        ops.PROFILE = None
        return n

    z = lambda *s: torch.zeros(*s, dtype=dt, device="cuda")
    assert name_of(z(50176, 128), z(320, 128)).startswith("gemm_w320_kernel<")
    assert not name_of(z(50176, 128), z(256, 128)).startswith("gemm_w320")            # N
    assert not name_of(z(12544, 128), z(320, 128)).startswith("gemm_w320")            # 49 / 98 row tiles
    assert not name_of(z(12544, 128), z(640, 128)).startswith("gemm_w320")            # 98 x 2 tiles of 128 rows: convs only by default
    assert name_of(z(12544, 64), z(640, 9 * 64), mode=1, conv=(28, 16, 28, 16, 28, 1, 0)).startswith("gemm_w320h_kernel<")
    try:
        lib.tt_gemm_set_big_tile(3)
        assert name_of(z(12544, 128), z(640, 128)).startswith("gemm_w320h_kernel<")
    finally:
        lib.tt_gemm_set_big_tile(1)
    assert not name_of(z(50176, 72), z(320, 72)).startswith("gemm_w320")              # K % 64
    assert not name_of(z(50176 * 4, 64), z(320, 9 * 64), mode=1, conv=(28, 64, 112, 32, 56, 2, 0)).startswith("gemm_w320")   # stride 2
    assert not name_of(torch.zeros(50176, 128, device="cuda"), torch.zeros(320, 128, device="cuda")).startswith("gemm_w320")  # TT_F32
    assert not name_of(z(50176, 128), z(640, 128), geglu=True).startswith("gemm_w320")


# ---- split-K route of the 128-row kernel (the two coarsest UNet levels: 100 / 28 tiles of 128 x 320, S workgroups per tile)
@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("m,n,k", [(3136, 1280, 5120), (3000, 1280, 4160), (1600, 1920, 6144)])
def test_split_k_linear_full_epilogue(ops, dtype, m, n, k):
    """FF2 of the third level (3136 rows, K = 5120: two K slices of 40 slabs), a ragged row count with an odd slab count
    (65 slabs: 32 + 33), 13 row tiles x 6 column tiles in three slices; every epilogue operand goes through splitk_epilogue_kernel."""
    a, w = rnd(m, k, dtype=dtype, seed=1), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5)
    bias = rnd(n, dtype=torch.float32, seed=3)
    rows_per = 1000
    rowvec = rnd((m + rows_per - 1) // rows_per, n, dtype=torch.float32, seed=4)
    res, bl = rnd(m, n, dtype=dtype, seed=5), rnd(m, n, dtype=dtype, seed=6)
    kw = dict(bias=bias.cuda(), acc_scale=0.75, rowvec=rowvec.cuda(), rowvec_rows=rows_per, residual=res.cuda(), blend=bl.cuda(), alpha=0.3)
    out, tiled, name = both(ops, a.cuda(), w.cuda(), **kw)
    assert name.startswith("gemm_w320h_kernel<") and name.endswith(", 0, 0>"), name
    ref = (a.float() @ w.float().T + bias) * 0.75 + rowvec.repeat_interleave(rows_per, 0)[:m] + res.float()
    ref = 0.3 * bl.float() + 0.7 * ref
    close(out, ref, dtype, scale=2.0)
    same_as_tiled(out, tiled, dtype)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("h,w,c0,c1,cout", [(8, 14, 128, 320, 1280),      # 63 slabs in two slices: the second one starts inside tap 4, in the SECOND source
                                            (8, 13, 320, 0, 1280),       # 2912 rows: ragged last row tile; 45 slabs, 22 + 23
                                            (4, 7, 640, 640, 1280)])     # the coarsest level: 28 tiles x 9 slices of 20 slabs (one tap each)
def test_split_k_conv3x3(ops, dtype, h, w, c0, c1, cout):
    """3x3 convs of the third and fourth level through the split-K route: the K walk of a slice starts at an arbitrary
    (tap, source, k step); bias + FiLM row + residual in the second pass."""
    from this_and_that_vdm_amd.packing import pack_conv3x3
    nimg, frames = 28, 14
    x0 = rnd(nimg, c0, h, w, dtype=dtype, seed=1)
    x1 = rnd(nimg, c1, h, w, dtype=dtype, seed=2) if c1 else None
    c = c0 + c1
    wt = rnd(cout, c, 3, 3, dtype=dtype, seed=3, scale=(9 * c) ** -0.5)
    bias = rnd(cout, dtype=torch.float32, seed=4)
    film = rnd(nimg // frames, cout, dtype=torch.float32, seed=7)
    res = rnd(nimg * h * w, cout, dtype=dtype, seed=8)
    xin = torch.cat([x0, x1], 1).float() if c1 else x0.float()
    ref = F.conv2d(xin, wt.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    ref = ref + film.repeat_interleave(frames * h * w, 0) + res.float()
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().cuda()
    kw = dict(a1=tok(x1) if c1 else None, mode=1, conv=(nimg, h, w, h, w, 1, 0), bias=bias.cuda(), rowvec=film.cuda(),
              rowvec_rows=frames * h * w, residual=res.cuda())
    out, tiled, name = both(ops, tok(x0), pack_conv3x3(wt).cuda(), **kw)
    assert name.startswith("gemm_w320h_kernel<") and name.endswith(", 1, 0>"), name
    close(out, ref, dtype, scale=2.0)
    same_as_tiled(out, tiled, dtype)


def test_split_k_is_deterministic_and_in_place(ops):
    """the slabs are summed in a fixed order: two runs agree bit for bit; the second pass may write over its residual."""
    dtype = torch.bfloat16
    m, n, k = 3136, 1280, 5120
    a, w = rnd(m, k, dtype=dtype, seed=1).cuda(), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5).cuda()
    x = rnd(m, n, dtype=dtype, seed=5).cuda()
    bias = rnd(n, dtype=torch.float32, seed=3).cuda()
    lib = ops._lib.load()
    lib.tt_gemm_set_big_tile(3)                           # Linear problems take the split-K route under this knob only
    try:
        o1 = ops.gemm(a, w, bias=bias, residual=x)
        o2 = ops.gemm(a, w, bias=bias, residual=x)
        assert torch.equal(o1, o2)
        ops.PROFILE = []
        ops.gemm(a, w, bias=bias, residual=x, out=x)
        torch.cuda.synchronize()
        name = ops.PROFILE[0][0]
        ops.PROFILE = None
    finally:
        lib.tt_gemm_set_big_tile(1)
    assert name.startswith("gemm_w320h_kernel<"), name
    assert torch.equal(x, o1)


def test_split_k_default_route_is_the_third_level_conv(ops):
    """without any knob: the 3x3 conv at 3136 rows runs on the split-K route, FF2 of the same level and the 784-row conv do not."""
    dt = torch.bfloat16
    z = lambda *s: torch.zeros(*s, dtype=dt, device="cuda")

    def name_of(a, w, **kw):
        ops.PROFILE = []
        ops.gemm(a, w, **kw)
        torch.cuda.synchronize()
        n = ops.PROFILE[0][0]
        ops.PROFILE = None
        return n

    assert name_of(z(3136, 1280), z(1280, 9 * 1280), mode=1, conv=(28, 8, 14, 8, 14, 1, 0)).startswith("gemm_w320h_kernel<")
    assert not name_of(z(3136, 5120), z(1280, 5120)).startswith("gemm_w320")
    assert not name_of(z(784, 1280), z(1280, 9 * 1280), mode=1, conv=(28, 4, 7, 4, 7, 1, 0)).startswith("gemm_w320")


def test_split_k_without_workspace_falls_back_to_the_tiled_plan(ops, monkeypatch):
    """a caller of the C ABI that passes no workspace (ws = NULL) gets the un-split tiled plan, same result."""
    from this_and_that_vdm_amd.packing import pack_conv3x3
    dtype = torch.bfloat16
    nimg, h, w, c, cout = 28, 8, 14, 320, 1280
    x = rnd(nimg, c, h, w, dtype=dtype, seed=1)
    wt = rnd(cout, c, 3, 3, dtype=dtype, seed=3, scale=(9 * c) ** -0.5)
    tok = x.permute(0, 2, 3, 1).reshape(-1, c).contiguous().cuda()
    wp = pack_conv3x3(wt).cuda()
    kw = dict(mode=1, conv=(nimg, h, w, h, w, 1, 0))

    def run():
        ops.PROFILE = []
        o = ops.gemm(tok, wp, **kw)
        torch.cuda.synchronize()
        n = ops.PROFILE[0][0]
        ops.PROFILE = None
        return o, n

    split, name = run()
    assert name.startswith("gemm_w320h_kernel<"), name
    lib = ops._lib.load()
    monkeypatch.setattr(lib, "tt_gemm_ws_bytes", lambda g: 0)
    plain, name = run()
    assert name.startswith("gemm_kernel<"), name
    ref = F.conv2d(x.float(), wt.float(), None, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    close(split, ref, dtype, scale=2.0)
    close(plain, ref, dtype, scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("m,n,k", [(50176, 320, 320), (50176 - 77, 320, 128), (25088, 320, 320), (12544, 640, 128)])
def test_linear_even_odd_row_vector(ops, dtype, m, n, k):
    """rowvec_rows = 1, rowvec_mod = 2 on the big-tile kernels (the merged output projection of the temporal block's two context
    classes at the finest level): against torch fp32 and against the tiled kernel on the same operands."""
    a, w = rnd(m, k, dtype=dtype, seed=1).cuda(), rnd(n, k, dtype=dtype, seed=2, scale=k ** -0.5).cuda()
    bias, rv = rnd(n, dtype=torch.float32, seed=3).cuda(), rnd(2, n, dtype=torch.float32, seed=4).cuda()
    res = rnd(m, n, dtype=dtype, seed=5).cuda()
    out, tiled, name = both(ops, a, w, bias=bias, rowvec=rv, rowvec_rows=1, rowvec_mod=2, residual=res)
    assert "gemm_w320" in name, name
    ref = a.float() @ w.float().T + bias + rv[torch.arange(m, device="cuda") % 2] + res.float()
    close(out, ref.cpu(), dtype, scale=2.0)
    same_as_tiled(out, tiled, dtype)
