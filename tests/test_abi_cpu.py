"""CPU: libttvdm.so loads and exports every symbol include/ttvdm.h declares (no compute calls)."""
import os
import re

import pytest

from this_and_that_vdm_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_and_binding_agree(lib):
    text = open(os.path.join(REPO, "include", "ttvdm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", text))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ttvdm.h but not exported"


def test_version_and_arch(lib):
    assert lib.tt_abi_version() == 11
    assert lib.tt_target_arch() == b"gfx950"


def test_struct_layout_matches_header():
    """ctypes mirrors of TtGemmArgs / TtAttnArgs must have the C compiler's size."""
    import subprocess, tempfile, ctypes
    src = '#include <stdio.h>\n#include "ttvdm.h"\nint main(){printf("%zu %zu\\n", sizeof(TtGemmArgs), sizeof(TtAttnArgs));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe])
        a, b = map(int, subprocess.check_output([exe]).split())
    assert a == ctypes.sizeof(_lib.TtGemmArgs) and b == ctypes.sizeof(_lib.TtAttnArgs)


def test_ops_refuse_cpu_tensors(lib):
    import torch
    from this_and_that_vdm_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(8, 16, dtype=torch.float16), torch.zeros(8, 16, dtype=torch.float16))


def test_gemm_dispatch_rules_of_the_persistent_kernel(lib):
    """tt_gemm_plan is host-only: which problems go to gemm_pp_kernel (reported as stages = 0, tile 256 x 256 x 64).
    The GEGLU projections of the three finest levels do (the 8 x 14 one through its 12 whole tile rows); anything with a per-row
    epilogue operand, fp32 storage, a conv mode, too few tiles or a poorly filled last round stays on the tiled kernels."""
    import ctypes as C

    def plan(m, n, k, dtype, **kw):
        g = _lib.TtGemmArgs()
        g.m, g.n, g.k0, g.mode, g.dtype = m, n, k, 0, dtype
        g.lda0, g.ldw, g.ldo = k, k, n
        g.ln_eps = 1e-5
        for name, v in kw.items():
            setattr(g, name, v)
        cfg = (C.c_int32 * 7)()
        assert lib.tt_gemm_plan(C.byref(g), cfg) == 0
        return list(cfg), lib.tt_gemm_ws_bytes(C.byref(g))

    from this_and_that_vdm_amd import ops
    bf16, f32 = ops.TT_BF16, ops.TT_F32
    pp = [256, 256, 64, 0]
    for m, n, k in ((50176, 2560, 320), (12544, 5120, 640), (3136, 10240, 1280), (200704, 2560, 320)):
        cfg, ws = plan(m, n, k, bf16, ln_fold=1, geglu=1)
        assert cfg[:4] == pp and ws == 0, (m, n, k, cfg)
    assert plan(3072, 10240, 1280, bf16)[0][:4] == pp                      # 480 tiles: 1.9 rounds, 94 % full
    assert plan(10752, 5120, 640, bf16, ln_fold=1, geglu=1)[0][:4] == pp   # 256x384: 840 tiles = 3.3 rounds, 82 % (the GEGLU projections' bar is 400 tiles / 75 %)
    assert plan(2688, 10240, 1280, bf16, ln_fold=1, geglu=1)[0][:4] == pp  # ... its third level: 10 whole tile rows = 400 tiles (78 %) + 128 rows on the tiled kernel
    assert plan(10752, 4864, 640, bf16, ln_fold=1)[0][:4] == pp            # without GEGLU (460 tiles / 90 %): 798 tiles, 78 % -> 40 tile rows (2.97 rounds) + 512 rows on the tiled kernel
    assert plan(784, 10240, 1280, bf16, ln_fold=1, geglu=1)[0][:4] != pp   # the coarsest level: 120 tiles
    assert plan(50176, 960, 320, bf16, ln_fold=1)[0][:4] != pp             # 784 tiles: 4 rounds at 77 %
    assert plan(784, 10240, 1280, bf16, ln_fold=1, geglu=1)[0][:4] != pp   # 160 tiles
    assert plan(50176, 2560, 320, f32)[0][:4] != pp                        # fp32 storage
    assert plan(50176, 2560, 320, bf16, residual=1)[0][:4] != pp           # per-row epilogue operand (any non-null pointer)
    assert plan(50176, 2560, 328, bf16)[0][:4] != pp                       # K not a multiple of 64


def test_gemm_dispatch_rules_of_the_big_tile_kernel(lib):
    """tt_gemm_plan, host-only: which problems go to gemm_w320_kernel (reported as tile 256 x 320 x 64, stages 0, 4 x 2 waves):
    16-bit problems with N = 320 t whose 256-row tiles fill >= 60 % of the CU x round slots (TT_W320_MIN_FILL; 168 tiles at the
    reference's default 256x384 qualify) -- Linear (also with the LayerNorm fold,
    two sources, residual / blend / row vector), conv3x3 stride 1, temporal conv; the knob tt_gemm_set_big_tile(0) turns it off."""
    import ctypes as C
    from this_and_that_vdm_amd import ops

    def plan(m, n, k, dtype=ops.TT_BF16, **kw):
        g = _lib.TtGemmArgs()
        g.m, g.n, g.k0, g.mode, g.dtype = m, n, k, 0, dtype
        g.lda0, g.ldw, g.ldo = k, k, n
        g.ln_eps = 1e-5
        for name, v in kw.items():
            setattr(g, name, v)
        cfg = (C.c_int32 * 7)()
        assert lib.tt_gemm_plan(C.byref(g), cfg) == 0
        return list(cfg), lib.tt_gemm_ws_bytes(C.byref(g))

    w320 = [256, 320, 64, 0, 4, 2, 1]
    conv = dict(mode=1, nimg=28, hin=32, win=56, hout=32, wout=56, stride=1, upsample=0)
    assert plan(50176, 320, 1280, residual=16, ld_res=320) == (w320, 0)                   # FF2 at 32x56
    assert plan(50176, 320, 320, residual=16, ld_res=320, blend=16, ld_blend=320)[0] == w320
    assert plan(50176, 960, 320, ln_fold=1)[0] == w320                                   # LayerNorm-folded QKV: 588 tiles, 77 %
    assert plan(50176, 320, 320, k1=320, lda1=320, **conv)[0] == w320                    # conv over a skip concat
    assert plan(50176, 320, 320, mode=2, frames=14, hw=1792)[0] == w320
    assert plan(200704, 320, 640)[0] == w320                                             # 64x112 latents: 784 tiles
    assert plan(200704, 320, 320)[0][:2] == [32, 320]                                    # ... whose 320 x 320 linears take the 32-row streaming kernel (from 131072 rows on)
    assert plan(50176, 320, 320, rowvec=16, rowvec_rows=1792, ld_rowvec=320)[0] == w320
    assert plan(50176, 320, 320, rowvec=16, rowvec_rows=16, ld_rowvec=320)[0] != w320     # row-vector groups shorter than a fragment row
    assert plan(50176, 320, 320, rowvec=16, rowvec_rows=1, rowvec_mod=2, ld_rowvec=320)[0] == w320   # ... except the even / odd form (ABI 7)
    assert plan(43008, 320, 320)[0] == w320                                              # 32x48 latents (256x384): 168 tiles, 66 %
    w320h = [128, 320, 64, 0, 2, 2, 1]
    l1conv = dict(k1=640, lda1=640, mode=1, nimg=28, hin=16, win=28, hout=16, wout=28, stride=1)
    assert plan(12544, 640, 640, **l1conv)[0] == w320h                                   # conv3x3 of the second level: 98 x 2 tiles of 128 rows
    assert plan(25088, 320, 320)[0] != w320h and plan(12544, 640, 2560, residual=16, ld_res=640)[0] != w320h   # linears: not by default
    assert plan(12544, 320, 320)[0][:2] not in ([256, 320], [128, 320])                  # 98 tiles of 128 rows
    assert plan(3136, 1280, 1280)[0][:2] not in ([256, 320], [128, 320])
    assert plan(50176, 640, 320, geglu=1)[0] != w320
    assert plan(50176, 320, 320, dtype=ops.TT_F32)[0] != w320
    assert plan(50176, 320, 72)[0] != w320                                               # K % 64
    assert plan(50176, 320, 320, **dict(conv, stride=2, hin=64, win=112))[0] != w320
    assert plan(50176, 320, 320, **dict(conv, upsample=1, hin=16, win=28))[0] != w320
    assert plan(50176, 320, 320, out_f32=1)[0] != w320
    assert plan(50176, 320, 320, residual=4, ld_res=320)[0] != w320                      # 16-bit epilogue operands must be 8-byte aligned
    assert plan(50176, 2560, 320, ln_fold=1, geglu=1)[0][:4] == [256, 256, 64, 0]        # the GEGLU projections stay on gemm_pp
    try:
        assert lib.tt_gemm_set_big_tile(0) == 0
        assert plan(50176, 320, 1280)[0] != w320 and plan(12544, 640, 640, **l1conv)[0] != w320h
        assert lib.tt_gemm_set_big_tile(2) == 0                                          # the 256-row kernel only
        assert plan(50176, 320, 1280)[0] == w320 and plan(12544, 640, 640, **l1conv)[0] != w320h
        assert lib.tt_gemm_set_big_tile(3) == 0                                          # the variant for every mode
        assert plan(25088, 320, 320)[0] == w320h and plan(12544, 640, 2560, residual=16, ld_res=640)[0] == w320h
    finally:
        lib.tt_gemm_set_big_tile(1)
    assert plan(50176, 320, 1280)[0] == w320 and plan(12544, 640, 640, **l1conv)[0] == w320h and plan(25088, 320, 320)[0] != w320h


def test_gemm_dispatch_rules_of_the_split_k_big_tile_route(lib):
    """tt_gemm_plan / tt_gemm_ws_bytes, host-only: the 128 x 320 kernel with S K-slices per tile (cfg[6] = S) takes the 3x3 convs
    of the third level (3136 rows at 32x56 latents) when the caller supplies the workspace; tt_gemm_set_big_tile(3) opens the route to
    Linear problems and the 28-tile level (measured +-0 there: off by default)."""
    import ctypes as C
    from this_and_that_vdm_amd import ops

    def args(m, n, k, **kw):
        g = _lib.TtGemmArgs()
        g.m, g.n, g.k0, g.mode, g.dtype = m, n, k, 0, ops.TT_BF16
        g.lda0, g.ldw, g.ldo = k, k, n
        g.ln_eps = 1e-5
        for name, v in kw.items():
            setattr(g, name, v)
        return g

    def plan(m, n, k, ws=True, **kw):
        g = args(m, n, k, **kw)
        need = lib.tt_gemm_ws_bytes(C.byref(g))
        if ws and need:
            g.ws, g.ws_bytes = 256, need                      # a fake (aligned) pointer: the plan only looks at its presence and size
        cfg = (C.c_int32 * 7)()
        assert lib.tt_gemm_plan(C.byref(g), cfg) == 0
        return list(cfg), need

    split = lambda s: [128, 320, 64, 0, 2, 2, s]
    conv = lambda hh, ww: dict(mode=1, nimg=28, hin=hh, win=ww, hout=hh, wout=ww, stride=1, upsample=0)
    assert plan(3136, 1280, 1280, **conv(8, 14)) == (split(2), 2 * 3136 * 1280 * 4)                 # conv3x3 of the third level: 100 tiles x 2 slices
    assert plan(3136, 1280, 1280, k1=1280, lda1=1280, **conv(8, 14))[0] == split(2)
    assert plan(3136, 1280, 1280, ws=False, **conv(8, 14))[0][:2] == [128, 128]                    # no workspace: an un-split tiled plan
    assert plan(3136, 1280, 5120, residual=16, ld_res=1280)[0][:2] == [128, 128]                   # FF2: measured +-0, stays tiled by default
    assert plan(784, 1280, 1280, **conv(4, 7))[0][:2] == [128, 128]                                # 28 tiles: measured +-0, stays tiled by default
    assert plan(3136, 1280, 64, **conv(8, 14))[0][:2] == [128, 128]                                # 9 slabs: too short for two slices
    assert plan(12544, 640, 640, **conv(16, 28))[0] == [128, 320, 64, 0, 2, 2, 1]                  # the second level: 196 tiles, no slices
    try:
        assert lib.tt_gemm_set_big_tile(3) == 0                                                    # the variant for every mode: Linear and the 28-tile level too
        assert plan(3136, 1280, 5120, residual=16, ld_res=1280) == (split(2), 2 * 3136 * 1280 * 4)
        assert plan(784, 1280, 1280, **conv(4, 7)) == (split(9), 9 * 784 * 1280 * 4)               # 28 tiles x 9 slices of 20 slabs
        assert plan(784, 1280, 1280, k1=1280, lda1=1280, **conv(4, 7))[0] == split(9)
        assert plan(3136, 1280, 2560)[0][1] != 320                                                 # 40 slabs: too short for two slices
        assert plan(784, 1280, 5120)[0][1] != 320                                                  # 28 tiles x 2 slices would leave the chip empty
        assert plan(3136, 1280, 3840, mode=2, frames=14, hw=112)[0][1] != 320                      # temporal conv: never split
        assert plan(3136, 1280, 5120, ln_fold=1)[0][1] != 320                                      # a K slice would see part of a LayerNorm row
        assert lib.tt_gemm_set_big_tile(4) == 0                                                    # everything of round 4 but the split-K route
        assert plan(3136, 1280, 1280, **conv(8, 14))[0][:2] == [128, 128]
        assert plan(50176, 320, 1280)[0][:2] == [256, 320]
    finally:
        lib.tt_gemm_set_big_tile(1)
    assert plan(3136, 1280, 1280, **conv(8, 14))[0] == split(2)


def test_integration_md_stub_matches_the_header(lib):
    """INTEGRATION.md section 4 shows the ctypes stub a maintainer of the reference would paste.  It is evaluated here as it stands
    in the document: its TtAttnArgs must have the C compiler's sizeof(TtAttnArgs) and the tested binding's field names / offsets
    (a stub that stops short of the struct's last field hands the library a struct it reads out of bounds), and the ABI number the
    document quotes must be the library's."""
    import ctypes
    import subprocess
    import tempfile
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "class TtAttnArgs" in b]
    assert len(blocks) == 1, "INTEGRATION.md must hold exactly one TtAttnArgs stub"
    ns = {}
    cwd = os.getcwd()
    os.chdir(REPO)                                   # the stub loads the library by its path relative to the repository root
    try:
        exec(compile(blocks[0], "INTEGRATION.md:stub", "exec"), ns)
    finally:
        os.chdir(cwd)
    stub = ns["TtAttnArgs"]
    src = '#include <stdio.h>\n#include "ttvdm.h"\nint main(){printf("%zu\\n", sizeof(TtAttnArgs));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe])
        want = int(subprocess.check_output([exe]))
    assert ctypes.sizeof(stub) == want, (ctypes.sizeof(stub), want)
    layout = lambda s: [(n, getattr(s, n).offset, getattr(s, n).size) for n, _ in s._fields_]
    assert layout(stub) == layout(_lib.TtAttnArgs)
    quoted = re.findall(r"`tt_abi_version\(\)` = (\d+)", text)
    assert quoted and all(int(q) == lib.tt_abi_version() for q in quoted), (quoted, lib.tt_abi_version())
