"""CPU: host-side logic added in round 6 (no kernels): the pre-split weight format of the split16 mode and the write ledger that guards the
producer -> GroupNorm hand-off."""
import pytest
import torch


def test_presplit_f32_pairs_reconstruct_the_values():
    """packing.presplit_f32: per aligned group of 4 k-values the eight fp16 values h0..h3 l0..l3 with x = 2^8 h + 2^-3 l up to
    max(2^-22 |x|, 2^-28); same shape / dtype / strides as the fp32 matrix; row slices and row permutations stay valid and keep the marker."""
    from this_and_that_vdm_amd.packing import PreSplitF32, presplit_f32
    g = torch.Generator().manual_seed(0)
    for scale in (1.0, 0.02, 1e-4, 3e4, 1e6):
        w = torch.randn(24, 64, generator=g) * scale
        p = presplit_f32(w)
        assert isinstance(p, PreSplitF32) and p.shape == w.shape and p.dtype == torch.float32 and p.stride() == w.stride()
        raw = p.as_subclass(torch.Tensor).view(torch.float16).view(24, 16, 2, 4)
        h, l = raw[:, :, 0].reshape(24, 64).double(), raw[:, :, 1].reshape(24, 64).double()
        err = (h * 256 + l / 8 - w.double()).abs()
        assert bool((err <= (w.double().abs() * 2.0 ** -21).clamp_min(2.0 ** -27)).all()), (scale, float(err.max()))
        assert bool(torch.isfinite(h).all() and torch.isfinite(l).all())
        assert isinstance(p[4:12], PreSplitF32) and torch.equal(p[4:12].as_subclass(torch.Tensor), presplit_f32(w[4:12]).as_subclass(torch.Tensor))
        perm = torch.randperm(24, generator=g)
        assert torch.equal(p[perm].as_subclass(torch.Tensor), presplit_f32(w[perm]).as_subclass(torch.Tensor))
    assert presplit_f32(p) is p
    with pytest.raises(ValueError):
        presplit_f32(torch.zeros(4, 6))                       # K % 4
    with pytest.raises(ValueError):
        presplit_f32(torch.zeros(4, 8, dtype=torch.float16))


def test_groupnorm_handoff_is_invalidated_by_any_write_to_the_storage():
    """ops._wrote / ops._handoff_valid: the sums a producer attached stay usable only while no ops.* launch has written the tensor's
    storage -- through the tensor itself, a view object of it, or a slice -- and while pointer and shape are the ones recorded."""
    from this_and_that_vdm_amd import ops
    x = torch.zeros(8, 16)
    serial = ops._wrote(x)
    x._tt_stats = ("sums", 4, serial, x.data_ptr(), tuple(x.shape))
    assert ops._handoff_valid(x, *x._tt_stats[2:])
    other = torch.zeros(8, 16)
    ops._wrote(other)                                        # another storage: no effect
    assert ops._handoff_valid(x, *x._tt_stats[2:])
    view = x.view(8, 16)                                      # a different Python object over the same storage
    ops._wrote(view)
    assert hasattr(x, "_tt_stats") and not ops._handoff_valid(x, *x._tt_stats[2:])      # the attribute survives on the base object, the check fails
    serial = ops._wrote(x)                                    # a write through the tensor itself drops the attribute
    assert not hasattr(x, "_tt_stats")
    x._tt_stats = ("sums", 4, serial, x.data_ptr(), tuple(x.shape))
    ops._wrote(x[2:4])                                        # a row slice
    assert not ops._handoff_valid(x, *x._tt_stats[2:])
    serial = ops._wrote(x)
    assert not ops._handoff_valid(x[1:], serial, x.data_ptr(), tuple(x.shape))          # same storage, other pointer / shape


def test_f32_split_switch_is_off_by_default_and_in_the_keys():
    import os
    from this_and_that_vdm_amd import ops
    assert ops.f32_split() == (os.environ.get("TT_F32_SPLIT", "0") not in ("", "0"))
    import inspect
    from this_and_that_vdm_amd.svd import denoise, denoiser_base
    assert "ops.f32_split()" in inspect.getsource(denoise.DenoiseLoop.begin)            # part of the graph key
    assert "ops.f32_split()" in inspect.getsource(denoiser_base.DenoiserBase._pack_key)  # ... and of the pack key (toggling repacks)
