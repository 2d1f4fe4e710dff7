"""GPU: the native temporal VAE decoder (decode_latents' callee, reference svd/pipeline_stable_video_diffusion_controlnet.py:257-283)
against the oracle restatement (oracle/vae.py) on identical weights: TT_F32 at the north-star tolerance, 16-bit modes by relative
L2; row softmax kernel vs torch; the pipeline's chunked decode_latents through the native decoder."""
import pytest
import torch

from tests.parity_common import assert_north_star, err_stats

pytestmark = pytest.mark.gpu
CFG = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=2)


REAL = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2)        # the shipped SVD temporal VAE decoder


def _pair(dtype, cfg=None):
    from oracle import vae as ov
    from this_and_that_vdm_amd.svd.autoencoder_kl_temporal_decoder import AutoencoderKLTemporalDecoder
    from this_and_that_vdm_amd.utils.synthetic import fill_parameters_
    cfg = CFG if cfg is None else cfg
    o = ov.AutoencoderKLTemporalDecoder(**cfg).eval()
    fill_parameters_(o, "vae.", round_to=dtype)
    p = AutoencoderKLTemporalDecoder(**cfg).eval()
    p.load_state_dict(o.state_dict())
    p = p.to(device="cuda:0", dtype=dtype)
    if dtype == torch.float32:
        p.compute_dtype = torch.float32
    return p, o


@pytest.mark.parametrize("dtype,rel", [(torch.float32, 1e-4), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@torch.no_grad()
def test_decoder_matches_oracle(dtype, rel):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    p, o = _pair(dtype)
    g = torch.Generator().manual_seed(3)
    z = (torch.randn(6, 4, 8, 4, generator=g) * 3.0).to(dtype).float()         # 2 videos x 3 frames, exactly representable
    ref = o.decode(z, num_frames=3)
    got = p.decode(z.cuda(), num_frames=3).sample
    assert got.shape == ref.shape == (6, 3, 64, 32) and got.dtype == torch.float32
    st = err_stats(got, ref)
    print(f"temporal VAE decoder {dtype} vs fp32 oracle: {st}")
    if dtype == torch.float32:
        assert_north_star(got, ref, "temporal VAE decoder (TT_F32)")
    assert st["rel_l2"] <= rel and st["cos"] >= 0.9995, st
    # frames of different videos do not mix: decoding the second video alone gives the same frames (fp32 mode: same launches per row)
    if dtype == torch.float32:
        alone = p.decode(z[3:].cuda(), num_frames=3).sample
        assert_north_star(alone, ref[3:], "second video decoded alone")


@pytest.mark.parametrize("dtype,rel", [(torch.float32, 1e-4), (torch.float16, 4.6e-3), (torch.bfloat16, 3.7e-2)])
@torch.no_grad()
def test_decoder_at_the_shipped_widths_matches_oracle(dtype, rel):
    """The decoder as SVD ships it -- block_out_channels (128, 256, 512, 512): the mid-block attention is ONE head of d = 512
    over L = h*w tokens (here 16x28 = 448: the per-frame score / softmax / PV loop with an fp32 score buffer), GroupNorm over
    128..512 channels up to 128x224 pixels, conv_out + the 3-tap frame conv -- on a small spatial case (2 frames of 16x28
    latents -> 128x224 pixels) against oracle/vae.py on identical weights.  TT_F32: every element inside rtol 1e-3 / atol 1e-4;
    16-bit storage: relative L2 (limits = 2 x the values measured on MI355X: fp16 2.27e-3, bf16 1.84e-2; TT_F32 measured 5.3e-6),
    cosine >= 0.999."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))
    try:
        p, o = _pair(dtype, REAL)
        g = torch.Generator().manual_seed(11)
        z = (torch.randn(2, 4, 16, 28, generator=g) * 3.0).to(dtype).float()
        ref = o.decode(z, num_frames=2)
        got = p.decode(z.cuda(), num_frames=2).sample
        assert got.shape == ref.shape == (2, 3, 128, 224) and got.dtype == torch.float32
        st = err_stats(got, ref)
        print(f"temporal VAE decoder at the shipped widths, {dtype} vs fp32 oracle: {st}")
        if dtype == torch.float32:
            assert_north_star(got, ref, "temporal VAE decoder at (128, 256, 512, 512), TT_F32")
        assert st["rel_l2"] <= rel and st["cos"] >= 0.999, st
    finally:
        torch.set_num_threads(threads)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_softmax_rows(dtype):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from this_and_that_vdm_amd import ops
    g = torch.Generator().manual_seed(0)
    for rows, cols in ((37, 1792), (5, 36), (130, 300)):
        ld = (cols + 7) // 8 * 8
        x = torch.full((rows, ld), float("nan"))
        x[:, :cols] = torch.randn(rows, cols, generator=g) * 4
        y = ops.softmax_rows(x.cuda(), dtype, cols=cols)
        assert y.shape == (rows, ld) and y.dtype == dtype
        ref = torch.softmax(x[:, :cols], dim=1)
        tol = {torch.float32: 1e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
        torch.testing.assert_close(y[:, :cols].float().cpu(), ref, rtol=tol, atol=tol)
        assert not y[:, cols:].float().any()                     # padding columns are zeros (K padding of the following GEMM)


@torch.no_grad()
def test_pipeline_decode_latents_runs_the_native_decoder_in_chunks():
    """decode_latents (reference :257-283): latents / scaling_factor, chunks of `decode_chunk_size` frames with num_frames = chunk
    length, concatenated and returned as fp32 [B, C, F, H, W] -- through the native decoder, against the oracle doing the same."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from this_and_that_vdm_amd.svd.pipeline_stable_video_diffusion_controlnet import _SVDPipelineCore
    p, o = _pair(torch.float32)
    pipe = _SVDPipelineCore.__new__(_SVDPipelineCore)
    pipe.vae = p
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 5, 4, 8, 4, generator=g)
    got = pipe.decode_latents(lat.cuda(), num_frames=5, decode_chunk_size=3)
    z = lat.flatten(0, 1) / o.scaling_factor
    ref = torch.cat([o.decode(z[:3], num_frames=3), o.decode(z[3:], num_frames=2)], 0)
    ref = ref.reshape(1, 5, *ref.shape[1:]).permute(0, 2, 1, 3, 4)
    assert got.shape == ref.shape == (1, 3, 5, 64, 32) and got.dtype == torch.float32
    assert_north_star(got, ref, "decode_latents through the native decoder")
