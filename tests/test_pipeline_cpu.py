"""CPU: pipeline pieces either side of the denoise loop that need no GPU (decode_latents), against
hand-computed expectations with the stub VAE."""
import pytest
import torch

from tests.stubs import StubCLIPVision, StubVAE
from this_and_that_vdm_amd.svd import StableVideoDiffusionControlNetPipeline, UNetSpatioTemporalConditionModel

KW = dict(block_out_channels=(64, 64, 64, 64), num_attention_heads=(1, 1, 1, 1), cross_attention_dim=32, num_frames=3)


class RecordingVAE(StubVAE):
    """Stub VAE whose decode takes ``num_frames`` (as AutoencoderKLTemporalDecoder.decode does) and records its calls."""

    def __init__(self):
        super().__init__()
        self.calls = []

    def decode(self, z, num_frames=None):
        self.calls.append((tuple(z.shape), num_frames))
        return super().decode(z, num_frames)


@pytest.mark.parametrize("frames,chunk", [(14, 8), (14, 14), (5, 2), (3, 8)])
def test_decode_latents_chunks_like_the_reference(frames, chunk):
    """reference :257-283: latents / scaling_factor, decoded ``decode_chunk_size`` frames at a time with
    num_frames = frames in THIS chunk, concatenated, reshaped to [B, C, F, H, W], float32."""
    vae = RecordingVAE()
    pipe = StableVideoDiffusionControlNetPipeline.from_pretrained(None, vae=vae, image_encoder=StubCLIPVision(),
                                                                  unet=UNetSpatioTemporalConditionModel(**KW))
    lat = torch.randn(1, frames, 4, 4, 6, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        out = pipe.decode_latents(lat, frames, decode_chunk_size=chunk)
        flat = lat.flatten(0, 1) / vae.config.scaling_factor
        want = torch.cat([StubVAE.decode(vae, flat[i:i + chunk]).sample for i in range(0, frames, chunk)], 0)
    want = want.reshape(1, frames, 3, 32, 48).permute(0, 2, 1, 3, 4).float()
    assert out.shape == (1, 3, frames, 32, 48) and out.dtype == torch.float32
    torch.testing.assert_close(out, want, rtol=0, atol=0)
    sizes = [min(chunk, frames - i) for i in range(0, frames, chunk)]
    assert [c[0][0] for c in vae.calls] == sizes and [c[1] for c in vae.calls] == sizes      # num_frames = chunk length


def test_decode_latents_without_num_frames_kwarg():
    """a VAE whose forward() has no ``num_frames`` parameter (plain AutoencoderKL) is called without it."""

    class PlainVAE(StubVAE):
        def forward(self, x):
            return self.decode(self.encode(x).latent_dist.mode())

        def decode(self, z, **kw):
            assert not kw, kw
            return super().decode(z)

    pipe = StableVideoDiffusionControlNetPipeline.from_pretrained(None, vae=PlainVAE(), image_encoder=StubCLIPVision(),
                                                                  unet=UNetSpatioTemporalConditionModel(**KW))
    with torch.no_grad():
        out = pipe.decode_latents(torch.zeros(1, 3, 4, 2, 2), 3, decode_chunk_size=2)
    assert out.shape == (1, 3, 3, 16, 16)
