"""MI355X-native SVD denoise path of This&That (UNet + GestureNet ControlNet + Euler loop) behind the reference's
model / pipeline API.  Compute = libttvdm.so (hand-written HIP for gfx950, include/ttvdm.h); there is no CPU fallback."""
__version__ = "0.1.0"
