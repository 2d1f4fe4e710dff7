// instantiation unit: the TT_F32 tile configurations (see gemm_kernel.h)
#include "gemm_kernel.h"
namespace ttg {
// TT_F32 (reference-precision mode): two tile shapes of the same kernel template.  BK counts elements, so 32 fp32
// elements give the 128-byte tile rows of the 16-bit BK = 64 configurations; no split-K (one summation order).
template <>
inline void launch<f32_tag>(GemmP& p, int cfg, hipStream_t st) {
  if (cfg == 0) launch_cfg<f32_tag, 128, 128, 32, 2, 2, 2, true>(p, st);
  else launch_cfg<f32_tag, 64, 64, 32, 4, 2, 2, true>(p, st);
}

void launch_f32(GemmP& p, int cfg, hipStream_t st) { launch<f32_tag>(p, cfg, st); }
}
