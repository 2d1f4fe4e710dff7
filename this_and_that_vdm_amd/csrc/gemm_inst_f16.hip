// instantiation unit: every tt_gemm tile configuration for one storage type (see gemm_kernel.h)
#include "gemm_kernel.h"
namespace ttg {
void launch_f16(GemmP& p, int cfg, hipStream_t st) { launch<f16_tag>(p, cfg, st); }
void launch_sq320_f16(const GemmP& p, hipStream_t st) {
  if (p.rowvec) { launch_sq320<f16_tag, true, true>(p, st); return; }          // (a row vector only rides on the residual form: the output projections)
  if (p.residual) launch_sq320<f16_tag, true>(p, st); else launch_sq320<f16_tag, false>(p, st);
}
}
