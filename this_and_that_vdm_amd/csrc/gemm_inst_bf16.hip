// instantiation unit: every tt_gemm tile configuration for one storage type (see gemm_kernel.h)
#include "gemm_kernel.h"
namespace ttg {
void launch_bf16(GemmP& p, int cfg, hipStream_t st) { launch<bf16_tag>(p, cfg, st); }
void launch_sq320_bf16(const GemmP& p, hipStream_t st) {
  if (p.rowvec) { launch_sq320<bf16_tag, true, true>(p, st); return; }          // (a row vector only rides on the residual form: the output projections)
  if (p.residual) launch_sq320<bf16_tag, true>(p, st); else launch_sq320<bf16_tag, false>(p, st);
}
}
#ifdef TT_GEMM_TIMELINE
extern "C" int tt_debug_timeline(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ttg::g_tl), n * 8); }
#endif
