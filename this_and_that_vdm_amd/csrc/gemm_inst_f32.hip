// instantiation unit: every tt_gemm tile configuration for one storage type (see gemm_kernel.h)
#include "gemm_kernel.h"
namespace ttg {
void launch_f32(GemmP& p, int cfg, hipStream_t st) { launch<f32_tag>(p, cfg, st); }
}
