// libttvdm: version / error plumbing (host only).
#include <stdarg.h>
#include <stdio.h>
#include "ttvdm.h"

static thread_local char g_err[512] = "";

void tt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" int tt_abi_version(void) { return 11; }   // 2: TtGemmArgs.ln_fold / ln_eps, TT_F32; 3: out_fp8, TtAttnArgs.fp8, tt_add_rowvec, tt_conv3x3; 4: tt_groupnorm_small; 5: tt_softmax_rows; 6: TtAttnArgs fused query projection (qx, wq, bq, qc, ln_eps), tt_gemm_set_big_tile; 7: TtGemmArgs.rowvec_mod; 8: TtGemmArgs.stats_out / stats_seg, tt_gemm_stats_rows, tt_groupnorm_tiles; 9: TtGemmArgs.gn_out (GroupNorm in the split-K reduction), tt_gemm_gn_fused; 10: TtAttnArgs.v_rows (row-major V); 11: tt_gemm_set_f32_split (split-fp16 products in TT_F32)
extern "C" const char* tt_target_arch(void) { return "gfx950"; }
extern "C" const char* tt_last_error(void) { return g_err; }
