// tt_gemm: gather-GEMM on gfx950 MFMA (v_mfma_f32_32x32x16_{bf16,f16}).
//
//   out[m][n] = epilogue( sum_{tap,src,c} A_src[rowmap(m,tap)][c] * W[n][(tap,src,c)] )
//
// One kernel family covers nn.Linear, 1x1 conv, 3x3 conv (stride 1/2, fused nearest-x2 upsample),
// and the (3,1,1) temporal conv, with up to two channel sources (skip-concat without a concat
// buffer).  Structure:
//   * BM x BN x BK tiles, WGM x WGN waves (4 or 8), each wave owns (BM/WGM) x (BN/WGN) as 32x32 MFMA fragments;
//   * A and W tiles go HBM -> LDS by 16-byte LDS-DMA (global_load_lds) into an NST-deep ring: tile kt+NST-1 is
//     issued while tile kt is computed, and the wait before tile kt is a COUNTED s_waitcnt vmcnt(G*(NST-2)), so
//     NST-2 tiles stay in flight across the (raw) s_barrier -- one barrier per K step, no vmcnt(0) in the loop
//     (cdna guide T3+T4).  Out-of-range rows / conv halo / K tails point their lane at a zero page;
//   * LDS image is lane-linear with the XOR swizzle applied on the SOURCE chunk and on the read
//     (conflict-free ds_read_b128, see common.h);
//   * MFMA operands are swapped (W fragment as the MFMA "A" operand) so each lane ends up with 4
//     consecutive output columns of one row: 8-byte epilogue loads/stores.
#include <stdlib.h>
#include "gemm_kernel.h"

namespace ttg {
// per-dtype instantiation units (gemm_inst_*.hip)
void launch_bf16(GemmP& p, int cfg, hipStream_t st);
void launch_f16(GemmP& p, int cfg, hipStream_t st);
void launch_f32(GemmP& p, int cfg, hipStream_t st);
void launch_sq320_bf16(const GemmP& p, hipStream_t st);
void launch_sq320_f16(const GemmP& p, hipStream_t st);
void launch_pp_bf16(GemmP& p, hipStream_t st);         // gemm_pp.hip: persistent 256 x 256 x 64 ping-pong kernel
void launch_pp_f16(GemmP& p, hipStream_t st);
void launch_w320_bf16(GemmP& p, hipStream_t st);       // gemm_w320.hip: 256 x 320 x 64 tiles for N = 320 t at the finest UNet level
void launch_w320_f16(GemmP& p, hipStream_t st);
void launch_w320h_bf16(GemmP& p, hipStream_t st);      // ... its 128-row variant (two K halves per slab) for problems with fewer rows
void launch_w320h_f16(GemmP& p, hipStream_t st);
}
using namespace ttg;

namespace {
struct TileCfg { int bm, bn, bk, nst, wgm, wgn; };
constexpr TileCfg kCfgs[] = {        // keep in step with launch<Tag>() in gemm_kernel.h
  {128, 128, 64, 2, 2, 2},   // 0: 4 waves, 64 KiB, 2 blocks/CU
  {128,  64, 64, 3, 2, 2},   // 1: 72 KiB
  { 64,  64, 64, 4, 2, 2},   // 2: small problems, deep ring (64 KiB)
  {256, 128, 32, 3, 4, 2},   // 3: 8 waves, 72 KiB -> 2 blocks/CU
  {256, 256, 32, 3, 2, 4},   // 4: 8 waves, 96 KiB
  {128, 128, 32, 3, 2, 2},   // 5: 48 KiB -> 3 blocks/CU
  {256, 128, 64, 3, 4, 2},   // 6: 8 waves, 144 KiB
  {128, 160, 64, 2, 4, 1},   // 7: N = 320 without tile waste, wave tile 32x160, 72 KiB
  {128, 320, 32, 3, 4, 2},   // 8: 8 waves, wave tile 32x160, 84 KiB
  {256, 256, 64, 2, 2, 4},   // 9: 8 waves, wave tile 128x64, 128 KiB, plain double buffer
  {128, 128, 64, 4, 2, 2},   // 10: 128 KiB, 1 block/CU, prefetch distance 3
  {128, 128, 64, 2, 4, 2},   // 11: 8 waves (wave tile 32x64), 64 KiB -> 16 waves/CU
  {256, 160, 32, 3, 8, 1},   // 12: N = 320/960, 8 waves (wave tile 32x160), 78 KiB -> 2 blocks/CU
  {256, 128, 32, 4, 4, 2},   // 13: like 3 with one more stage (96 KiB, 1 block/CU)
  {128, 128, 32, 5, 4, 2},   // 14: 8 waves, 80 KiB, prefetch distance 4 (x32) -> 2 blocks/CU
  {128, 128, 64, 3, 4, 2},   // 15: 8 waves, 96 KiB, prefetch distance 2 -> 1 block/CU
  {128, 128, 64, 4, 4, 2},   // 16: 8 waves, 128 KiB, prefetch distance 3 -> 1 block/CU
  {256, 256, 32, 4, 2, 4},   // 17: 8 waves, wave tile 128x64, 128 KiB: three 32 KiB tiles in flight
  {256, 128, 64, 2, 4, 2},   // 18: 8 waves, wave tile 64x64, 96 KiB, plain double buffer
  {256, 128, 32, 5, 4, 2},   // 19: 8 waves, wave tile 64x64, 120 KiB: four 24 KiB tiles in flight
  {128, 128, 128, 2, 4, 2},  // 20: 8 waves, 128-deep K steps (256-byte rows), 128 KiB double buffer: half the barriers per MFMA (opt-in, see make_plan)
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

int g_forced_cfg = -2;
int forced_cfg() {
  if (g_forced_cfg == -2) { const char* e = getenv("TT_GEMM_CFG"); g_forced_cfg = e ? atoi(e) : -1; }
  return g_forced_cfg;
}

// tuning switch: TT_GEMM_GROUP_M=<rows> overrides the group height of the tile order (gemm_kernel.h: launch_cfg picks it otherwise)
int g_group_m = -2;
int group_m_override() {
  if (g_group_m == -2) { const char* e = getenv("TT_GEMM_GROUP_M"); g_group_m = e ? atoi(e) : 0; }
  return g_group_m;
}

int g_forced_split = -2;
int forced_split() {
  if (g_forced_split == -2) { const char* e = getenv("TT_GEMM_SPLITK"); g_forced_split = e ? atoi(e) : -1; }
  return g_forced_split;
}

// Tile + split-K plan, tuned on MI355X with tools/gemm_bench.py (shapes of the SVD UNet at 256x448 and 512x896):
//   * N = 320 / 960 (multiple of 160 but not of 128): 128x160 tiles, no column waste;
//   * tall problems with wide N: 256x128, 8 waves;
//   * otherwise the largest of 128x128 / 128x64 / 64x64 that still yields ~1.5 workgroups per CU;
//   * few tiles but a long K (convs at the two coarsest levels): 128x128 tiles with the K loop split over S
//     workgroups, fp32 slabs reduced in fixed order by a second kernel.
struct Plan { int cfg, splitk; };

// `wide_ok`: the 256x128 tile pays for tall-and-wide problems (GEGLU projections, fused QKV) unless the epilogue reads
// per-row tensors (residual / blend / row vector): its 128-VGPR budget has no room to preload them, so they would be read
// between the stores (measured: 16 us epilogue instead of 3)
Plan make_plan(int m, int n, long ktot, bool allow_split, bool wide_ok = true, bool mode0 = false, bool k128 = false) {
  Plan pl{0, 1};
  const int f = forced_cfg();
  const long b128 = (long)ceil_div(m, 128) * ceil_div(n, 128);
  const long b12864 = (long)ceil_div(m, 128) * ceil_div(n, 64);
  if (f >= 0 && f < kNumCfgs) pl.cfg = f;
  else if (n % 160 == 0 && n % 128 != 0 && n <= 960 && m >= 2048) pl.cfg = 7;
  // (256x256 tiles -- cfg 9 -- win 6..25 % on the wide GEGLU / QKV projections in isolation (tools/gemm_cold.py) but lose
  // inside the two-branch step graph, where one 128 KiB block per CU shuts the other branch's kernels out: FF1 at 32x56
  // 144 vs 137 us, QKV at 16x28 60 vs 48 us per launch -- profiles/r2_*_kernel_by_grid.txt history.  Not routed.)
  else if (m >= 8192 && n >= 1024 && wide_ok) pl.cfg = 3;
  else if (m < 2048 && n >= 2560 && n % 160 == 0) pl.cfg = 7;
  else if (b128 >= 384) pl.cfg = 11;                 // 128x128 with 8 waves (32x64 wave tiles): 16 waves per CU
  else if (b12864 >= 384) pl.cfg = 1;
  else pl.cfg = 2;
  const int fs = forced_split();
  if (fs >= 1) { pl.splitk = allow_split ? fs : 1; return pl; }
  const long kt = (ktot + 63) / 64;
  static int deep = -1;
  if (deep < 0) { const char* e = getenv("TT_GEMM_DEEP"); deep = e ? atoi(e) : 1; }
  if (deep && f < 0 && b128 <= 256 && ktot > 0) {
    // at most one 128x128 tile per CU: LDS is free for a 4-deep ring (prefetch distance 3), which beats two resident
    // blocks with a double buffer once the loads really overlap the MFMAs; K is split only as far as whole CUs are idle.
    // (128-deep K steps for channel counts that are multiples of 128 -- cfg 20, one barrier pair per 16 MFMAs of a wave -- are faster
    // in isolation: FF2 at 3136 rows 690 -> 791 TFLOP/s, K = 1280 509 -> 556, the 3x3 conv 792 -> 848 (tools/gemm_bench.py 16 20), and
    // slower in the step: 31.19 / 31.23 -> 31.48 / 31.51 ms, restricted to K >= 4096 30.83 / 31.08 -> 31.26 / 31.17: a double buffer
    // has one K step of prefetch where the 4-deep ring has three.  Opt-in: TT_GEMM_DEEP=2.)
    // Short-K linears that leave half of the CUs without a 128 x 128 tile (the half-row output projections of the cross-attentions at the
    // third level, everything K = C at the coarsest one) on 64 x 64 tiles, two blocks per CU, all resident at once: faster on warm operands
    // (tools/small_gemm_sweep.py, us in a graph of ten launches: 1568 x 1280 x 1280 16.8 -> 12.3, 784 rows 16.6 -> 11.3, 392 rows 13.7 -> 10.0),
    // NOT in the step, where every weight comes from HBM once and 64-row tiles fetch it twice as often: 29.06 / 29.18 -> 29.10 / 29.10 ms
    // (one call, interleaved).  Opt-in: TT_GEMM_SMALL64=1.  Default (2): only wide problems (n >= 4096) of at most 64 rows -- the 64-row tail of the third level's
    // GEGLU projection (3136 = 12 x 256 + 64 rows; 21 launches per step), half of whose 128-row tiles was padding: 29.78 / 29.76 -> 29.70 / 29.67 ms.
    static int small64 = -1;
    if (small64 < 0) { const char* e = getenv("TT_GEMM_SMALL64"); small64 = e ? atoi(e) : 2; }
    if (small64 && (small64 != 2 || (m <= 64 && n >= 4096)) && mode0 && ktot <= 2048 && (long)ceil_div(m, 64) * ceil_div(n, 64) <= 512) { pl.cfg = 2; return pl; }   // (2: only the <= 64-row tails)
    // (Round 6, tools/coarse_conv_probe.py / profiles/r6_coarse_conv_probe.txt: on cold weights in isolation the 70-tile convs of the coarsest
    // level are 18-20 % faster as two resident double-buffered 8-wave workgroups per CU with six K slices -- cfg 11: 53.6 -> 43.9 us,
    // 92.7 -> 74.2 us, reduction pass included -- and the step does not move: 28.97 / 29.06 vs 29.05 / 29.00 ms, one call, interleaved.
    // Not routed: once more, an isolated sweep does not predict the step.)
    pl.cfg = (deep == 2 && k128) ? 20 : 16;
    long s = allow_split ? 256 / b128 : 1;
    if (s > kt / 8) s = kt / 8;
    if (s > 16) s = 16;
    pl.splitk = s >= 2 ? (int)s : 1;
    return pl;
  }
  if (allow_split && f < 0 && b128 < 384 && kt >= 40) {
    long s = (512 + b128 - 1) / b128;
    if (s > kt / 8) s = kt / 8;
    if (s > 16) s = 16;
    if (s >= 2) { pl.cfg = 11; pl.splitk = (int)s; }
  }
  return pl;
}

constexpr TileCfg kCfgsF32[] = {{128, 128, 32, 2, 2, 2}, {64, 64, 32, 4, 2, 2}};
int plan_f32(int m, int n, int min_tiles = 256) { return (long)ceil_div(m, 128) * ceil_div(n, 128) >= min_tiles ? 0 : 1; }

}  // namespace


extern "C" int tt_gemm_set_tile_override(int32_t cfg) {
  if (cfg < -1 || cfg >= kNumCfgs) TT_FAIL(TT_EINVAL, "tt_gemm_set_tile_override: cfg %d (valid -1..%d)", cfg, kNumCfgs - 1);
  g_forced_cfg = cfg;
  return TT_OK;
}

// a row vector the streaming kernel can carry: at most two distinct rows over the launch (both preloaded), on the residual form
static bool sq320_rowvec_ok(const TtGemmArgs* a) {
  if (!a->rowvec) return true;
  static int rv = -1;
  if (rv < 0) { const char* e = getenv("TT_SQ320_ROWVEC"); rv = e ? atoi(e) : 1; }
  if (!rv || !a->residual) return false;
  if (a->rowvec_mod == 2 && a->rowvec_rows == 1) return true;
  return a->rowvec_mod == 0 && a->rowvec_rows >= 32 && a->rowvec_rows % 32 == 0 && (long)a->rowvec_rows * 2 >= a->m;
}
static int g_sq320 = -1;
extern "C" int tt_gemm_set_streaming_square(int32_t on) {
  g_sq320 = on == 2 ? 2 : (on ? 1 : 0);          // 2: by size (the default)
  return TT_OK;
}

bool sq320_ok(const TtGemmArgs* a) {
  // 0 off, 1 on, 2 (default) by size: at 64x112 latents (200 704 rows: 784 big tiles = 3.06 rounds of 196) the 32-row streaming kernel wins
  // (block l0hi 5.46 -> 5.39 ms, fp8 4.995 -> 4.91; 512x896 step 109.6 -> 109.1 ms), at 32x56 it loses (29.04 -> 29.32 ms): one call each, interleaved
  if (g_sq320 < 0) { const char* e = getenv("TT_GEMM_SQ320"); g_sq320 = e ? atoi(e) : 2; }
  static int min_rows = -1;
  if (min_rows < 0) { const char* e = getenv("TT_SQ320_MIN_ROWS"); min_rows = e ? atoi(e) : 131072; }
  if (g_sq320 == 2 && a->m < min_rows) return false;
  return g_sq320 && a->dtype != TT_F32 && !a->ln_fold && !a->out_fp8 && forced_cfg() < 0 && a->mode == 0 && a->k1 == 0 && a->k0 == SQ_K && a->n == SQ_N && a->m >= 4096 &&
         !a->geglu && sq320_rowvec_ok(a) && !a->out_f32 && !a->out_col_hw &&
         (!a->blend || (a->blend == a->residual && a->ld_blend == a->ld_res));
}
// The persistent big-tile kernel (gemm_pp.hip) takes tall-and-wide Linear problems without per-row epilogue operands when every CU
// gets about two or more 256 x 256 tiles and the last round of tiles is nearly full: the GEGLU projections at the three finest UNet
// levels (1960 / 980 / 480 tiles).  A problem whose row count is not a multiple of 256 and that only qualifies without its
// ragged last tile row is launched in two parts -- whole tile rows on the persistent kernel, the remaining < 256 rows on the tiled
// kernel (3136 rows: 480 tiles = 1.9 rounds + 64 rows, instead of 520 tiles = 3 rounds; the persistent kernel has no short cut for
// a ragged tile -- its out-of-range rows are zero-filled by the DMA bounds check and multiplied like any others, so it costs a
// whole tile time -- and cutting tiles along K stream-K fashion costs more in partial-tile traffic than it saves: DESIGN.md
// section 6.0).  TT_GEMM_PP=0 keeps everything on the tiled kernels (A/B).
// The constants below (256 CUs, one workgroup per CU) are MI355X in SPX mode: the only target of this library (gfx950, tt_target_arch).
static int g_pp = -1;
bool pp_ok(const TtGemmArgs* a) {
  if (g_pp < 0) { const char* e = getenv("TT_GEMM_PP"); g_pp = e ? atoi(e) : 1; }
  if (!g_pp || forced_cfg() >= 0 || a->dtype == TT_F32 || a->mode != 0 || a->k1 != 0 || a->ln_fold > 1 || a->out_fp8 || a->residual || a->blend ||
      a->rowvec || a->out_f32 || a->out_col_hw || (a->k0 & 63) || a->k0 < 128 || (a->n & 15) || (a->ldo & 7))
    return false;
  const long tiles = (long)ceil_div(a->m, 256) * ceil_div(a->n, 256);
  const long rounds = (tiles + 255) / 256;
  // The GEGLU projections have their own, lower bar (TT_PP_GEGLU_MIN_TILES / _MIN_FILL, A/B): their alternative is the tiled template at
  // 640-670 TFLOP/s, which a persistent launch beats even with a 75 %-full last round -- 256x384: 10752 x 5120 (840 tiles, 82 %) and
  // 2688 x 10240 (400 whole tiles, 78 %): --res ref 27.72 -> 27.06 ms with 400 / 75 against 460 / 90 (one call, interleaved).  256x448 and
  // 512x896 do not change: their projections pass either bar or fall below both.
  static int g_min_tiles = -1, g_min_fill = -1;
  if (g_min_tiles < 0) { const char* e = getenv("TT_PP_GEGLU_MIN_TILES"); g_min_tiles = e ? atoi(e) : 400; }
  if (g_min_fill < 0) { const char* e = getenv("TT_PP_GEGLU_MIN_FILL"); g_min_fill = e ? atoi(e) : 75; }
  if (a->geglu) return tiles >= g_min_tiles && tiles * 100 >= rounds * 256 * g_min_fill;
  static int p_min_tiles = -1, p_min_fill = -1;              // (TT_PP_MIN_TILES / TT_PP_MIN_FILL, A/B)
  if (p_min_tiles < 0) { const char* e = getenv("TT_PP_MIN_TILES"); p_min_tiles = e ? atoi(e) : 460; }
  if (p_min_fill < 0) { const char* e = getenv("TT_PP_MIN_FILL"); p_min_fill = e ? atoi(e) : 90; }
  return tiles >= p_min_tiles && tiles * 100 >= rounds * 256 * p_min_fill;      // ~2 rounds of tiles per CU or more, last round >= 90 % full on average
}
// The 256 x 320 big-tile kernel (gemm_w320.hip) takes 16-bit problems whose output width is a multiple of 320 and whose row count
// fills most of a round of 256 CUs with 256-row tiles (the finest UNet level: 50176 rows = 196 tiles): Linear (one or two sources,
// optional LayerNorm fold of the A rows), conv3x3 stride 1 and the temporal conv, with bias / scale / row vector (groups of >= 32
// rows) / residual / AlphaBlender epilogues.  TT_GEMM_W320=0 keeps them on the tiled kernels (A/B).
// TT_F32 products: 0 = exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak), 1 = "split16": every fp32 operand split on the fly into
// fp16 hi + lo, three 16-bit MFMAs per product block (gemm_kernel.h, `mma`): ~2^-21 relative per product at 3 / 16 of the issue time.
static int g_f32_split = -1;
extern "C" int tt_gemm_set_f32_split(int32_t on) { g_f32_split = on ? 1 : 0; return TT_OK; }
static int f32_split() {
  if (g_f32_split < 0) { const char* e = getenv("TT_F32_SPLIT"); g_f32_split = e && atoi(e) ? 1 : 0; }
  return g_f32_split;
}
int tt_internal_f32_split() { return f32_split(); }
static int g_w320 = -1;
extern "C" int tt_gemm_set_big_tile(int32_t on) {
  // internal: bit 0 the 256 x 320 kernel, bit 1 its 128 x 320 variant, bit 2: that one for conv3x3 only, bit 3: its split-K route
  g_w320 = on == 0 ? 0 : (on == 2 ? 1 : (on == 3 ? 11 : (on == 4 ? 7 : 15)));
  return TT_OK;
}
static void w320_init() {
  if (g_w320 < 0) { const char* e = getenv("TT_GEMM_W320"); tt_gemm_set_big_tile(e ? atoi(e) : 1); }
}
// what both kernels of gemm_w320.hip need of a problem
static bool w320_eligible(const TtGemmArgs* a) {
  w320_init();
  if (!g_w320 || forced_cfg() >= 0 || a->dtype == TT_F32 || a->n % 320 || (a->k0 & 63) || (a->k1 & 63) || (a->mode == 1 ? 9 : a->mode == 2 ? 3 : 1) * (a->k0 + a->k1) < 128 || a->geglu ||
      a->out_fp8 || a->out_f32 || a->out_col_hw || a->ln_fold > 1 || (a->ln_fold && (a->mode != 0 || a->k1)))
    return false;
  if (a->mode == 1 && (a->stride != 1 || a->upsample || a->hin != a->hout || a->win != a->wout || a->win >= 32768 || a->hin >= 32768)) return false;
  if (a->rowvec && a->rowvec_rows < 32 && !(a->rowvec_rows == 1 && a->rowvec_mod == 2)) return false;
  // 8-byte (16-bit operands) / 16-byte (fp32 vectors) epilogue accesses
  if ((a->ldo & 3) || (a->residual && (a->ld_res & 3)) || (a->blend && (a->ld_blend & 3)) || (a->rowvec && (a->ld_rowvec & 3))) return false;
  if ((((size_t)a->out | (size_t)a->residual | (size_t)a->blend) & 7) || (((size_t)a->rowvec | (size_t)a->bias) & 15)) return false;
  return true;
}
static int w320_force() {                                    // tuning aid: TT_W320_FORCE=1|2 routes every eligible problem to the 256- / 128-row kernel whatever the fill
  static int force = -1;
  if (force < 0) { const char* e = getenv("TT_W320_FORCE"); force = e ? atoi(e) : 0; }
  return force;
}
// 0: not served; 1: 256 x 320 tiles (gemm_w320_kernel); 2: 128 x 320 tiles (gemm_w320h_kernel: problems whose 256-row tiles would fill
// less than 70 % of a round, e.g. the second UNet level at 32x56 latents -- 12544 rows, N = 640: 98 x 2 = 196 tiles of 128 rows).
// By default the variant takes conv3x3 problems only: A/B in the step 31.06 (256-row kernel only) / 31.01 (variant for all modes) /
// 30.84 ms (variant for the convs); in isolation +9..16 % on the convs, +-0 on the linears and temporal convs (tools/w320_bench.py).
static int w320_min_fill() {                                 // percent of the CU x round slots the tiles must fill (TT_W320_MIN_FILL, A/B)
  static int fill = -1;
  if (fill < 0) { const char* e = getenv("TT_W320_MIN_FILL"); fill = e ? atoi(e) : 60; }
  return fill;
}
int w320_route(const TtGemmArgs* a) {
  if (!w320_eligible(a)) return 0;
  if (w320_force() == 1 || w320_force() == 2) return w320_force();
  for (int half = 0; half < 2; ++half) {
    if (half && (!(g_w320 & 2) || ((g_w320 & 4) && a->mode != 1))) break;     // tt_gemm_set_big_tile(2): the 256-row kernel only
    const long tiles = (long)ceil_div(a->m, half ? 128 : 256) * (a->n / 320);
    const long rounds = (tiles + 255) / 256;
    if (tiles * 100 >= rounds * 256 * w320_min_fill()) return 1 + half;   // >= 70 % of the CU x round slots busy (196 / 392 / 588 / 784 tiles: 77 %)
  }
  return 0;
}
// Split-K route of the 128 x 320 kernel for the coarse UNet levels (3136 / 784 rows at 32x56 latents: 100 / 28 tiles), where no tile
// shape fills the chip and the tiled 128 x 128 kernel keeps the matrix pipe 35 % busy (profiles/r4_gemm_sq_counters.txt): S workgroups
// per tile, each with >= 20 (conv3x3) / 32 (Linear) slabs of K, at most 256 workgroups and at least 160; the fp32 slabs are summed in a
// fixed order by splitk_epilogue_kernel.  Returns S (0: not served).  Measured (tools/w320_split_probe.py, us, tiled -> split):
//   conv3x3 at 3136 rows, 1280 / 1920 / 2560 -> 1280 channels   129.7 -> 108.5   173.2 -> 145.3   231.0 -> 190.3   (S = 2)
//   FF2 at 3136 rows (K = 5120, S = 2) 64.2 -> 63.5;   conv3x3 at 784 rows (S = 9) 46.4 -> 47.1, 79.7 -> 75.6
// -- the second pass and the fp32 slabs cost ~25 us per problem, so by default only the 3x3 convs with >= 64 tiles take the route;
// tt_gemm_set_big_tile(3) opens it to the Linear problems and the 28-tile level as well (tests, A/B).
int w320_split(const TtGemmArgs* a) {
  if (!w320_eligible(a) || !(g_w320 & 8) || a->ln_fold || a->mode == 2 || w320_route(a)) return 0;
  const long tiles = (long)ceil_div(a->m, 128) * (a->n / 320);
  if ((g_w320 & 4) && (a->mode != 1 || tiles < 64)) return 0;
  const long slabs = (long)(a->mode == 1 ? 9 : 1) * (a->k0 + a->k1) / 64;
  long s = 256 / tiles;
  const long by_k = slabs / (a->mode == 1 ? 20 : 32);
  if (s > by_k) s = by_k;
  if (s > 16) s = 16;
  if (s < 2 || tiles * s < 160 || s * a->m * a->n * 4 >= (1L << 31)) return 0;
  return (int)s;
}
bool w320_ok(const TtGemmArgs* a) { return w320_route(a) != 0; }
static bool ws_fits(const TtGemmArgs* a, int split) { return a->ws && (size_t)a->ws_bytes >= (size_t)split * a->m * a->n * sizeof(float); }
// rows the persistent kernel takes when the problem is launched in two parts (0: one launch)
// (also for row counts that ARE whole tile rows but leave the last round of tiles poorly filled: 10752 x 5120 at 256x384 -- 840 tiles = 3.3
// rounds -- runs its first 38 tile rows, 2.97 rounds, on the persistent kernel and the last 1024 rows on the tiled one; at most 1/8 of the
// rows go to the tail)
static int pp_split_rows(const TtGemmArgs* a) {
  if (a->m < 512 || pp_ok(a)) return 0;
  static int whole = -1;                                                     // TT_PP_SPLIT_WHOLE=0: ragged row counts only (A/B)
  if (whole < 0) { const char* e = getenv("TT_PP_SPLIT_WHOLE"); whole = e ? atoi(e) : 1; }
  if ((a->m & 255) == 0 && (!whole || w320_route(a) || sq320_ok(a))) return 0;       // (whole tile rows: only problems no big-tile kernel of their own serves)
  TtGemmArgs head = *a;
  for (int tr = a->m >> 8; tr >= 2 && (long)(a->m - tr * 256) * 8 <= a->m; --tr) {
    head.m = tr * 256;
    if (head.m < a->m && pp_ok(&head)) return head.m;
  }
  return 0;
}
static void pp_split(const TtGemmArgs* a, int rows, TtGemmArgs* head, TtGemmArgs* tail) {
  const size_t es = 2;                                      // 16-bit storage (pp_ok)
  *head = *a; *tail = *a;
  head->m = rows;
  tail->m = a->m - rows;
  tail->a0 = (const char*)a->a0 + (size_t)rows * a->lda0 * es;
  tail->out = (char*)a->out + (size_t)rows * a->ldo * es;
}

// tile shapes whose fused-LayerNorm variants are built (launch<Tag>() in gemm_kernel.h): the ones the planner picks
static bool ln_capable(int cfg) { return cfg == 1 || cfg == 2 || cfg == 3 || cfg == 7 || cfg == 9 || cfg == 11 || cfg == 16; }
static Plan plan_for(const TtGemmArgs* a) {
  if (a->dtype == TT_F32) {
    // split16: the 64 x 64 tiles' 32 x 32 wave tiles convert two operand fragments per product block (VALU-bound, ~110 TFLOP/s against
    // ~210 on the 128 x 128 tiles' 64 x 64 wave tiles), so the big tile is taken from 160 tiles on (3136 x 1280: 250 tiles = one round)
    static int split_min = -1;
    if (split_min < 0) { const char* e = getenv("TT_F32_SPLIT_MIN_TILES"); split_min = e ? atoi(e) : 160; }
    // (256 x 128 tiles with 8 waves -- the same wave tiles, 25 % less LDS fill per flop, one workgroup per CU -- measured SLOWER:
    // 109.3 -> 114.1 ms / step, one call; the loop is bound by the VALU conversions + MFMA issue of its two waves per SIMD, not by the fill)
    // Other shapes of the big tile, each one gpurun call against 106-108 ms / step for this one: 128 x 128 x 16 with a 4-deep / 3-deep ring
    // (three / two tiles in flight instead of one) 112.7 / 112.0; x 32 with a 3-deep ring (96 KiB: one workgroup per CU) 135.5; 8 waves
    // as 4 x 2 / 2 x 4 (wave tiles 32 x 64 / 64 x 32 at a 128-register budget) 123.4 / 234.2; 128 x 64 x 32, 3-deep 127.7.
    // (4 waves as 4 x 1 -- every A fragment converted by one wave instead of two -- 105.1 -> 104.0 ms, within the noise; as 1 x 4 149.6.)
    // Where the time goes (ablation builds `make variant DEFS=-DTT_SPLIT_ABL=1|2|3`, wrong numbers, one call): 105.4 ms as built; without
    // any conversion 74.8; with one MFMA per product block instead of three 75.1; with neither 65.9 -- conversions and the two extra MFMAs
    // cost ~9 ms each on their own and ~40 ms together: inside one wave they run back to back (the fragment consumers are pinned by
    // sched_barrier: the raw-read hazard of gemm_kernel.h), so the matrix pipe idles while a wave converts.
    if (f32_split()) {
      // split16, few tiles and a long K (the coarsest level's convs: 784 rows x 1280 x 11 520 = 70 tiles of 128 x 128, 360 K steps of 32): the
      // 64 x 64 tiles ran them at ~100 TFLOP/s (260 workgroups, one round, VALU-bound wave tiles).  The K loop split over S workgroups per
      // 128 x 128 tile, fp32 slabs summed in a fixed order by the reduction pass (the 16-bit modes' plan for these shapes; bit-reproducible).
      static int sk = -1;
      if (sk < 0) { const char* e = getenv("TT_F32_SPLITK"); sk = e ? atoi(e) : 1; }
      const long b128 = (long)ceil_div(a->m, 128) * ceil_div(a->n, 128);
      const long kt = (long)(a->mode == 1 ? 9 : (a->mode == 2 ? 3 : 1)) * (a->k0 + a->k1) / 32;
      if (sk && !a->geglu && !a->ln_fold && !a->out_col_hw && b128 < split_min && kt >= 64) {
        long sp = 448 / b128;
        if (sp > kt / 16) sp = kt / 16;
        if (sp > 16) sp = 16;
        if (sp >= 2) return Plan{0, (int)sp};
      }
    }
    return Plan{plan_f32(a->m, a->n, f32_split() ? split_min : 256), 1};
  }
  const int taps = a->mode == 1 ? 9 : (a->mode == 2 ? 3 : 1);
  const bool allow = !a->geglu && !a->ln_fold && !a->out_fp8;       // a K slice would see only part of a LayerNorm row
  const bool k128 = (a->k0 & 127) == 0 && (a->k1 & 127) == 0;
  Plan pl = make_plan(a->m, a->n, (long)taps * (a->k0 + a->k1), allow, !(a->residual || a->blend || a->rowvec), a->mode == 0, k128);
  if (a->ln_fold && !ln_capable(pl.cfg)) {           // a forced tile shape without the fused variant: planner's own choice
    const int keep = g_forced_cfg;
    g_forced_cfg = -1;
    pl = make_plan(a->m, a->n, (long)taps * (a->k0 + a->k1), false, !(a->residual || a->blend || a->rowvec), a->mode == 0);
    g_forced_cfg = keep;
  }
  return pl;
}

// the plan of a problem whose split plan cannot be served (no / too small a workspace)
static Plan unsplit_plan(const TtGemmArgs* a) {
  if (a->dtype == TT_F32) return Plan{plan_f32(a->m, a->n, 256), 1};
  return Plan{make_plan(a->m, a->n, 0, false, !(a->residual || a->blend || a->rowvec)).cfg, 1};
}

extern "C" int tt_gemm_plan(const TtGemmArgs* a, int32_t cfg[7]) {
  if (!a || !cfg || a->m <= 0 || a->n <= 0) TT_FAIL(TT_EINVAL, "tt_gemm_plan: bad arguments");
  if (pp_ok(a) || pp_split_rows(a)) {             // the persistent ping-pong kernel (all or all but the last < 256 rows): stages = 0 marks it (gemm_pp_kernel<dtype, ln, geglu>)
    cfg[0] = 256; cfg[1] = 256; cfg[2] = 64; cfg[3] = 0; cfg[4] = 2; cfg[5] = 4; cfg[6] = 1;
    return TT_OK;
  }
  if (sq320_ok(a)) {          // the streaming kernel for the 320 x 320 linears: 32-row tiles, ring depth 3 (5 without residual)
    cfg[0] = SQ_ROWS; cfg[1] = SQ_N; cfg[2] = SQ_K; cfg[3] = a->residual ? 3 : 5; cfg[4] = 1; cfg[5] = SQ_WAVES; cfg[6] = 1;
    return TT_OK;
  }
  if (const int route = w320_route(a)) {          // gemm_w320_kernel / gemm_w320h_kernel<dtype, mode, ln>: 256 (128) x 320 tiles, 4 x 2 (2 x 2) waves, stages = 0
    cfg[0] = route == 1 ? 256 : 128; cfg[1] = 320; cfg[2] = 64; cfg[3] = 0; cfg[4] = route == 1 ? 4 : 2; cfg[5] = 2; cfg[6] = 1;
    return TT_OK;
  }
  if (const int split = w320_split(a)) {          // ... its split-K route (needs the workspace, like every split plan below)
    if (ws_fits(a, split)) {
      cfg[0] = 128; cfg[1] = 320; cfg[2] = 64; cfg[3] = 0; cfg[4] = 2; cfg[5] = 2; cfg[6] = split;
      return TT_OK;
    }
  }
  Plan pl = plan_for(a);
  if (pl.splitk > 1 && (!a->ws || (size_t)a->ws_bytes < (size_t)pl.splitk * a->m * a->n * sizeof(float) ||
                        (long)pl.splitk * a->m * a->n * 4 >= (1L << 31)))
    pl = unsplit_plan(a);
  const TileCfg& t = a->dtype == TT_F32 ? kCfgsF32[pl.cfg] : kCfgs[pl.cfg];
  cfg[0] = t.bm; cfg[1] = t.bn; cfg[2] = t.bk; cfg[3] = t.nst; cfg[4] = t.wgm; cfg[5] = t.wgn; cfg[6] = pl.splitk;
  return TT_OK;
}

// rows per statistics tile of the route tt_gemm takes (0: no statistics epilogue on it); see include/ttvdm.h
extern "C" int32_t tt_gemm_stats_rows(const TtGemmArgs* a) {
  if (!a || a->m <= 0 || a->n <= 0) return 0;
  if (a->geglu || a->out_fp8 || a->out_f32 || a->out_col_hw > 0 || a->ln_fold == 2) return 0;
  if (pp_ok(a) || pp_split_rows(a) || sq320_ok(a)) return 0;
  // the statistics variants exist for the straight-line epilogues: bias / scale / row vector of >= 32-row groups (or the even / odd
  // form) / residual / a blend with the residual itself -- not for the in-pass operand loads (gemm_kernel.h, `inpass`)
  int32_t cfg[7];
  if (tt_gemm_plan(a, cfg) != TT_OK) return 0;
  const int seg = a->stats_seg;
  if (cfg[6] != 1) {                                                   // split-K: the reduction kernel takes the sums, on tiles of any height:
    if (seg <= 0) return 0;                                            // the largest divisor of the consumer's segment up to 128 rows
    for (int r = seg < 128 ? seg : 128; r > 0; --r)
      if (seg % r == 0) return a->m % r == 0 ? r : 0;
    return 0;
  }
  if (a->blend && !(a->blend == a->residual && a->ld_blend == a->ld_res)) return 0;
  if (a->rowvec && a->rowvec_rows < 32 && !(a->rowvec_rows == 1 && a->rowvec_mod == 2)) return 0;
  if (cfg[0] > 128 && cfg[1] != 320 && a->residual) return 0;         // 256-row tiles of the tiled template read the residual in-pass
  if (a->dtype == TT_F32 && f32_split() && a->residual) return 0;      // ... and so do the split-fp16 product variants on every tile shape
  if (cfg[1] == 320 || cfg[3] == 0) {                                  // the big-tile kernels: whole tiles; the 128-row one also the 64 rows of a wave row
    if (a->m % cfg[0]) return 0;
    return (cfg[0] == 128 && seg > 0 && seg % 128 && seg % 64 == 0) ? 64 : cfg[0];
  }
  // the tiled template: whole tiles (BM rows) if they divide the segment (or no hint), else the rows of one wave row (BM / WGM: 32 or 64)
  const int wave_rows = cfg[0] / cfg[4];
  const int r = (seg <= 0 || seg % cfg[0] == 0) ? cfg[0] : (seg % wave_rows == 0 ? wave_rows : cfg[0]);
  return a->m % r == 0 ? r : 0;
}

// GroupNorm inside the split-K reduction (TtGemmArgs.gn_out): see include/ttvdm.h
extern "C" int32_t tt_gemm_gn_fused(const TtGemmArgs* a) {
  if (!a || a->m <= 0 || a->n <= 0 || a->dtype == TT_F32) return 0;
  if (a->geglu || a->out_fp8 || a->out_f32 || a->out_col_hw > 0 || a->stats_out) return 0;
  const int seg = a->stats_seg;
  if (seg <= 0 || a->m % seg || (a->n & 127)) return 0;               // 32 groups of a whole number of column quads
  if (pp_ok(a) || pp_split_rows(a) || sq320_ok(a)) return 0;
  int32_t cfg[7];
  if (tt_gemm_plan(a, cfg) != TT_OK || cfg[6] <= 1) return 0;         // only where a reduction pass runs anyway
  return splitk_gn_rows(seg, a->n, nullptr) > 0 ? 1 : 0;
}

extern "C" size_t tt_gemm_ws_bytes(const TtGemmArgs* a) {
  if (!a || a->m <= 0 || a->n <= 0) return 0;
  if (pp_ok(a)) return 0;
  if (!pp_split_rows(a) && (sq320_ok(a) || w320_ok(a))) return 0;
  if (const int rows = pp_split_rows(a)) { TtGemmArgs head, tail; pp_split(a, rows, &head, &tail); return tt_gemm_ws_bytes(&tail); }
  if (const int split = w320_split(a)) return (size_t)split * a->m * a->n * sizeof(float);
  const Plan pl = plan_for(a);
  return pl.splitk > 1 ? (size_t)pl.splitk * a->m * a->n * sizeof(float) : 0;
}

extern "C" int tt_gemm(const TtGemmArgs* a, tt_stream_t stream) {
  if (!a || !a->a0 || !a->w || !a->out) TT_FAIL(TT_EINVAL, "tt_gemm: null operand");
  if (a->m <= 0 || a->n <= 0 || a->k0 <= 0) TT_FAIL(TT_EINVAL, "tt_gemm: empty problem m=%d n=%d k0=%d", a->m, a->n, a->k0);
  if (a->dtype != TT_BF16 && a->dtype != TT_F16 && a->dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_gemm: bad dtype");
  if ((a->k0 & 7) || (a->k1 & 7) || (a->n & 3)) TT_FAIL(TT_EINVAL, "tt_gemm: k0/k1 must be multiples of 8 and n of 4");
  if ((a->lda0 & 7) || (a->k1 && (a->lda1 & 7)) || (a->ldw & 7)) TT_FAIL(TT_EINVAL, "tt_gemm: row strides must be multiples of 8 elements");
  if (a->k1 && !a->a1) TT_FAIL(TT_EINVAL, "tt_gemm: k1 > 0 without a1");
  if (a->mode < 0 || a->mode > 2) TT_FAIL(TT_EINVAL, "tt_gemm: bad mode %d", a->mode);
  if (const int rows = pp_split_rows(a)) {                  // whole tile rows -> persistent kernel, the ragged rest -> tiled kernel
    TtGemmArgs head, tail;
    pp_split(a, rows, &head, &tail);
    const int rc = tt_gemm(&head, stream);
    return rc != TT_OK ? rc : tt_gemm(&tail, stream);
  }
  if (a->geglu && ((a->n & 15) || a->residual || a->blend || a->rowvec || a->out_f32 || a->out_col_hw))
    TT_FAIL(TT_EINVAL, "tt_gemm: geglu needs n %% 16 == 0 and no other epilogue terms");
  if (a->rowvec && a->rowvec_rows <= 0) TT_FAIL(TT_EINVAL, "tt_gemm: rowvec_rows");
  if (a->gn_out) {
    if (!a->gn_gamma || !a->gn_beta || (a->ld_gn & 3) || ((size_t)a->gn_out & 7)) TT_FAIL(TT_EINVAL, "tt_gemm: gn_out needs gn_gamma, gn_beta, ld_gn %% 4 == 0 and an 8-byte aligned pointer");
    if (!tt_gemm_gn_fused(a)) TT_FAIL(TT_EUNSUPPORTED, "tt_gemm: gn_out on a problem without a split-K reduction pass (tt_gemm_gn_fused(args) == 0)");
  }
  if (a->stats_out && tt_gemm_stats_rows(a) == 0)
    TT_FAIL(TT_EUNSUPPORTED, "tt_gemm: stats_out on a route without a statistics epilogue (tt_gemm_stats_rows(args) == 0)");
  if (a->rowvec_mod < 0 || (a->rowvec_mod > 0 && !a->rowvec)) TT_FAIL(TT_EINVAL, "tt_gemm: rowvec_mod %d (>= 0, needs rowvec)", a->rowvec_mod);
  if (a->ln_fold < 0 || a->ln_fold > 2) TT_FAIL(TT_EINVAL, "tt_gemm: ln_fold %d (0 none, 1 rows of A, 2 rows of W)", a->ln_fold);
  if (a->ln_fold && (a->mode != 0 || a->k1 != 0 || !(a->ln_eps > 0.f)))
    TT_FAIL(TT_EINVAL, "tt_gemm: ln_fold needs mode 0, one source spanning the whole LayerNorm row (k0 = C) and ln_eps > 0");
  if (a->ln_fold == 2 && a->geglu) TT_FAIL(TT_EINVAL, "tt_gemm: ln_fold 2 (columns) cannot be combined with geglu");
  if (a->out_fp8 && (a->dtype == TT_F32 || a->mode != 0 || a->geglu || a->residual || a->blend || a->rowvec || a->out_f32 || (a->ldo & 3)))
    TT_FAIL(TT_EINVAL, "tt_gemm: out_fp8 is for plain 16-bit linears (no geglu / residual / blend / rowvec / fp32 out), ldo %% 4 == 0");
  GemmP p;
  p.a0 = (const char*)a->a0; p.a1 = (const char*)a->a1; p.k0 = a->k0; p.k1 = a->k1;
  p.lda0 = a->lda0; p.lda1 = a->lda1; p.w = (const char*)a->w; p.ldw = a->ldw;
  p.m = a->m; p.n = a->n; p.mode = a->mode;
  p.nimg = a->nimg; p.hin = a->hin; p.win = a->win; p.hout = a->hout; p.wout = a->wout;
  p.stride = a->stride; p.upsample = a->upsample; p.frames = a->frames; p.hw = a->hw;
  p.bias = a->bias; p.acc_scale = a->acc_scale;
  p.rowvec = a->rowvec; p.rowvec_rows = a->rowvec_rows; p.ld_rowvec = a->ld_rowvec; p.rowvec_mod = a->rowvec ? a->rowvec_mod : 0;
  p.geglu = a->geglu; p.residual = (const char*)a->residual; p.ld_res = a->ld_res;
  p.blend = (const char*)a->blend; p.ld_blend = a->ld_blend; p.alpha = a->alpha;
  const int es = a->dtype == TT_F32 ? 4 : 2;         // bytes per stored element
  p.out = (char*)a->out; p.ldo = a->ldo; p.out_f32 = a->dtype == TT_F32 ? 0 : a->out_f32;   // TT_F32 stores fp32 anyway
  p.out_col_hw = a->out_col_hw; p.out_col_hwp = a->out_col_hwp;
  p.ln_fold = a->ln_fold; p.ln_eps = a->ln_eps; p.out_fp8 = a->out_fp8; p.stats = a->stats_out; p.stat_rows = a->gn_out ? a->stats_seg : (a->stats_out ? tt_gemm_stats_rows(a) : 0);
  p.f32_split = a->dtype == TT_F32 ? f32_split() : 0;
  p.presplit = a->presplit;
  if (a->presplit) {
    if ((a->presplit & ~3) || a->dtype != TT_F32 || !p.f32_split)
      TT_FAIL(TT_EINVAL, "tt_gemm: presplit operands need TT_F32 with tt_gemm_set_f32_split(1)");
    if (((a->presplit & 1) && a->ln_fold == 1) || ((a->presplit & 2) && a->ln_fold == 2))
      TT_FAIL(TT_EINVAL, "tt_gemm: the LayerNorm statistics cannot be taken from a pre-split operand");
    if ((a->presplit & 1) && a->mode != 0) TT_FAIL(TT_EINVAL, "tt_gemm: a pre-split A operand is a Linear operand (mode 0)");
  }
  p.gn_out = (char*)a->gn_out; p.ld_gn = a->ld_gn; p.gn_gamma = a->gn_gamma; p.gn_beta = a->gn_beta; p.gn_eps = a->gn_eps; p.gn_silu = a->gn_silu;
  if (p.mode == 1) {
    if (p.nimg <= 0 || p.hin <= 0 || p.win <= 0 || p.hout <= 0 || p.wout <= 0 || p.stride < 1)
      TT_FAIL(TT_EINVAL, "tt_gemm: conv geometry");
    if ((long)p.nimg * p.hout * p.wout != p.m) TT_FAIL(TT_EINVAL, "tt_gemm: m != nimg*hout*wout");
  }
  if (p.mode == 2) {
    if (p.frames <= 0 || p.hw <= 0 || p.m % ((long)p.frames * p.hw)) TT_FAIL(TT_EINVAL, "tt_gemm: tconv geometry");
  }
  p.taps = p.mode == 1 ? 9 : (p.mode == 2 ? 3 : 1);
  {
    const long rows = p.mode == 1 ? (long)p.nimg * p.hin * p.win : (long)p.m;
    const long a0b = ((rows - 1) * p.lda0 + p.k0) * es, a1b = p.k1 ? ((rows - 1) * p.lda1 + p.k1) * es : 16;
    const long wb = ((long)(p.n - 1) * p.ldw + (long)p.taps * (p.k0 + p.k1)) * es;
    if (a0b >= (1L << 31) || a1b >= (1L << 31) || wb >= (1L << 31))
      TT_FAIL(TT_EUNSUPPORTED, "tt_gemm: operand larger than 2 GiB (32-bit buffer offsets)");
    p.a0_bytes = (unsigned)a0b; p.a1_bytes = (unsigned)a1b; p.w_bytes = (unsigned)wb;
    // epilogue operands (bounds-checked descriptors; 0 bytes = absent -> loads return 0)
    const long n_out = p.geglu ? p.n / 2 : p.n;
    const long outb = ((long)(p.m - 1) * p.ldo + (p.out_col_hw > 0 ? p.ldo : n_out)) * (p.out_f32 ? 4 : (p.out_fp8 ? 1 : es));
    const long resb = p.residual ? ((long)(p.m - 1) * p.ld_res + p.n) * es : 0;
    const long blb = p.blend ? ((long)(p.m - 1) * p.ld_blend + p.n) * es : 0;
    long rv_last = p.rowvec ? (p.m - 1) / p.rowvec_rows : 0;             // last row-vector index the kernel can form
    if (p.rowvec_mod > 0 && rv_last > p.rowvec_mod - 1) rv_last = p.rowvec_mod - 1;
    const long rvb = p.rowvec ? (rv_last * p.ld_rowvec + p.n) * 4 : 0;
    if (outb >= (1L << 31) || resb >= (1L << 31) || blb >= (1L << 31) || rvb >= (1L << 31))
      TT_FAIL(TT_EUNSUPPORTED, "tt_gemm: epilogue operand larger than 2 GiB (32-bit buffer offsets)");
    p.out_bytes = (unsigned)outb; p.res_bytes = (unsigned)resb; p.blend_bytes = (unsigned)blb;
    p.bias_bytes = p.bias ? (unsigned)p.n * 4u : 0u; p.rowvec_bytes = (unsigned)rvb; p.ws_bytes = 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (pp_ok(a)) {
    p.splitk = 1; p.ws = nullptr; p.ws_bytes = 0; p.group_m_override = group_m_override(); p.group_m = 1;
    if (a->dtype == TT_BF16) launch_pp_bf16(p, st); else launch_pp_f16(p, st);
    TT_CHECK_LAUNCH("tt_gemm");
    return TT_OK;
  }
  if (sq320_ok(a)) {
    p.splitk = 1; p.ws = nullptr; p.ws_bytes = 0; p.group_m_override = 0; p.group_m = 1;
    if (a->dtype == TT_BF16) launch_sq320_bf16(p, st); else launch_sq320_f16(p, st);
    TT_CHECK_LAUNCH("tt_gemm");
    return TT_OK;
  }
  if (const int route = w320_route(a)) {
    p.splitk = 1; p.ws = nullptr; p.ws_bytes = 0; p.group_m_override = 0; p.group_m = 1;
    if (route == 1) { if (a->dtype == TT_BF16) launch_w320_bf16(p, st); else launch_w320_f16(p, st); }
    else { if (a->dtype == TT_BF16) launch_w320h_bf16(p, st); else launch_w320h_f16(p, st); }
    TT_CHECK_LAUNCH("tt_gemm");
    return TT_OK;
  }
  if (const int split = w320_split(a)) {
    if (ws_fits(a, split)) {                      // (no workspace: the tiled kernels' un-split plan below)
      p.splitk = split; p.ws = (float*)a->ws; p.ws_bytes = (unsigned)((long)split * a->m * a->n * 4); p.group_m_override = 0; p.group_m = 1;
      if (a->dtype == TT_BF16) launch_w320h_bf16(p, st); else launch_w320h_f16(p, st);
      TT_CHECK_LAUNCH("tt_gemm");
      return TT_OK;
    }
  }
  Plan pl = plan_for(a);
  if (pl.splitk > 1 && (!a->ws || (size_t)a->ws_bytes < (size_t)pl.splitk * a->m * a->n * sizeof(float)))
    pl = unsplit_plan(a);            // no workspace: un-split plan (still correct)
  if (pl.splitk > 1 && (long)pl.splitk * a->m * a->n * 4 >= (1L << 31))
    pl = unsplit_plan(a);            // slabs beyond the 32-bit offsets: un-split plan
  p.splitk = pl.splitk;
  p.group_m_override = group_m_override(); p.group_m = 1;
  p.ws = (float*)a->ws;
  p.ws_bytes = pl.splitk > 1 ? (unsigned)((long)pl.splitk * a->m * a->n * 4) : 0u;
  if (a->dtype == TT_BF16) launch_bf16(p, pl.cfg, st);
  else if (a->dtype == TT_F16) launch_f16(p, pl.cfg, st);
  else launch_f32(p, pl.cfg, st);
  TT_CHECK_LAUNCH("tt_gemm");
  return TT_OK;
}
