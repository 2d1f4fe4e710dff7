// gemm_kernel.h -- the tt_gemm kernel family (included by the per-dtype instantiation units gemm_inst_*.hip, which
// compile in parallel, and by gemm.hip for the launch-parameter struct).
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace ttg {


// Division by a launch-uniform divisor without the ~25-instruction software sequence (x2 for the remainder): the tiled template
// ran ~500 scalar instructions before its first load, most of them integer divisions, and the scalar unit is shared by all waves
// of a CU -- 16 waves x 500 instructions ~ 2.4 us of every launch (tools/gemm_timeline.py).
//   q = floor(n / d) = (n * mul) >> sh   for 0 <= n < 2^31, 1 <= d < 2^31,  mul = ceil(2^(31 + l) / d), sh = 31 + l, l = ceil(log2 d)
// (Granlund-Montgomery with N = 31: exact; mul <= 2^32 only for d = 1 ... handled: l = 0 -> mul = 2^31, sh = 31.)
struct FastDiv { unsigned mul, sh; };
static inline FastDiv make_fastdiv(long d_) {
  unsigned long long d = d_ < 1 ? 1 : (unsigned long long)d_;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  FastDiv f;
  f.mul = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  f.sh = 31 + l;
  return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.sh); }

struct GemmP {
  const char* a0; const char* a1;
  int k0, k1; long lda0, lda1;
  const char* w; long ldw;
  int m, n, mode;
  int nimg, hin, win, hout, wout, stride, upsample;
  int frames, hw;
  const float* bias; float acc_scale;
  const float* rowvec; int rowvec_rows; long ld_rowvec;
  int rowvec_mod;                   // > 0: row m takes rowvec[(m / rowvec_rows) % rowvec_mod] (TtGemmArgs.rowvec_mod)
  int geglu;
  const char* residual; long ld_res;
  const char* blend; long ld_blend; float alpha;
  char* out; long ldo; int out_f32;
  int out_col_hw, out_col_hwp;
  int nk0, nk1, taps, kt_total;     // derived: K steps per source, taps, total K steps
  int tiles_m, tiles_n;
  int group_m_override;             // > 0: host override (TT_GEMM_GROUP_M), else launch_cfg picks
  int group_m;                      // tile rows per group of the launch order (>= 1; see the tile mapping in gemm_kernel)
  unsigned a0_bytes, a1_bytes, w_bytes;   // extents for the buffer descriptors (< 2 GiB each)
  unsigned out_bytes, res_bytes, blend_bytes, bias_bytes, rowvec_bytes, ws_bytes;   // epilogue descriptors (0 = absent)
  int splitk;                       // > 1: block (tile, s) reduces K slice s and writes an fp32 slab to ws
  float* ws;                        // [splitk][m][n] fp32 partial sums
  int ln_fold; float ln_eps;        // fused LayerNorm of the A rows (1) / W rows (2): see gemm_kernel, MODE 3 / 4
  int out_fp8;                      // store e4m3 bytes (operands of the fp8 attention path) instead of 16-bit values
  float* stats;                     // != NULL: per (row tile, column) sum and sum of squares of the stored output (TtGemmArgs.stats_out)
  int stat_rows;                    // rows per statistics tile (tt_gemm_stats_rows); with gn_out: rows per GroupNorm segment
  char* gn_out; long ld_gn;         // != NULL: the split-K reduction also writes act(GroupNorm(out)) (TtGemmArgs.gn_out)
  const float* gn_gamma; const float* gn_beta; float gn_eps; int gn_silu;
  int f32_split;                    // TT_F32 only: products on the 16-bit matrix pipe as three split-fp16 terms (tt_gemm_set_f32_split)
  int presplit;                     // ... bit 0: a0 / a1, bit 1: w already hold the fp16 (h, l) pairs (TtGemmArgs.presplit)
  // launch-uniform divisors of the tiled template as multiply-shift pairs (fill_fastdivs, called by launch_cfg)
  FastDiv fd_splitk, fd_per_group, fd_group_m, fd_last_rows, fd_per_tap, fd_hwo, fd_wout, fd_hw, fd_frames, fd_rv_rows, fd_rv_mod;
};

// 4 floats -> 4 OCP e4m3 bytes (saturating at +-448: e4m3fn has no infinity, an overflow would become NaN)
__device__ __forceinline__ unsigned pack4_fp8(const float* v) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[0], -448.f, 448.f), __builtin_amdgcn_fmed3f(v[1], -448.f, 448.f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[2], -448.f, 448.f), __builtin_amdgcn_fmed3f(v[3], -448.f, 448.f), w, true);
  return (unsigned)w;
}
__device__ __forceinline__ void st32(__amdgpu_buffer_rsrc_t r, int off, unsigned a) {
  __builtin_amdgcn_raw_buffer_store_b32(a, r, off, 0, 0);
}

// ---- GroupNorm statistics of the OUTPUT, gathered beside the epilogue (GemmP.stats).  The epilogue passes a 32-row x 64 (32)
// column chunk through the wave's LDS strip; with statistics on, every lane writes the values it STORES (rounded to the storage type,
// as fp32) back into the strip slot it just read, and after the chunk's passes lane c adds up column c of the strip -- 32 rows, fixed
// order, two live registers (a version that kept 8 running sums per lane through the passes spilled 10-17 registers next to the 160
// accumulators of gemm_w320).  The wave's column sums go to its staging rows in LDS (plain store for the wave's first fragment row,
// read-add-store by the same lane for the following ones); the tile-level step then adds the waves of one tile column in a fixed order:
//   stats[(tile_m * 2 + 0) * n + col] = sum,  stats[(tile_m * 2 + 1) * n + col] = sum of squares     (fp32, one row tile = BM rows)
// No atomics: graph replays and eager launches stay bit-identical.
__device__ __forceinline__ void colstat_strip(const char* ebuf, int q_per_row, int lane, float* srow_sum, float* srow_sq, bool first) {
  if (lane < q_per_row * 4) {                                // one lane per column of the chunk
    const int quad = lane >> 2, e4 = (lane & 3) * 4;
    float a = 0.f, b = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const float x = *(const float*)(ebuf + r * 256 + ((quad ^ (r & (q_per_row - 1))) << 4) + e4);      // strip_off(r, quad, q_per_row)
      a += x; b = fmaf(x, x, b);
    }
    if (!first) { a += srow_sum[lane]; b += srow_sq[lane]; }
    srow_sum[lane] = a; srow_sq[lane] = b;
  }
}

// ---- running sum and sum of squares of one 16-byte MFMA operand chunk (fused LayerNorm statistics).  16-bit storage: two
// packed dot products per dword (v_dot2_f32_*: x . (1,1) and x . x, accumulated in fp32) -- the operand stays packed.
typedef __attribute__((ext_vector_type(2))) _Float16 ln_half2;
typedef __attribute__((ext_vector_type(2))) __bf16 ln_bf162;
template <typename Tag> __device__ __forceinline__ void ln_stat(const raw_u32x4_t& f, float& s, float& q);
template <> __device__ __forceinline__ void ln_stat<bf16_tag>(const raw_u32x4_t& f, float& s, float& q) {
  // (hipcc / ROCm 7.2 miscompiles __builtin_bit_cast(.., f[d]) on an ext-vector subscript inside a loop: every
  // iteration reads element 0 -- tools/dot2_test.hip.  Copy the dwords to scalars first.)
  const ln_bf162 one = __builtin_bit_cast(ln_bf162, 0x3F803F80u);
  const unsigned w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const ln_bf162 x = __builtin_bit_cast(ln_bf162, w[d]);
    s = __builtin_amdgcn_fdot2_f32_bf16(x, one, s, false);
    q = __builtin_amdgcn_fdot2_f32_bf16(x, x, q, false);
  }
}
template <> __device__ __forceinline__ void ln_stat<f16_tag>(const raw_u32x4_t& f, float& s, float& q) {
  const ln_half2 one = __builtin_bit_cast(ln_half2, 0x3C003C00u);
  const unsigned w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const ln_half2 x = __builtin_bit_cast(ln_half2, w[d]);
    s = __builtin_amdgcn_fdot2(x, one, s, false);
    q = __builtin_amdgcn_fdot2(x, x, q, false);
  }
}
template <> __device__ __forceinline__ void ln_stat<f32_tag>(const raw_u32x4_t& f, float& s, float& q) {
  const unsigned w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) { const float x = __uint_as_float(w[d]); s += x; q = fmaf(x, x, q); }
}
// fp32 storage (TT_F32, the mode that has to meet atol 1e-4 elementwise): sums of x - c with the pivot c = the row's first
// element, so E[(x-c)^2] - E[x-c]^2 does not cancel when |row mean| >> sigma (unshifted, the fp32 sums cost 6e-5 of the
// output at 5 sigma and 5e-3 at 50 sigma; shifted ~1e-6 at any mean).  The 16-bit types keep the packed dot products:
// their storage rounding (1e-3 / 8e-3) covers the one-pass error up to 50 sigma (tests/test_ops_gpu.py).
__device__ __forceinline__ void ln_stat_shifted(const raw_u32x4_t& f, float c, float& s, float& q) {
  const unsigned w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) { const float x = __uint_as_float(w[d]) - c; s += x; q = fmaf(x, x, q); }
}


// ---- optional per-block timeline (make timeline): thread 0 of every block stamps the 100 MHz wall clock
#ifdef TT_GEMM_TIMELINE
static __device__ long long g_tl[8 * 8192];
#define TL(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_tl[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define TL(i) do { } while (0)
#endif

// ---- branch-free global access for the epilogue: 128-bit buffer descriptors with hardware bounds checking.  An
// offset of kInv (>= num_records) makes a load return 0 and drops a store, so ragged rows/columns and absent operands
// (null base, 0 records) need no exec-masked branch -- with branches hipcc serialises every pass behind
// `s_waitcnt vmcnt(0)` and the epilogue of one tile costs 4-8 us; straight-line it is ~1 us.
constexpr int kInv = (int)0x80000000;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint2 ld64(__amdgpu_buffer_rsrc_t r, int off) {
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ float4 ld128f(__amdgpu_buffer_rsrc_t r, int off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st64(__amdgpu_buffer_rsrc_t r, int off, unsigned a, unsigned b) {
  __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){a, b}, r, off, 0, 0);
}
__device__ __forceinline__ void st128f(__amdgpu_buffer_rsrc_t r, int off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, off, 0, 0);
}
// 4 consecutive elements of the tag's storage type (8 bytes for the 16-bit tags, 16 for f32_tag)
template <typename Tag> __device__ __forceinline__ typename Elem<Tag>::quad_t ldq(__amdgpu_buffer_rsrc_t r, int off) { return ld64(r, off); }
template <> __device__ __forceinline__ uint4 ldq<f32_tag>(__amdgpu_buffer_rsrc_t r, int off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
template <typename Tag> __device__ __forceinline__ void stq(__amdgpu_buffer_rsrc_t r, int off, const float* v) {
  st64(r, off, pack2<Tag>(v[0], v[1]), pack2<Tag>(v[2], v[3]));
}
template <> __device__ __forceinline__ void stq<f32_tag>(__amdgpu_buffer_rsrc_t r, int off, const float* v) {
  st128f(r, off, make_float4(v[0], v[1], v[2], v[3]));
}
template <typename Tag> __device__ __forceinline__ typename Elem<Tag>::quad_t zero_quad();
template <> __device__ __forceinline__ uint2 zero_quad<bf16_tag>() { return make_uint2(0, 0); }
template <> __device__ __forceinline__ uint2 zero_quad<f16_tag>() { return make_uint2(0, 0); }
template <> __device__ __forceinline__ uint4 zero_quad<f32_tag>() { return make_uint4(0, 0, 0, 0); }

// ---- epilogue on 4 consecutive output columns (gn .. gn+3) of row gm, in two halves: epi_load requests the operands (bias, row vector,
// residual, blend), epi_finish does the arithmetic and stores.  The split-K reduction kernels request the operands of all their rows BEFORE
// they sum the slabs, so one memory round trip covers slabs and operands (epilogue_quad = both halves back to back: a round trip per operand).
template <typename Tag> struct EpiOps { float4 bias, rv; typename Elem<Tag>::quad_t res, bl; };
template <typename Tag>
__device__ __forceinline__ EpiOps<Tag> epi_load(const GemmP& p, int gm, int gn) {
  typedef typename Elem<Tag>::quad_t quad_t;
  constexpr int ES = Elem<Tag>::ES;
  EpiOps<Tag> o;
  o.bias = o.rv = make_float4(0.f, 0.f, 0.f, 0.f);
  o.res = o.bl = zero_quad<Tag>();
  if (p.bias) o.bias = *(const float4*)(p.bias + gn);
  if (p.rowvec) {
    int rg = gm / p.rowvec_rows;
    if (p.rowvec_mod > 0) rg %= p.rowvec_mod;
    o.rv = *(const float4*)(p.rowvec + (long)rg * p.ld_rowvec + gn);
  }
  if (p.residual) o.res = *(const quad_t*)(p.residual + ((long)gm * p.ld_res + gn) * ES);
  if (p.blend) o.bl = *(const quad_t*)(p.blend + ((long)gm * p.ld_blend + gn) * ES);
  return o;
}
template <typename Tag>
__device__ __forceinline__ void epi_finish(const GemmP& p, const EpiOps<Tag>& o, int gm, int gn, float (&v)[4]) {
  typedef typename Elem<Tag>::quad_t quad_t;
  constexpr int ES = Elem<Tag>::ES;
  if (p.bias) { v[0] += o.bias.x; v[1] += o.bias.y; v[2] += o.bias.z; v[3] += o.bias.w; }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] *= p.acc_scale;
  if (p.rowvec) { v[0] += o.rv.x; v[1] += o.rv.y; v[2] += o.rv.z; v[3] += o.rv.w; }
  if (p.residual) {
    float r4[4];
    quad_to_f32<Tag>(o.res, r4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += r4[e];
  }
  if (p.blend) {
    float r4[4];
    quad_to_f32<Tag>(o.bl, r4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = p.alpha * r4[e] + (1.0f - p.alpha) * v[e];
  }
  if (p.out_fp8) {
    if (p.out_col_hw > 0) {
      const unsigned w = pack4_fp8(v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = gn + e;
        const long oc = (long)(c / p.out_col_hw) * p.out_col_hwp + (c % p.out_col_hw);
        *(unsigned char*)(p.out + (long)gm * p.ldo + oc) = (unsigned char)(w >> (8 * e));
      }
    } else {
      *(unsigned*)(p.out + (long)gm * p.ldo + gn) = pack4_fp8(v);
    }
  } else if (p.out_f32) {
    *(float4*)(p.out + ((long)gm * p.ldo + gn) * 4) = make_float4(v[0], v[1], v[2], v[3]);
  } else if (p.out_col_hw > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = gn + e;
      const long oc = (long)(c / p.out_col_hw) * p.out_col_hwp + (c % p.out_col_hw);
      store1<Tag>(p.out + ((long)gm * p.ldo + oc) * ES, v[e]);
    }
  } else {
    *(quad_t*)(p.out + ((long)gm * p.ldo + gn) * ES) = f32_to_quad<Tag>(v);
  }
}

template <typename Tag>
__device__ __forceinline__ void epilogue_quad(const GemmP& p, int gm, int gn, float (&v)[4]) {
  const EpiOps<Tag> o = epi_load<Tag>(p, gm, gn);
  epi_finish<Tag>(p, o, gm, gn, v);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `tiles` of the G-load groups issued last are still in flight (tiles <= 3)
template <int G> __device__ __forceinline__ void wait_tiles(int tiles) {
  static_assert(3 * G <= 63, "vmcnt field is 6 bits");
  if (tiles <= 0) wait_vmcnt<0>();
  else if (tiles == 1) wait_vmcnt<G>();
  else if (tiles == 2) wait_vmcnt<2 * G>();
  else wait_vmcnt<3 * G>();
}

// LDS bytes of one workgroup and the waves per SIMD the register allocator must leave room for: as many workgroups per
// CU as the 160 KiB of LDS admit (at most 2 -- more did not pay), i.e. blocks * waves / 4 SIMDs.  Declaring it keeps e.g.
// the 256x128 8-wave kernel at <= 128 VGPRs (130 would halve its occupancy).
constexpr int gemm_lds_bytes(int BM, int BN, int CPR, int NST, int NT) {      // CPR = 16-byte chunks per tile row
  return NST * (((BM * CPR + NT - 1) / NT) + ((BN * CPR + NT - 1) / NT)) * NT * 16;
}
constexpr int gemm_min_waves(int BM, int BN, int CPR, int NST, int NT) {
  const int blocks = (160 * 1024) / gemm_lds_bytes(BM, BN, CPR, NST, NT) >= 2 ? 2 : 1;
  const int w = blocks * (NT / 64) / 4;
  return w < 1 ? 1 : w;
}

// KMODE: 0 Linear, 1 conv3x3, 2 temporal conv, and two Linear variants with a fused LayerNorm (transformer blocks):
//   3  out = LN(A rows) W^T        4  out = A LN(W rows)^T   (the swapped V^T projection: the tokens are the "W" operand)
// The caller folds gamma into the weights, centres them over k (rows sum to zero, so the mean of x drops out of x W^T) and
// folds beta into the bias; what remains is the per-token 1/sigma.  Every lane sees ALL K values of "its" operand row pass
// through its registers as MFMA fragments (row l31, chunk parity hi), so sum and sum of squares cost two packed dot
// products per dword next to the MFMAs -- no statistics pass, no extra memory traffic, no normalised copy of the tensor.
// One-pass variance (E[x^2] - E[x]^2, fp32 sums): relative error ~6e-8 * (1 + mean^2 / var), i.e. fine for |row mean| <= ~50 sigma
// (tests/test_ops_gpu.py::test_gemm_fused_layernorm_rows_with_large_row_means); far beyond that use tt_layernorm + a plain GEMM.
template <typename Tag, int BM, int BN, int BK, int NST, int WGM, int WGN, int KMODE_>
__global__ __launch_bounds__(64 * WGM * WGN, gemm_min_waves(BM, BN, BK / Elem<Tag>::EPC, NST, 64 * WGM * WGN))
void gemm_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool SPLIT = (KMODE_ & 8) != 0;            // f32_tag only: KMODE + 8 = the same kernel with split-fp16 products (see `mma`)
  constexpr bool PRE_B = (KMODE_ & 16) != 0, PRE_A = (KMODE_ & 32) != 0;      // ... + 16 / + 32: the W / A operand arrives pre-split (GemmP.presplit)
  constexpr int KMODE = KMODE_ & 7;
  static_assert(SPLIT || (!PRE_A && !PRE_B), "pre-split operands belong to the split variants");
  static_assert(!SPLIT || std::is_same<Tag, f32_tag>::value, "split products are a TT_F32 mode");
  constexpr int MODE = KMODE >= 3 ? 0 : KMODE;         // gather mode
  constexpr int LN = KMODE >= 3 ? KMODE - 2 : 0;       // 0 none, 1 statistics of A rows, 2 of W rows
  TL(0);
  kernarg_touch<sizeof(GemmP)>();
  typedef typename Elem<Tag>::quad_t quad_t;
  constexpr int ES = Elem<Tag>::ES, EPC = Elem<Tag>::EPC;   // bytes per element, elements per 16-byte chunk
  constexpr int NT = 64 * WGM * WGN;                   // threads
  constexpr int CPR = BK / EPC;                        // 16-byte chunks per tile row
  constexpr int WTM = BM / WGM, WTN = BN / WGN, FM = WTM / 32, FN = WTN / 32;
  // 16-byte chunks staged per thread per tile; when the tile does not divide evenly the last pass is padded
  // (rows >= BM/BN of the LDS image are never read; their lanes load the zero page so every wave issues the
  // same number of loads and the counted vmcnt stays valid)
  constexpr int AR = (BM * CPR + NT - 1) / NT, BR = (BN * CPR + NT - 1) / NT;
  constexpr int A_BYTES = AR * NT * 16, B_BYTES = BR * NT * 16, STAGE = A_BYTES + B_BYTES;
  constexpr int G = AR + BR;
  constexpr int RPI = NT / CPR;                        // tile rows covered per staging pass
  static_assert(NT % CPR == 0 && NST >= 2 && NST <= 5 && WTM % 32 == 0 && WTN % 32 == 0, "tile/threads mismatch");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware tile order: the dispatcher places block b on XCD b%8; give each XCD a contiguous run of
  // tiles (same A rows, neighbouring W columns) so its private L2 sees the reuse (guide T1, bijective form).
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int split = 0;                                            // slices of one tile sit next to each other
  if (p.splitk > 1) { const int t = fdiv(bid, p.fd_splitk); split = bid - t * p.splitk; bid = t; }
  // Within an XCD's run the tiles go in groups of `group_m` tile rows, column by column inside a group: the ~64 workgroups
  // that are resident on an XCD at a time then cover group_m rows x 64/group_m columns, i.e. every K slice they fetch is
  // shared by group_m (W) or 64/group_m (A) of them instead of the whole window sharing ONE A row and streaming every W
  // column.  The K loop is bound by the LDS fill (ablation in DESIGN.md section 6: without the loads the same loop runs
  // ~2x faster, without the MFMAs only ~1.2x), and what misses L2 pays the fabric latency.
  int tile_m, tile_n;
  {
    const int gm = p.group_m;
    const int per_group = gm * p.tiles_n;
    const int grp = fdiv(bid, p.fd_per_group);
    const int first = grp * gm;
    const int r = bid - grp * per_group;
    // every group has gm tile rows except a ragged last one (tiles_m % gm rows): two precomputed divisors
    const bool full = first + gm <= p.tiles_m;
    const int rows = full ? gm : p.tiles_m - first;
    tile_n = full ? fdiv(r, p.fd_group_m) : fdiv(r, p.fd_last_rows);
    tile_m = first + (r - tile_n * rows);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // K-tile range of this block
  // (kt_total * splitk < 2^31: at most a few thousand K steps, splitk <= 16)
  const int kt_lo = p.splitk > 1 ? fdiv(p.kt_total * split, p.fd_splitk) : 0;
  const int KT = p.splitk > 1 ? fdiv(p.kt_total * (split + 1), p.fd_splitk) - kt_lo : p.kt_total;

  // ---- staging: buffer_load ... lds through 128-bit resource descriptors.  Every lane owns a 32-bit byte offset per
  // staged 16-byte chunk (computed once per tile, or once per conv tap); the K position is a SCALAR offset.  Lanes that
  // must read zeros (ragged M/N, conv halo, K tail) use an offset beyond num_records: the hardware bounds check returns 0.
  // This keeps the per-K-step instruction count at ~1 per load (the loop is otherwise instruction-issue bound).
  constexpr int INV = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a0, 0, p.a0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a1 ? p.a1 : p.a0), 0, p.a1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const int crow = tid / CPR, cchunk = tid % CPR;
  int a_chunk[AR];              // source chunk (swizzled) per staged row
  int va0[AR], va1[AR];         // byte offsets into a0 / a1 for the current tap
  int a_img[AR], a_y[AR], a_x[AR];
  bool a_valid[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = i * RPI + crow;
    a_chunk[i] = cchunk ^ tile_swz<CPR>(r);
    const int gm = m0 + r;
    a_valid[i] = gm < p.m && r < BM;
    const int g = a_valid[i] ? gm : 0;
    va0[i] = va1[i] = INV;
    a_img[i] = a_y[i] = a_x[i] = 0;
    if constexpr (MODE == 0) {
      if (a_valid[i]) {
        va0[i] = (int)(((long)g * p.lda0 + a_chunk[i] * EPC) * ES);
        va1[i] = (int)(((long)g * p.lda1 + a_chunk[i] * EPC) * ES);
      }
    } else if constexpr (MODE == 1) {
      const int hwo = p.hout * p.wout;
      a_img[i] = fdiv(g, p.fd_hwo);
      const int rem = g - a_img[i] * hwo;
      a_y[i] = fdiv(rem, p.fd_wout);
      a_x[i] = rem - a_y[i] * p.wout;
    } else {
      const int fr = fdiv(g, p.fd_hw);
      a_img[i] = fr - fdiv(fr, p.fd_frames) * p.frames;   // frame index
      a_y[i] = g;                         // row
    }
  }
  int b_chunk[BR], vb[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int r = i * RPI + crow;
    b_chunk[i] = cchunk ^ tile_swz<CPR>(r);
    const int gn = n0 + r;
    vb[i] = (gn < p.n && r < BN) ? (int)(((long)gn * p.ldw + b_chunk[i] * EPC) * ES) : INV;
  }

  // K-step iterator state for the NEXT tile to stage (uniform)
  int s_tap, s_src, s_kc;
  {
    const int per_tap = p.nk0 + p.nk1;
    s_tap = kt_lo ? fdiv(kt_lo, p.fd_per_tap) : 0;
    const int rem = kt_lo - s_tap * per_tap;
    s_src = rem >= p.nk0 ? 1 : 0;
    s_kc = rem - (s_src ? p.nk0 : 0);
  }
  bool tap_dirty = true;
  auto stage = [&](int slot) {
    if constexpr (MODE != 0) {
      if (tap_dirty) {                       // new tap: refresh the per-lane row offsets (uniform branch)
        tap_dirty = false;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          bool ok = a_valid[i];
          long row;
          if constexpr (MODE == 1) {
            const int dy = s_tap / 3 - 1, dx = s_tap - (dy + 1) * 3 - 1;
            int iy = a_y[i] * p.stride + dy, ix = a_x[i] * p.stride + dx;
            const int hv = p.upsample ? p.hin * 2 : p.hin, wv = p.upsample ? p.win * 2 : p.win;
            ok = ok && iy >= 0 && iy < hv && ix >= 0 && ix < wv;
            if (p.upsample) { iy >>= 1; ix >>= 1; }
            row = ((long)a_img[i] * p.hin + iy) * p.win + ix;
          } else {
            const int f = a_img[i] + s_tap - 1;
            ok = ok && f >= 0 && f < p.frames;
            row = (long)a_y[i] + (long)(s_tap - 1) * p.hw;
          }
          va0[i] = ok ? (int)((row * p.lda0 + a_chunk[i] * EPC) * ES) : INV;
          va1[i] = ok ? (int)((row * p.lda1 + a_chunk[i] * EPC) * ES) : INV;
        }
      }
    }
    // everything below is wave-uniform except the per-lane offsets; force the scalars into SGPRs so the buffer
    // instructions get their soffset / descriptor without waterfall loops (cdna guide T20)
    const int src = __builtin_amdgcn_readfirstlane(s_src);
    const int ksrc = src ? p.k1 : p.k0;
    const int kbase = __builtin_amdgcn_readfirstlane(s_kc) * BK;
    const bool tail = kbase + BK > ksrc;     // only the last K step of a source can have dead chunks
    char* lds_a = smem + slot * STAGE + wid * 1024;
    char* lds_b = smem + slot * STAGE + A_BYTES + wid * 1024;
    const int soff_a = kbase * ES;
    const int soff_w = __builtin_amdgcn_readfirstlane(
        (int)(((long)s_tap * (p.k0 + p.k1) + (src ? p.k0 : 0) + kbase) * ES));
    auto issue = [&](const __amdgpu_buffer_rsrc_t& ra, const int (&va)[AR], auto has_tail) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int v = va[i];
        if constexpr (decltype(has_tail)::value) { if (kbase + a_chunk[i] * EPC >= ksrc) v = INV; }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(lds_a + i * (NT * 16)), 16, v, soff_a, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        int v = vb[i];
        if constexpr (decltype(has_tail)::value) { if (kbase + b_chunk[i] * EPC >= ksrc) v = INV; }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lds_b + i * (NT * 16)), 16, v, soff_w, 0, 0);
      }
    };
    if (!tail) {
      if (!src) issue(ra0, va0, std::false_type{}); else issue(ra1, va1, std::false_type{});
    } else {
      if (!src) issue(ra0, va0, std::true_type{}); else issue(ra1, va1, std::true_type{});
    }
    if (++s_kc == (s_src ? p.nk1 : p.nk0)) {
      s_kc = 0;
      if (s_src == 0 && p.nk1 > 0) s_src = 1;
      else { s_src = 0; ++s_tap; tap_dirty = true; }
    }
  };

  const int wr = wid / WGN, wc = wid - wr * WGN;
  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, hi = lane >> 5;
  int a_lds_row[FM], b_lds_row[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) a_lds_row[i] = wr * WTM + i * 32 + l31;
#pragma unroll
  for (int j = 0; j < FN; ++j) b_lds_row[j] = wc * WTN + j * 32 + l31;

  constexpr int KS = CPR / 2;                 // fragment reads per tile row (one chunk per lane half): MFMA k sub-steps (even)
  static_assert(KS % 2 == 0, "the pipelined loop reads fragment sets in pairs");
  // Fragment reads are raw ds_read_b128 (common.h): with compiler-visible LDS loads hipcc puts `s_waitcnt vmcnt(0)` in
  // front of the first read of every K step -- AFTER the next tile's DMA has been issued -- so load latency and MFMA time
  // add up instead of overlapping (1.1-1.5 us per 64-deep step, measured).  The reads of one fragment set are retired
  // by lds_wait<reads issued after them>() before the MFMAs that consume them.
  constexpr int NF = FM + FN;                 // ds_read_b128 per fragment set
  const unsigned lds_base = lds_addr(smem);
  auto read_frags = [&](unsigned sa, int ks, raw_u32x4_t (&af)[FM], raw_u32x4_t (&bf)[FN]) {
    const unsigned sb = sa + A_BYTES;
    const int chunk = ks * 2 + hi;
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = lds_read16_raw(sa + tile_off<CPR>(a_lds_row[i], chunk));
#pragma unroll
    for (int j = 0; j < FN; ++j) bf[j] = lds_read16_raw(sb + tile_off<CPR>(b_lds_row[j], chunk));
  };
  constexpr int NLN = LN == 1 ? FM : (LN == 2 ? FN : 1);
  float ln_s[NLN], ln_q[NLN];
#pragma unroll
  for (int i = 0; i < NLN; ++i) ln_s[i] = ln_q[i] = 0.f;
  constexpr bool LN_SHIFT = LN != 0 && std::is_same<Tag, f32_tag>::value;     // see ln_stat_shifted
  float ln_c[LN_SHIFT ? NLN : 1];
  if constexpr (LN_SHIFT) {
#pragma unroll
    for (int i = 0; i < NLN; ++i) {
      const int row = LN == 1 ? m0 + a_lds_row[i] : n0 + b_lds_row[i];
      const bool ok = row < (LN == 1 ? p.m : p.n);
      ln_c[i] = ok ? *(const float*)((LN == 1 ? p.a0 : p.w) + (long)row * (LN == 1 ? p.lda0 : p.ldw) * 4) : 0.f;
    }
  }
  // TT_F32 "split16" (round 6): fp32-class products at the 16-bit matrix rate.  Every fp32 operand x is split on the fly into two fp16
  // parts on DIFFERENT binary scales,
  //     h = fp16(x 2^-8),   l = fp16((x - 2^8 h) 2^3)          x = 2^8 h + 2^-3 l  up to  max(2^-22 |x|, 2^-28)
  // so that |x| up to 2^24 fits (an unscaled fp16 hi part overflows at 65504: proj_in of the tiny test model sees 1.8e5, and inf x 0
  // is NaN) while the lo part -- which also repairs the denormal quantisation of h for |x| < 2^-6 -- stays a normal fp16 number for
  // every x whose h is normal (fp16 denormals are KEPT by the gfx950 MFMA: tools/mfma_split_probe.hip).  Then
  //     a b = 2^16 a_h b_h + 2^5 (a_h b_l + a_l b_h) + 2^-6 a_l b_l (dropped: 2^-22 of the product)
  // runs as three v_mfma_f32_32x32x16_f16 per product block into TWO fp32 accumulators (the hi x hi sum and the cross-term sum carry
  // different scales), combined once after the K loop.  One 16-deep MFMA needs 8 k-values per lane: two consecutive fragment sets
  // (4 fp32 each) -- the loops below consume the sets in pairs already, so the even set is split and parked in registers (phase 0)
  // and the odd set triggers the MFMAs (phase 1).  Both operands use the same (lane half, slot) -> k map, which is all the instruction
  // needs.  Per 32 x 32 x 16 block: 3 x 8 passes against 8 x 16 passes of v_mfma_f32_32x32x2_f32.
  uint2 sp_ah[SPLIT ? FM : 1], sp_al[SPLIT ? FM : 1], sp_bh[SPLIT ? FN : 1], sp_bl[SPLIT ? FN : 1];
  f32x16_t accx[SPLIT ? FM : 1][SPLIT ? FN : 1];        // cross terms a_h b_l + a_l b_h
  if constexpr (SPLIT) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[i][j][r] = 0.f;
  }
  // conversion of one fragment set (4 fp32 values): 10 VALU -- two packed scalings, two packed conversions for h, and one
  // v_fma_mixlo / mixhi_f16 per value for l = fp16(8 x - 2048 h): the instruction widens its fp16 operand, fuses and rounds once.
  // An operand the caller pre-split (GemmP.presplit: packed weights) arrives as (h01, h23, l01, l23) already: no VALU at all.
  const float kM2048 = -2048.0f;
  auto split4 = [&](const raw_u32x4_t& f, bool pre, uint2& h, uint2& l) {
    if (pre) { h = make_uint2(f.x, f.y); l = make_uint2(f.z, f.w); return; }
    const f32x2_t v01 = (f32x2_t){__uint_as_float(f.x), __uint_as_float(f.y)}, v23 = (f32x2_t){__uint_as_float(f.z), __uint_as_float(f.w)};
    const f16x2_t h01 = __builtin_convertvector(v01 * 0.00390625f, f16x2_t), h23 = __builtin_convertvector(v23 * 0.00390625f, f16x2_t);   // fp16(x 2^-8)
    h = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    const f32x2_t e01 = v01 * 8.0f, e23 = v23 * 8.0f;                      // (x - 2^8 h) 2^3 = 8 x - 2048 h, exact in fp32
    unsigned l01, l23;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h.x), "v"(kM2048), "v"(e01.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h.x), "v"(kM2048), "v"(e01.y));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h.y), "v"(kM2048), "v"(e23.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h.y), "v"(kM2048), "v"(e23.y));
    l = make_uint2(l01, l23);
  };
#if defined(TT_SPLIT_ABL) && (TT_SPLIT_ABL & 1)      // ablation build (timing only, wrong numbers): no conversions at all
  constexpr bool pre_a = true, pre_b = true;
#else
  constexpr bool pre_a = PRE_A, pre_b = PRE_B;         // compile-time: a branch would cut the block the scheduler interleaves
#endif
  // Software pipeline of the split products (TT_SPLIT_PIPE, default on): the MFMAs of chunk pair p are ISSUED during the conversion of the
  // even chunk of pair p + 1 -- an in-order wave can only overlap its own VALU with its own MFMAs if they alternate in program order (the
  // ablation of 6.R6: conversions and the two extra MFMAs cost 9 ms each alone and 40 ms together when they run back to back).  The
  // operands of the pending pair wait in op_* (32 registers for a 64 x 64 wave tile; the pair's temporaries before); the first phase 0 of
  // a launch issues MFMAs on zero operands (adds nothing), the last pair is flushed after the K loop.
#ifndef TT_SPLIT_PIPE
#define TT_SPLIT_PIPE 1
#endif
  // block rows of the pending pair issued under the EVEN chunk's conversion; the rest would go under the odd chunk's.  Default: all of
  // them under the even chunk -- spreading them (1 of 2 rows) keeps the odd chunk's halves in temporaries next to the pending operands
  // and spills 40-88 bytes per lane on the 64 x 64 wave tiles: 93.4 -> 104.2 ms / step (one call, interleaved).
#ifndef TT_SPLIT_PIPE_ROWS
#define TT_SPLIT_PIPE_ROWS 99
#endif
  constexpr int PIPE_SPLIT_ROWS = FM > 1 ? (TT_SPLIT_PIPE_ROWS < FM ? TT_SPLIT_PIPE_ROWS : FM) : FM;
  uint4 op_ah[SPLIT ? FM : 1], op_al[SPLIT ? FM : 1], op_bh[SPLIT ? FN : 1], op_bl[SPLIT ? FN : 1];
  if constexpr (SPLIT) {
#pragma unroll
    for (int i = 0; i < FM; ++i) op_ah[i] = op_al[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) op_bh[j] = op_bl[j] = make_uint4(0, 0, 0, 0);
  }
  // the same conversion in plain C (14 VALU, no inline asm: every instruction is visible to the scheduler, which is asked to alternate
  // them with the pending MFMAs); used where the conversion is hidden under MFMAs anyway
  auto split4c = [&](const raw_u32x4_t& f, bool pre, uint2& h, uint2& l) {
    if (pre) { h = make_uint2(f.x, f.y); l = make_uint2(f.z, f.w); return; }
    const f32x2_t v01 = (f32x2_t){__uint_as_float(f.x), __uint_as_float(f.y)}, v23 = (f32x2_t){__uint_as_float(f.z), __uint_as_float(f.w)};
    const f16x2_t h01 = __builtin_convertvector(v01 * 0.00390625f, f16x2_t), h23 = __builtin_convertvector(v23 * 0.00390625f, f16x2_t);
    h = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    const f32x2_t r01 = v01 * 8.0f - __builtin_convertvector(h01, f32x2_t) * 2048.0f, r23 = v23 * 8.0f - __builtin_convertvector(h23, f32x2_t) * 2048.0f;
    l = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(r01, f16x2_t)), __builtin_bit_cast(unsigned, __builtin_convertvector(r23, f16x2_t)));
  };
  auto issue_pending = [&](auto lo_tag, auto hi_tag) {               // the three MFMA sweeps of the pair whose operands wait in op_*, block rows [LO, HI)
   constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
   if constexpr (SPLIT) {
#if !defined(TT_SPLIT_ABL) || !(TT_SPLIT_ABL & 2)
#pragma unroll
    for (int i = LO; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) accx[i][j] = Cvt<f16_tag>::mfma32(op_bl[j], op_ah[i], accx[i][j]);
#endif
#pragma unroll
    for (int i = LO; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = Cvt<f16_tag>::mfma32(op_bh[j], op_ah[i], acc[i][j]);
#if !defined(TT_SPLIT_ABL) || !(TT_SPLIT_ABL & 2)
#pragma unroll
    for (int i = LO; i < HI; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) accx[i][j] = Cvt<f16_tag>::mfma32(op_bh[j], op_al[i], accx[i][j]);
#endif
   }
  };
  auto mma = [&](const raw_u32x4_t (&af)[FM], const raw_u32x4_t (&bf)[FN], int phase) {
    if constexpr (LN == 1) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if constexpr (LN_SHIFT) ln_stat_shifted(af[i], ln_c[i], ln_s[i], ln_q[i]); else ln_stat<Tag>(af[i], ln_s[i], ln_q[i]);
      }
    } else if constexpr (LN == 2) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr (LN_SHIFT) ln_stat_shifted(bf[j], ln_c[j], ln_s[j], ln_q[j]); else ln_stat<Tag>(bf[j], ln_s[j], ln_q[j]);
      }
    }
    if constexpr (SPLIT && TT_SPLIT_PIPE) {
      if (phase == 0) {                      // (constant after unrolling) even chunk: convert it UNDER the MFMAs of the previous pair
#pragma unroll
        for (int i = 0; i < FM; ++i) split4c(af[i], pre_a, sp_ah[i], sp_al[i]);
#pragma unroll
        for (int j = 0; j < FN; ++j) split4c(bf[j], pre_b, sp_bh[j], sp_bl[j]);
        issue_pending(std::integral_constant<int, 0>{}, std::integral_constant<int, PIPE_SPLIT_ROWS>{});
        // ask for MFMA, VALU, VALU, ... (one MFMA holds the matrix pipe for 8 passes; two or three conversions fit under it)
#pragma unroll
        for (int g = 0; g < 3 * PIPE_SPLIT_ROWS * FN; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
      } else {                               // odd chunk: convert (under the rest of the pending MFMAs), combine with the parked even half
        uint2 th[FM], tl[FM], ubh[FN], ubl[FN];
        constexpr bool HIDDEN = PIPE_SPLIT_ROWS < FM;          // MFMAs left to hide the conversion under: the plain-C form; else the 10-VALU asm form
#pragma unroll
        for (int i = 0; i < FM; ++i) { if constexpr (HIDDEN) split4c(af[i], pre_a, th[i], tl[i]); else split4(af[i], pre_a, th[i], tl[i]); }
#pragma unroll
        for (int j = 0; j < FN; ++j) { if constexpr (HIDDEN) split4c(bf[j], pre_b, ubh[j], ubl[j]); else split4(bf[j], pre_b, ubh[j], ubl[j]); }
        issue_pending(std::integral_constant<int, PIPE_SPLIT_ROWS>{}, std::integral_constant<int, FM>{});
#pragma unroll
        for (int g = 0; g < 3 * (FM - PIPE_SPLIT_ROWS) * FN; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          op_ah[i] = make_uint4(sp_ah[i].x, sp_ah[i].y, th[i].x, th[i].y); op_al[i] = make_uint4(sp_al[i].x, sp_al[i].y, tl[i].x, tl[i].y);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          op_bh[j] = make_uint4(sp_bh[j].x, sp_bh[j].y, ubh[j].x, ubh[j].y); op_bl[j] = make_uint4(sp_bl[j].x, sp_bl[j].y, ubl[j].x, ubl[j].y);
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // conversions and MFMAs stay in front of the next raw fragment read
    } else if constexpr (SPLIT) {
      if (phase == 0) {                      // (constant after unrolling)
#pragma unroll
        for (int i = 0; i < FM; ++i) split4(af[i], pre_a, sp_ah[i], sp_al[i]);
#pragma unroll
        for (int j = 0; j < FN; ++j) split4(bf[j], pre_b, sp_bh[j], sp_bl[j]);
      } else {
        uint4 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          uint2 h, l;
          split4(af[i], pre_a, h, l);
          ah[i] = make_uint4(sp_ah[i].x, sp_ah[i].y, h.x, h.y); al[i] = make_uint4(sp_al[i].x, sp_al[i].y, l.x, l.y);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          uint2 h, l;
          split4(bf[j], pre_b, h, l);
          bh[j] = make_uint4(sp_bh[j].x, sp_bh[j].y, h.x, h.y); bl[j] = make_uint4(sp_bl[j].x, sp_bl[j].y, l.x, l.y);
        }
        // three sweeps over the blocks, so that the two MFMAs into one cross-term accumulator sit FM x FN x 2 instructions apart
        // (back to back they wait for each other's result)
#if !defined(TT_SPLIT_ABL) || !(TT_SPLIT_ABL & 2)      // ablation bit 1: one MFMA per product block instead of three
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) accx[i][j] = Cvt<f16_tag>::mfma32(bl[j], ah[i], accx[i][j]);
#endif
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = Cvt<f16_tag>::mfma32(bh[j], ah[i], acc[i][j]);
#if !defined(TT_SPLIT_ABL) || !(TT_SPLIT_ABL & 2)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) accx[i][j] = Cvt<f16_tag>::mfma32(bh[j], al[i], accx[i][j]);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);     // the conversions read raw-asm fragment registers: same pinning as the statistics below
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = Cvt<Tag>::mfma32(make_uint4(bf[j].x, bf[j].y, bf[j].z, bf[j].w), make_uint4(af[i].x, af[i].y, af[i].z, af[i].w), acc[i][j]);
    }
    // The fragment registers are written by raw asm ds_reads whose data lands later (cdna guide 5.7 item 1).  If the
    // scheduler sinks the statistics VALU below the NEXT raw read of the same fragment set, the two values are live at once,
    // the new read gets other registers and a v_mov copies them back at the loop edge -- before the data has landed
    // (NaNs under load, measured).  Pin everything that reads the fragments in front of whatever follows.
    if constexpr (LN != 0) __builtin_amdgcn_sched_barrier(0);
  };

  // residual operands of every (row, quad) this lane will finish (epilogue layout, see below): ONE batch of loads.
  // Tiles with register headroom issue it BEFORE the main loop so the whole K loop hides the latency; the 256-row
  // tiles (128-VGPR budget) issue it at the top of the epilogue, where it overlaps the barrier and the LDS transposition.
  // (The AlphaBlender source is usually the residual tensor itself -- temporal ResBlock -- and then shares the
  // preloaded value; a distinct blend tensor is read in-pass.)
  constexpr int NCH = (FN + 1) / 2;                    // 64-column chunks per fragment row
  constexpr bool EARLY_RES = BM <= 128 && !SPLIT;     // (the split variant needs the 64 registers for its operand halves)
  quad_t resv[EARLY_RES ? FM : 1][EARLY_RES ? NCH : 1][8];   // 256-row tiles read the residual inside the passes
  const bool direct = (p.out_col_hw > 0 || p.out_f32) && p.splitk == 1;   // rare layouts keep the simple per-fragment path
  const bool blend_is_res = p.blend && p.blend == p.residual && p.ld_blend == p.ld_res;
  auto preload_residual = [&]() {
    if constexpr (EARLY_RES) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int pass = 0; pass < 8; ++pass) resv[i][c][pass] = zero_quad<Tag>();
    }
    // (no residual: nothing is requested -- loads through a 0-byte descriptor return 0, but still cost their issue slots in the
    // texture addresser: 0.5 us per launch for two resident 128 x 128 tiles, tools/gemm_timeline.py)
    if constexpr (EARLY_RES) if (!direct && !p.geglu && p.splitk == 1 && p.residual) {
      const __amdgpu_buffer_rsrc_t r_res = make_rsrc(p.residual, p.res_bytes);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int nfr = (2 * c + 1 < FN) ? 2 : 1;
          const int q_per_row = nfr * 8, rows_per_pass = 64 / q_per_row;
          const int gn = n0 + wc * WTN + c * 64 + (lane % q_per_row) * 4;
#pragma unroll
          for (int pass = 0; pass < 8; ++pass) {
            resv[i][c][pass] = zero_quad<Tag>();
            if (pass * rows_per_pass < 32) {
              const int gm = m0 + wr * WTM + i * 32 + pass * rows_per_pass + lane / q_per_row;
              resv[i][c][pass] = ldq<Tag>(r_res, (gm < p.m && gn < p.n) ? (int)(((long)gm * p.ld_res + gn) * ES) : kInv);
            }
          }
        }
    }
  };
  if constexpr (NST == 2) {
    // plain double buffer: wait tile kt, barrier, issue tile kt+1, compute tile kt.  The first tile is requested before
    // the residual batch (whose address arithmetic costs ~2 us); the first wait covers both.
    TL(6);
    stage(0);
    TL(7);
    // (requested behind the SECOND tile instead -- under the MFMAs of the first K step -- the batch disturbs the K loop's DMA stream:
    // 12544 x 640 x 640 + residual 18.7 -> 19.3 us, K = 2560 49.3 -> 51.9 us, step 30.67 -> 30.72 ms, one call; round 5)
    preload_residual();
    TL(1);
    for (int kt = 0; kt < KT; ++kt) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef TT_GEMM_TIMELINE
      if (kt == 0) TL(2);
#endif
      if (kt + 1 < KT) stage((kt + 1) & 1);
      const unsigned sa = lds_base + (kt & 1) * STAGE;
      raw_u32x4_t af[2][FM], bf[2][FN];         // fragment sets double-buffered: set ks+1 is read under the MFMAs of ks
      read_frags(sa, 0, af[0], bf[0]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
          read_frags(sa, ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
          lds_wait<NF>();
        } else {
          lds_wait<0>();
        }
        mma(af[ks & 1], bf[ks & 1], ks & 1);
      }
    }
  } else {
    TL(6);
    preload_residual();                       // before the ring fill: the counted waits below assume the DMAs come last
    TL(1);
    // software pipeline: fragments double-buffered in registers; the wait+barrier for tile kt+1 sits BEFORE the last
    // MFMA group of tile kt, so the first fragments of tile kt+1 are read (and tile kt+NST-1 is issued) under MFMAs.
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < KT) stage(s);
    wait_tiles<G>(min(KT - 1, NST - 2));      // tile 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    raw_u32x4_t afA[FM], bfA[FN], afB[FM], bfB[FN];
    read_frags(lds_base, 0, afA, bfA);
    int slot = 0, fill = NST - 1;             // slot of tile kt ; slot the next staged tile goes to
    // The last tile is peeled: inside the steady loop set A is (re)defined by the raw read at ONE program point on every
    // path, so no value merge of "old fragments / new fragments" exists and hipcc has no reason to copy raw-read registers
    // (a copy would run before the data has landed: cdna guide 5.7 item 1).
    auto tile = [&](int kt, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      const unsigned sa = lds_base + slot * STAGE;
      const int nslot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        read_frags(sa, ks + 1, afB, bfB);
        lds_wait<NF>();                       // set A (issued before set B) has landed
        mma(afA, bfA, 0);
        if (ks + 2 < KS) {
          read_frags(sa, ks + 2, afA, bfA);
          lds_wait<NF>();                     // set B has landed
        } else if constexpr (!LAST) {
          // tile kt+1 must have landed; tiles kt+2 .. min(KT-1, kt+NST-2) may stay in flight
          wait_tiles<G>(min(KT - 2 - kt, NST - 3));
          __builtin_amdgcn_s_barrier();       // all waves: tile kt+1 visible, tile kt-1's slot free
          asm volatile("" ::: "memory");
          if (kt + NST - 1 < KT) stage(fill);
          read_frags(lds_base + nslot * STAGE, 0, afA, bfA);
          lds_wait<NF>();
        } else {
          lds_wait<0>();
        }
        mma(afB, bfB, 1);
      }
      slot = nslot;
      fill = fill + 1 == NST ? 0 : fill + 1;
    };
    for (int kt = 0; kt + 1 < KT; ++kt) tile(kt, std::false_type{});
    tile(KT - 1, std::true_type{});
  }

  if constexpr (SPLIT && TT_SPLIT_PIPE) issue_pending(std::integral_constant<int, 0>{}, std::integral_constant<int, FM>{});       // the last chunk pair's MFMAs
  if constexpr (SPLIT) {                               // 2^16 (hi x hi) + 2^5 (cross terms)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(accx[i][j][r], 32.0f, acc[i][j][r] * 65536.0f);
  }
  // ---- fused LayerNorm: 1/sigma of the operand rows from the sums gathered beside the MFMAs (the two lane halves hold the
  // even / odd chunks of a row).  Rows: scale the accumulators now (lane <-> row l31).  Columns: the lane that owns W row
  // l31 of fragment j publishes 1/sigma of output column j*32 + l31 in a wave-private LDS array behind the ring; the
  // epilogue multiplies by it where a lane holds four consecutive columns.
  constexpr int CS_OFF = NST * STAGE;                  // MODE 4 only: WGM*WGN KiB more dynamic LDS (launch_mode)
  constexpr int STAT_OFF = NST * STAGE + (KMODE == 4 ? WGM * WGN * 1024 : 0);    // GemmP.stats: 8 * BN * WGM bytes more (launch_mode)
  if constexpr (LN != 0) {
    const float inv_k = 1.0f / (float)p.k0;
    float rs[NLN];
#pragma unroll
    for (int i = 0; i < NLN; ++i) {
      const float sm = (ln_s[i] + __shfl_xor(ln_s[i], 32)) * inv_k, sq = (ln_q[i] + __shfl_xor(ln_q[i], 32)) * inv_k;
      rs[i] = rsqrtf(fmaxf(sq - sm * sm, 0.f) + p.ln_eps);
    }
    if constexpr (LN == 1) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= rs[i];
    } else {
      float* cs = (float*)(smem + CS_OFF + wid * 1024);
#pragma unroll
      for (int j = 0; j < FN; ++j) cs[j * 32 + l31] = rs[j];          // both lane halves write the same value
    }
  }
  const float* colscale = (const float*)(smem + CS_OFF + wid * 1024);   // read only when LN == 2

  // ---- epilogue.  The MFMA layout gives a lane row m = .. + l31 and columns n = .. + 8g + 4hi + {0..3}: stored
  // directly, one instruction would touch 32 rows x 16 bytes.  Instead each wave transposes its accumulators through a
  // private LDS strip (fp32, 32 rows x 64 columns at a time) and re-reads them so that 16 consecutive lanes cover 64
  // consecutive columns of one row: every residual/blend load and every store instruction then covers 4 rows x 128
  // contiguous bytes.  All global accesses go through bounds-checked descriptors (see make_rsrc) so the pass loops are
  // straight-line code; the arithmetic (fp32, same order) is what epilogue_quad does.
  TL(3);
  if (!direct) {
    const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(p.bias, p.bias_bytes);
    const __amdgpu_buffer_rsrc_t r_out = p.splitk > 1 ? make_rsrc(p.ws, p.ws_bytes) : make_rsrc(p.out, p.out_bytes);
    // strip = 32 rows x 256 bytes per wave, 16-byte quads XOR-swizzled by the row (no padding: 8 KiB per wave)
    auto strip_off = [](int row, int quad, int nq) { return row * 256 + ((quad ^ (row & (nq - 1))) << 4); };
    char* ebuf = smem + wid * 8192;
    static_assert(WGM * WGN * 8192 <= NST * STAGE, "epilogue strips do not fit the ring");
    if (!p.geglu) {
      // bias of the 4 columns this lane finishes in each 64-column chunk (row-independent: loaded once)
      float4 bias4[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int q_per_row = ((2 * c + 1 < FN) ? 2 : 1) * 8;
        const int gn = n0 + wc * WTN + c * 64 + (lane % q_per_row) * 4;
        bias4[c] = ld128f(r_bias, gn < p.n ? gn * 4 : kInv);
      }
      const float alpha = p.blend ? p.alpha : 0.0f, one_m_alpha = 1.0f - alpha;
      __syncthreads();                                 // all waves are done with the operand ring
      TL(4);
      // Operand variants (uniform dispatch, each straight-line):
      //   FILM   the row vector (time-embedding FiLM term, one vector per rowvec_rows output rows) of the <= 2 row groups
      //          a 32-row fragment spans is loaded up front and selected per row;
      //   INPASS operands that cannot be held in registers are loaded inside the pass batches: a blend tensor distinct
      //          from the residual, a row vector with groups shorter than 32 rows, and -- on the 256-row tiles, whose
      //          128-VGPR budget has no room for the preload -- the residual.  A load issued after a store waits for
      //          that store (in-order vmcnt), so each batch loads first and stores last; the planner keeps such
      //          epilogues off the 256-row tiles.
      auto run = [&](auto film_tag, auto inpass_tag, auto out8_tag, auto stats_tag) {
        constexpr bool FILM = decltype(film_tag)::value, INPASS = decltype(inpass_tag)::value, OUT8 = decltype(out8_tag)::value;
        // GemmP.stats, compile-time like the operand variants (a run-time `if` inside the pass loops breaks the straight-line code)
        constexpr bool STATS = decltype(stats_tag)::value;
        const __amdgpu_buffer_rsrc_t r_rv = make_rsrc(p.rowvec, (FILM || INPASS) ? p.rowvec_bytes : 0);
        const __amdgpu_buffer_rsrc_t r_bl = make_rsrc(p.blend, INPASS ? p.blend_bytes : 0);
        const __amdgpu_buffer_rsrc_t r_res = make_rsrc(p.residual, (INPASS && !EARLY_RES) ? p.res_bytes : 0);
        const int rv_rows = p.rowvec ? p.rowvec_rows : 1;
        // periodic row vector (rowvec_mod): group indices wrap; the even / odd form (one row per group, period 2) stays on the
        // two-vector path below with the row's parity as the selector
        const bool parity = FILM && p.rowvec_mod == 2 && rv_rows == 1;
        auto wrap = [&](int g) { return p.rowvec_mod > 0 ? g - fdiv(g, p.fd_rv_mod) * p.rowvec_mod : g; };
        float* sstage = (float*)(smem + STAT_OFF) + wid * (2 * WTN);      // staging rows behind the ring (launch_mode adds the bytes)
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int mb = m0 + wr * WTM + i * 32;
          const int grp_raw = fdiv(mb, p.fd_rv_rows);    // row group of the fragment's first row (uniform)
          const int grp0 = parity ? 0 : wrap(grp_raw), grp1 = parity ? 1 : wrap(grp_raw + 1);
          const int grp_split = parity ? mb : (grp_raw + 1) * rv_rows;    // first row of the next group (parity: see the selector)
          const int sel_split = parity ? 0x7fffffff : grp_split, sel_odd = parity ? 1 : 0;
#pragma unroll
          for (int jc = 0; jc < FN; jc += 2) {
            const int nfr = (jc + 1 < FN) ? 2 : 1;       // fragments in this chunk (compile-time after unrolling)
            const int q_per_row = nfr * 8;               // 4-column quads per strip row
            const int rows_per_pass = 64 / q_per_row;
            const int qq = lane % q_per_row, rr = lane / q_per_row;
            const int gn = n0 + wc * WTN + jc * 32 + qq * 4;
            const float4 b4 = bias4[jc / 2];
            float4 cs4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (LN == 2) cs4 = *(const float4*)(colscale + jc * 32 + qq * 4);
            float4 film_lo = make_float4(0.f, 0.f, 0.f, 0.f), film_hi = film_lo;
            if constexpr (FILM) {
              film_lo = ld128f(r_rv, (mb < p.m && gn < p.n) ? (int)(((long)grp0 * p.ld_rowvec + gn) * 4) : kInv);
              film_hi = ld128f(r_rv, (grp_split < p.m && gn < p.n) ? (int)(((long)grp1 * p.ld_rowvec + gn) * 4) : kInv);
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              if (jc + jj < FN) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                  *(float4*)(ebuf + strip_off(l31, jj * 8 + 2 * g + hi, nfr * 8)) =
                      make_float4(acc[i][jc + jj][g * 4], acc[i][jc + jj][g * 4 + 1], acc[i][jc + jj][g * 4 + 2], acc[i][jc + jj][g * 4 + 3]);
              }
            }
            constexpr int PB = INPASS ? (BM > 128 ? 2 : 4) : 8;   // passes per batch, sized to the register budget
#pragma unroll
            for (int pb = 0; pb < 8; pb += PB) {
              quad_t rqv[PB], blv[PB];
              float4 rvv[PB];
#pragma unroll
              for (int k = 0; k < PB; ++k) {
                const int pass = pb + k;
                rqv[k] = blv[k] = zero_quad<Tag>();
                rvv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pass * rows_per_pass < 32) {
                  const int gm = mb + pass * rows_per_pass + rr;
                  const bool ok = gm < p.m && gn < p.n;
                  if constexpr (EARLY_RES) rqv[k] = resv[i][jc / 2][pass];
                  else if constexpr (INPASS) rqv[k] = ldq<Tag>(r_res, ok ? (int)(((long)gm * p.ld_res + gn) * ES) : kInv);
                  if constexpr (INPASS) {
                    rvv[k] = ld128f(r_rv, ok ? (int)(((long)wrap(fdiv(gm, p.fd_rv_rows)) * p.ld_rowvec + gn) * 4) : kInv);
                    blv[k] = ldq<Tag>(r_bl, ok ? (int)(((long)gm * p.ld_blend + gn) * ES) : kInv);
                  }
                }
              }
#pragma unroll
              for (int k = 0; k < PB; ++k) {
                const int pass = pb + k;
                if (pass * rows_per_pass < 32) {
                  const int r = pass * rows_per_pass + rr;
                  const float4 t = *(const float4*)(ebuf + strip_off(r, qq, q_per_row));
                  const int gm = mb + r;
                  const bool ok = gm < p.m && gn < p.n;
                  if (p.splitk > 1) {                    // uniform: fp32 partial sums of K slice `split`
                    st128f(r_out, ok ? (int)((((long)split * p.m + gm) * p.n + gn) * 4) : kInv, t);
                  } else {
                    float v[4] = {(t.x * cs4.x + b4.x) * p.acc_scale, (t.y * cs4.y + b4.y) * p.acc_scale,
                                  (t.z * cs4.z + b4.z) * p.acc_scale, (t.w * cs4.w + b4.w) * p.acc_scale};
                    const quad_t rq = rqv[k];
                    quad_t bq = rq;
                    if constexpr (FILM) {
                      const float4 f = ((gm >= sel_split) | ((gm & sel_odd) != 0)) ? film_hi : film_lo;     // branch-free (see gemm_w320.hip)
                      v[0] += f.x; v[1] += f.y; v[2] += f.z; v[3] += f.w;
                    }
                    if constexpr (INPASS) {
                      v[0] += rvv[k].x; v[1] += rvv[k].y; v[2] += rvv[k].z; v[3] += rvv[k].w;
                      bq = (p.blend && !blend_is_res) ? blv[k] : rq;
                    }
                    float r4[4], b4v[4];
                    quad_to_f32<Tag>(rq, r4);
                    quad_to_f32<Tag>(bq, b4v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = alpha * b4v[e] + one_m_alpha * (v[e] + r4[e]);
                    if constexpr (OUT8) st32(r_out, ok ? (int)((long)gm * p.ldo + gn) : kInv, pack4_fp8(v));
                    else {
                      const quad_t packed = f32_to_quad<Tag>(v);
                      if constexpr (ES == 2) st64(r_out, ok ? (int)(((long)gm * p.ldo + gn) * ES) : kInv, packed.x, packed.y);
                      else stq<Tag>(r_out, ok ? (int)(((long)gm * p.ldo + gn) * ES) : kInv, v);
                      if constexpr (STATS) {               // (rows >= m of a ragged last tile add nothing)
                        float sv[4];
                        quad_to_f32<Tag>(packed, sv);
                        const bool live = gm < p.m;
                        *(float4*)(ebuf + strip_off(r, qq, q_per_row)) = make_float4(live ? sv[0] : 0.f, live ? sv[1] : 0.f, live ? sv[2] : 0.f, live ? sv[3] : 0.f);
                      }
                    }
                  }
                }
              }
            }
            if constexpr (STATS) colstat_strip(ebuf, q_per_row, lane, sstage + jc * 32, sstage + WTN + jc * 32, i == 0);
          }
        }
      };
      const bool inpass = (p.blend && !blend_is_res) || (p.rowvec && p.rowvec_rows < 32 && !(p.rowvec_mod == 2 && p.rowvec_rows == 1)) ||
                          (!EARLY_RES && p.residual);
      constexpr std::false_type no{};
      constexpr std::true_type yes{};
      const bool kstats = p.stats && p.splitk == 1;          // (a K slice leaves the sums to the reduction kernel)
      if (inpass) run(no, yes, no, no);                      // (never with statistics: tt_gemm_stats_rows)
      else if (p.rowvec) { if (kstats) run(yes, no, no, yes); else run(yes, no, no, no); }
      else if (MODE == 0 && ES == 2 && p.out_fp8) {          // Q | K and V^T of the fp8 attention path (linear, no residual)
        if constexpr (MODE == 0 && ES == 2) run(no, no, yes, no);
      } else { if (kstats) run(no, no, no, yes); else run(no, no, no, no); }
      if (kstats) {
        __syncthreads();
        const float* st0 = (const float*)(smem + STAT_OFF);
        if (p.stat_rows == BM) {                             // one statistics tile per output tile: the waves of one tile column, wave rows in order
          for (int c = tid; c < BN; c += NT) {
            const int wcc = c / WTN, col = c - wcc * WTN;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
              const float* row = st0 + (w * WGN + wcc) * (2 * WTN);
              a += row[col]; b += row[WTN + col];
            }
            if (n0 + c < p.n) {
              p.stats[((long)tile_m * 2) * p.n + n0 + c] = a;
              p.stats[((long)tile_m * 2 + 1) * p.n + n0 + c] = b;
            }
          }
        } else {                                             // one per wave row (WTM rows; tt_gemm_stats_rows): segments that are no whole number of tiles
          for (int c = tid; c < BN * WGM; c += NT) {
            const int w = c / BN, cc = c - w * BN, wcc = cc / WTN, col = cc - wcc * WTN;
            const float* row = st0 + (w * WGN + wcc) * (2 * WTN);
            const long srow = (long)tile_m * WGM + w;
            if (n0 + cc < p.n && srow * WTM < p.m) {
              p.stats[(srow * 2) * p.n + n0 + cc] = row[col];
              p.stats[(srow * 2 + 1) * p.n + n0 + cc] = row[WTN + col];
            }
          }
        }
      }
    } else {
      // GEGLU: value/gate pairs are lane-local (regs g=0/2 value, g=1/3 gate); gelu in registers, then the 16
      // output columns of each fragment go through the strip: 2 fragments -> 32 output columns = 64 bytes per row.
      float4 bval[FN][2], bgate[FN][2];                  // bias of this lane's value / gate quads (row-independent)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int gnp = n0 + wc * WTN + j * 32 + tt * 16 + 4 * hi;      // packed column of the value quad
          bval[j][tt] = ld128f(r_bias, gnp < p.n ? gnp * 4 : kInv);
          bgate[j][tt] = ld128f(r_bias, gnp < p.n ? (gnp + 8) * 4 : kInv);
        }
      __syncthreads();                                 // all waves are done with the operand ring
      TL(4);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int mb = m0 + wr * WTM + i * 32;
#pragma unroll
        for (int jc = 0; jc < FN; jc += 2) {
          const int nfr = (jc + 1 < FN) ? 2 : 1;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            if (jc + jj < FN) {
#pragma unroll
              for (int tt = 0; tt < 2; ++tt) {
                const float4 bv = bval[jc + jj][tt], bg = bgate[jc + jj][tt];
                const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, bgv[4] = {bg.x, bg.y, bg.z, bg.w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  v[e] = (acc[i][jc + jj][(2 * tt) * 4 + e] + bvv[e]) * gelu_erf_f(acc[i][jc + jj][(2 * tt + 1) * 4 + e] + bgv[e]);
                *(float4*)(ebuf + strip_off(l31, jj * 4 + tt * 2 + hi, nfr * 4)) = make_float4(v[0], v[1], v[2], v[3]);
              }
            }
          }
          const int q_per_row = nfr * 4;               // output quads per strip row (16 output columns per fragment)
          const int rows_per_pass = 64 / q_per_row;
          const int qq = lane % q_per_row, rr = lane / q_per_row;
          const int oc = ((n0 + wc * WTN + jc * 32) >> 1) + qq * 4;     // output column
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            if (pass * rows_per_pass < 32) {
              const int r = pass * rows_per_pass + rr;
              const float4 t = *(const float4*)(ebuf + strip_off(r, qq, q_per_row));
              const int gm = mb + r;
              const float t4[4] = {t.x, t.y, t.z, t.w};
              stq<Tag>(r_out, (gm < p.m && oc * 2 < p.n) ? (int)(((long)gm * p.ldo + oc) * ES) : kInv, t4);
            }
          }
        }
      }
    }
#ifdef TT_GEMM_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);
    TL(5);
#endif
    return;
  }
  // direct path: fp32 output and the padded transposed output (never combined with split-K: see tt_gemm)
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int gm = m0 + wr * WTM + i * 32 + l31;
    if (gm >= p.m) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nb = n0 + wc * WTN + j * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int gn = nb + 8 * g + 4 * hi;
        if (gn >= p.n) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e];
        if constexpr (LN == 2) {
          const float4 c4 = *(const float4*)(colscale + j * 32 + 8 * g + 4 * hi);
          v[0] *= c4.x; v[1] *= c4.y; v[2] *= c4.z; v[3] *= c4.w;
        }
        epilogue_quad<Tag>(p, gm, gn, v);
      }
    }
  }
}

// ---- (opt-in: tt_gemm_set_streaming_square / TT_GEMM_SQ320=1) square C = 320 linears of the finest UNet level (to_out /
// to_q / proj_in / proj_out at 32x56 latents: ~60 launches per step, M = 50176).  Measured: 31 us against 36 us for the
// tiled kernel in isolation, but 0.1-0.3 ms per step SLOWER inside the two-branch graph (one 160 KiB block per CU leaves
// no room for the other branch's kernels), hence off by default.  The tiled kernel above needs 1.5 rounds of lock-stepped blocks for them and re-reads W from L2
// for every tile (36-50 us against ~20 us for their 64-96 MB at HBM speed).  Here W never touches LDS: each of the 10
// waves of a block keeps its 32 output columns x 320 K of W in registers (20 MFMA operands = 80 VGPRs) for the whole
// launch, blocks are persistent over 32-row tiles of A, and LDS holds only a deep ring of A (and residual) tiles filled
// by LDS-DMA, so the launch streams A / residual / out at HBM speed with several tiles in flight per CU.
// Epilogue terms: bias, scale, residual, AlphaBlender with the residual as its source (the plain variant above).
constexpr int SQ_K = 320, SQ_N = 320, SQ_ROWS = 32, SQ_WAVES = SQ_N / 32, SQ_NT = 64 * SQ_WAVES;
constexpr int SQ_CPR = SQ_K / 8;                               // 16-byte chunks per tile row (40)
constexpr int SQ_TILE_BYTES = SQ_ROWS * SQ_K * 2;              // 20 KiB: one A (or residual) tile
constexpr int SQ_PASSES = SQ_ROWS * SQ_CPR / SQ_NT;            // DMA instructions per thread per tile (2)
static_assert(SQ_ROWS * SQ_CPR % SQ_NT == 0 && SQ_CPR % 8 == 0, "sq320 staging");
__device__ __forceinline__ int sq_swz(int row) { return (row >> 1) & 7; }     // 640-byte rows: see tile_swz
__device__ __forceinline__ int sq_off(int row, int chunk) { return (row * SQ_CPR + (chunk ^ sq_swz(row))) << 4; }

// HAS_RV (round 6): a row vector with at most TWO distinct rows over the launch -- the even / odd form (rowvec_rows = 1, rowvec_mod = 2: the temporal
// block's output projection) or two groups of rowvec_rows rows (a multiple of 32: the zero-context bias per CFG half) -- both rows are loaded ONCE
// before the tile loop (a load inside it would sit in the counted vmcnt stream of the DMA ring) and selected per row by arithmetic.
template <typename Tag, bool HAS_RES, bool HAS_RV = false>
__global__ __launch_bounds__(SQ_NT) void sq320_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = HAS_RES ? 2 * SQ_PASSES : SQ_PASSES;       // DMA instructions per thread per ring slot
  constexpr int SLOT = HAS_RES ? 2 * SQ_TILE_BYTES : SQ_TILE_BYTES;
  constexpr int NST = HAS_RES ? 3 : 5;                         // ring depth: 120 / 100 KiB + 40 KiB of strips
  constexpr int KS = SQ_K / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int ntiles = (p.m + SQ_ROWS - 1) / SQ_ROWS;
  const int my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.a0, p.a0_bytes);
  const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.residual, p.res_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  // ---- W operands of this wave's 32 columns: lane (column l31, k-half hi) holds k = ks*16 + hi*8 .. +8
  uint4 wf[KS];
  {
    const int voff = (int)(((long)(wid * 32 + l31) * p.ldw + hi * 8) * 2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rw, voff, ks * 32, 0);
      wf[ks] = make_uint4(v.x, v.y, v.z, v.w);
    }
  }
  const int qq = lane & 7, rr_ = lane >> 3;                    // epilogue layout: 8 quads per 32-column row, 8 rows per pass
  const int gn = wid * 32 + qq * 4;
  const float4 b4 = ld128f(make_rsrc(p.bias, p.bias_bytes), gn * 4);
  float4 rv0 = make_float4(0.f, 0.f, 0.f, 0.f), rv1 = rv0;
  const bool rv_parity = HAS_RV && p.rowvec_mod == 2;           // row r takes rv[r & 1]; otherwise rv[r >= rowvec_rows]
  if constexpr (HAS_RV) {
    const __amdgpu_buffer_rsrc_t r_rv = make_rsrc(p.rowvec, p.rowvec_bytes);
    rv0 = ld128f(r_rv, gn * 4);
    rv1 = ld128f(r_rv, (int)((p.ld_rowvec + gn) * 4));           // (one group only: beyond the descriptor -> zeros, never selected)
  }
  const float alpha = p.blend ? p.alpha : 0.0f, one_m_alpha = 1.0f - alpha;
  const __amdgpu_buffer_rsrc_t r_out = make_rsrc(p.out, p.out_bytes);

  // per-thread source offsets inside a tile (the tile's first row is a scalar offset)
  int voa[SQ_PASSES], vor[SQ_PASSES], vrow[SQ_PASSES];
#pragma unroll
  for (int i = 0; i < SQ_PASSES; ++i) {
    const int c = i * SQ_NT + tid, row = c / SQ_CPR, ch = (c % SQ_CPR) ^ sq_swz(row);
    vrow[i] = row;
    voa[i] = (int)(((long)row * p.lda0 + ch * 8) * 2);
    vor[i] = (int)(((long)row * p.ld_res + ch * 8) * 2);
  }
  auto stage = [&](int j, int slot) {        // j-th tile of this block; beyond the end it still issues G (dropped) loads
    // straight-line on purpose: with a branch around the DMA hipcc loses count of the loads in flight and drains them
    // (vmcnt(0)) before every tile.  Past the last tile every row fails the bounds test, so nothing is fetched.
    const int m0 = ((int)blockIdx.x + j * (int)gridDim.x) * SQ_ROWS;
    char* la = smem + slot * SLOT + wid * 1024;
    const int soa = __builtin_amdgcn_readfirstlane((int)((long)m0 * p.lda0 * 2));
    const int sor = __builtin_amdgcn_readfirstlane((int)((long)m0 * p.ld_res * 2));
#pragma unroll
    for (int i = 0; i < SQ_PASSES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(la + i * (SQ_NT * 16)), 16,
                                               m0 + vrow[i] < p.m ? voa[i] : kInv, soa, 0, 0);
    if constexpr (HAS_RES) {
#pragma unroll
      for (int i = 0; i < SQ_PASSES; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (__attribute__((address_space(3))) void*)(la + SQ_TILE_BYTES + i * (SQ_NT * 16)), 16,
                                                 m0 + vrow[i] < p.m ? vor[i] : kInv, sor, 0, 0);
    }
  };
  const unsigned lds_base = lds_addr(smem);
  const unsigned strip = lds_base + NST * SLOT + wid * 4096;    // 32 rows x 128 bytes (fp32 x 32 columns), swizzled
  auto strip_off = [](int row, int quad) { return row * 128 + ((quad ^ (row & 7)) << 4); };

  // one tile: `after` = VMEM instructions this thread issued after the DMA of tile j (all of them may stay in flight)
  auto tile = [&](int j, int slot, int fill, auto after_tag) {
    constexpr int AFTER = decltype(after_tag)::value;
    wait_vmcnt<AFTER>();
    __builtin_amdgcn_s_barrier();            // tile j visible to all waves; every wave is done with tile j-1's slot
    asm volatile("" ::: "memory");
    stage(j + NST - 1, fill);
    const unsigned sa = lds_base + slot * SLOT;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A fragments in batches of 2 (raw LDS reads, see lds_read16_raw): batch b+1 is in flight under batch b's MFMAs
    constexpr int FB = 2, NB = KS / FB;
    static_assert(KS % FB == 0, "fragment batches");
    raw_u32x4_t af[2][FB];
#pragma unroll
    for (int f = 0; f < FB; ++f) af[0][f] = lds_read16_raw(sa + sq_off(l31, f * 2 + hi));
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
      if (bt + 1 < NB) {
#pragma unroll
        for (int f = 0; f < FB; ++f) af[(bt + 1) & 1][f] = lds_read16_raw(sa + sq_off(l31, ((bt + 1) * FB + f) * 2 + hi));
        lds_wait<FB>();
      } else {
        lds_wait<0>();
      }
#pragma unroll
      for (int f = 0; f < FB; ++f) {
        const raw_u32x4_t a4 = af[bt & 1][f];
        acc = Cvt<Tag>::mfma32(wf[bt * FB + f], make_uint4(a4.x, a4.y, a4.z, a4.w), acc);
      }
    }
    // epilogue of the wave's 32 x 32 block: transpose through the strip, 4 passes of 8 rows x 64 bytes.
    // The raw ds_write is invisible to hipcc's hazard recogniser: the MFMA results need their 18 wait states by hand.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int g = 0; g < 4; ++g)
      lds_write16_raw(strip + strip_off(l31, 2 * g + hi), acc[g * 4], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3]);
    const int m0 = ((int)blockIdx.x + j * (int)gridDim.x) * SQ_ROWS;
    raw_u32x4_t tq[4];
    raw_u32x2_t rq[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 8 + rr_;
      tq[pass] = lds_read16_raw(strip + strip_off(r, qq));
      rq[pass] = (raw_u32x2_t){0u, 0u};
      if constexpr (HAS_RES) rq[pass] = lds_read8_raw(sa + SQ_TILE_BYTES + sq_off(r, gn >> 3) + (gn & 7) * 2);
    }
    lds_wait<0>();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 8 + rr_;
      const float t4[4] = {__uint_as_float(tq[pass].x), __uint_as_float(tq[pass].y), __uint_as_float(tq[pass].z), __uint_as_float(tq[pass].w)};
      float v[4], r4[4];
      unpack4<Tag>(make_uint2(rq[pass].x, rq[pass].y), r4);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
      const int gm = m0 + r;
      float rv[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HAS_RV) {
        const bool second = rv_parity ? (gm & 1) != 0 : gm >= p.rowvec_rows;
        const float4 f = second ? rv1 : rv0;
        rv[0] = f.x; rv[1] = f.y; rv[2] = f.z; rv[3] = f.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = alpha * r4[e] + one_m_alpha * ((t4[e] + bb[e]) * p.acc_scale + rv[e] + r4[e]);
      st64(r_out, gm < p.m ? (int)(((long)gm * p.ldo + gn) * 2) : kInv, pack2<Tag>(v[0], v[1]), pack2<Tag>(v[2], v[3]));
    }
  };

  // ---- ring: tiles 0 .. NST-2 in flight before the loop; iteration j waits for tile j and issues tile j+NST-1.
  // VMEM instructions per iteration after its wait: G loads + 4 stores, hence the counts below.
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) stage(s, s);
  int slot = 0, fill = NST - 1;
  auto advance = [&]() { slot = slot + 1 == NST ? 0 : slot + 1; fill = fill + 1 == NST ? 0 : fill + 1; };
  int j = 0;
  // first NST-1 iterations: fewer instructions separate a tile's DMA from its use
  if (j < my_tiles) { tile(j, slot, fill, std::integral_constant<int, (NST - 2) * G>{}); advance(); ++j; }
  if (j < my_tiles) { tile(j, slot, fill, std::integral_constant<int, (NST - 3 > 0 ? NST - 3 : 0) * G + (G + 4)>{}); advance(); ++j; }
  if constexpr (NST > 3) {
    if (j < my_tiles) { tile(j, slot, fill, std::integral_constant<int, (NST - 4 > 0 ? NST - 4 : 0) * G + 2 * (G + 4)>{}); advance(); ++j; }
    if (j < my_tiles) { tile(j, slot, fill, std::integral_constant<int, (NST - 5 > 0 ? NST - 5 : 0) * G + 3 * (G + 4)>{}); advance(); ++j; }
  }
  for (; j < my_tiles; ++j) { tile(j, slot, fill, std::integral_constant<int, 4 + (NST - 2) * (G + 4)>{}); advance(); }
}

// split-K second pass: sum the fp32 slabs in a fixed order (bit-reproducible) and run the normal epilogue
template <typename Tag>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const GemmP p) {
  kernarg_touch<sizeof(GemmP)>();
  const long quads = (long)p.m * (p.n >> 2);
  const int nq = p.n >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long)gridDim.x * blockDim.x) {
    const int gm = (int)(q / nq), gn = (int)(q - (long)gm * nq) * 4;
    const EpiOps<Tag> o = epi_load<Tag>(p, gm, gn);           // (requested with the slabs: one round trip)
    float4 a = *(const float4*)(p.ws + (long)gm * p.n + gn);
    for (int s2 = 1; s2 < p.splitk; ++s2) {
      const float4 b = *(const float4*)(p.ws + ((long)s2 * p.m + gm) * p.n + gn);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    epi_finish<Tag>(p, o, gm, gn, v);
  }
}

// ... the same with the GroupNorm tile sums of the output (GemmP.stats; the split-K routes of the two coarsest UNet levels): block (row tile of
// p.stat_rows rows, chunk of 32 columns), thread (column quad, one of 32 row lanes) walks its rows (one of a 28-row image, four of a 98- or
// 112-row tile), adds what it stores; the row lanes meet in LDS in a fixed order.  The caller picks the tile height (tt_gemm_stats_rows).
template <typename Tag>
__global__ __launch_bounds__(256) void splitk_epilogue_stats_kernel(const GemmP p) {
  __shared__ float red[32][8][8];
  kernarg_touch<sizeof(GemmP)>();
  const int R = p.stat_rows, rt = blockIdx.x, tid = threadIdx.x;
  const int qd = tid & 7, rl = tid >> 3;
  const int gn = (int)blockIdx.y * 32 + qd * 4;
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  if (gn < p.n) {
    for (int r = rl; r < R; r += 32) {
      const int gm = rt * R + r;                              // < m: m is a multiple of R (tt_gemm_stats_rows)
      const EpiOps<Tag> o = epi_load<Tag>(p, gm, gn);
      float4 a = *(const float4*)(p.ws + (long)gm * p.n + gn);
      for (int s2 = 1; s2 < p.splitk; ++s2) {
        const float4 b = *(const float4*)(p.ws + ((long)s2 * p.m + gm) * p.n + gn);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      float v[4] = {a.x, a.y, a.z, a.w};
      epi_finish<Tag>(p, o, gm, gn, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float x = round_store<Tag>(v[e]); cs[e] += x; cq[e] = fmaf(x, x, cq[e]); }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[rl][qd][e] = cs[e]; red[rl][qd][4 + e] = cq[e]; }
  __syncthreads();
  if (tid < 64) {                                             // thread (column quad, one of its eight sums): the 32 row lanes in order
    const int q2 = tid >> 3, e = tid & 7, c = (int)blockIdx.y * 32 + q2 * 4 + (e & 3);
    if (c < p.n) {
      float t = red[0][q2][e];
#pragma unroll
      for (int l = 1; l < 32; ++l) t += red[l][q2][e];
      p.stats[((long)rt * 2 + (e >> 2)) * p.n + c] = t;
    }
  }
}
// ... and with the GroupNorm that reads the output (GemmP.gn_out; the coarsest UNet levels, where everything is bound by launches):
// block (segment of p.stat_rows rows = one image or the frames of a video, one of the 32 groups), thread (column quad of the group, row lane)
// finishes MAXR rows -- slabs summed in order, epilogue, `out` stored --, keeps the STORED values in registers, the block adds them up
// (fp64, fixed order: lanes by shuffle, waves through LDS), and every thread normalises what it holds.  One launch instead of the reduction
// and a GroupNorm launch; the normalised tensor is computed from the same rounded values and the same formula as tt_groupnorm_tiles.
// Row r of the segment belongs to lane r % rl_n, slot r / rl_n; the slab loads of all MAXR slots are issued before the first is used
// (slots beyond the segment re-read its last row and count for nothing: no branches around the loads).
template <typename Tag, int MAXR>
__global__ __launch_bounds__(1024) void splitk_epilogue_gn_kernel(const GemmP p, const int rl_n) {
  typedef typename Elem<Tag>::quad_t quad_t;
  constexpr int ES = Elem<Tag>::ES;
  __shared__ double wsum[16][2];
  __shared__ float s_stat[2];
  kernarg_touch<sizeof(GemmP)>();
  const int R = p.stat_rows, sg = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x;
  const int cpg = p.n >> 5, nq = cpg >> 2;
  const int rl = tid / nq, qd = tid - rl * nq;
  const int gn = grp * cpg + qd * 4;
  const bool lane_ok = rl < rl_n;
  float x[MAXR][4];
  int gmr[MAXR];
  bool ok[MAXR];
  EpiOps<Tag> eo[MAXR];
  const float4 g4 = *(const float4*)(p.gn_gamma + gn), b4 = *(const float4*)(p.gn_beta + gn);
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int r = rl + i * rl_n;
    ok[i] = lane_ok && r < R;
    gmr[i] = sg * R + (ok[i] ? r : R - 1);                    // < m: m is a multiple of R (tt_gemm_gn_fused)
    eo[i] = epi_load<Tag>(p, gmr[i], gn);
    const float4 a = *(const float4*)(p.ws + (long)gmr[i] * p.n + gn);
    x[i][0] = a.x; x[i][1] = a.y; x[i][2] = a.z; x[i][3] = a.w;
  }
  for (int s2 = 1; s2 < p.splitk; ++s2) {
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
      const float4 b = *(const float4*)(p.ws + ((long)s2 * p.m + gmr[i]) * p.n + gn);
      x[i][0] += b.x; x[i][1] += b.y; x[i][2] += b.z; x[i][3] += b.w;
    }
  }
  double cs = 0.0, cq = 0.0;
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    if (ok[i]) {                                              // (the epilogue stores `out`)
      epi_finish<Tag>(p, eo[i], gmr[i], gn, x[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[i][e] = round_store<Tag>(x[i][e]); cs += (double)x[i][e]; cq += (double)x[i][e] * (double)x[i][e]; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cs += __shfl_xor(cs, o); cq += __shfl_xor(cq, o); }
  if ((tid & 63) == 0) { wsum[tid >> 6][0] = cs; wsum[tid >> 6][1] = cq; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < (((int)blockDim.x + 63) >> 6); ++w) { a += wsum[w][0]; b += wsum[w][1]; }
    const double cnt = (double)R * cpg, mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    s_stat[0] = (float)mean;
    s_stat[1] = (float)(1.0 / sqrt(var + (double)p.gn_eps));
  }
  __syncthreads();
  const float mean = s_stat[0], rstd = s_stat[1];
  const float sc[4] = {rstd * g4.x, rstd * g4.y, rstd * g4.z, rstd * g4.w};
  const float sh[4] = {b4.x - mean * sc[0], b4.y - mean * sc[1], b4.z - mean * sc[2], b4.w - mean * sc[3]};
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    if (ok[i]) {
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = fmaf(x[i][e], sc[e], sh[e]); y[e] = p.gn_silu ? silu_f(t) : t; }
      *(quad_t*)(p.gn_out + ((long)gmr[i] * p.ld_gn + gn) * ES) = f32_to_quad<Tag>(y);
    }
  }
}
// rows a thread of splitk_epilogue_gn_kernel holds for segments of `seg` rows: the smallest of 1, 2, 4, 8 with which (row lanes x column
// quads of a group) fit 1024 threads (0: the segment does not fit one block); *rl_n = row lanes
static inline int splitk_gn_rows(int seg, int n, int* rl_n) {
  const int nq = n / 128;                                   // column quads of one group (n / 32 channels)
  if (nq <= 0 || nq > 256 || seg <= 0) return 0;
  for (int mr = 1; mr <= 8; mr *= 2) {
    const int lanes = (seg + mr - 1) / mr;
    if ((long)lanes * nq <= 1024) { if (rl_n) *rl_n = lanes; return mr; }
  }
  return 0;
}
// second pass of a split-K launch (with or without the statistics)
template <typename Tag>
static inline void launch_splitk_epilogue(const GemmP& p, hipStream_t st) {
  if (p.gn_out) {
    int rl_n = 0;
    const int mr = splitk_gn_rows(p.stat_rows, p.n, &rl_n);
    const dim3 grid(p.m / p.stat_rows, 32), block((rl_n * (p.n / 128) + 63) / 64 * 64);
    if (mr == 1) hipLaunchKernelGGL((splitk_epilogue_gn_kernel<Tag, 1>), grid, block, 0, st, p, rl_n);
    else if (mr == 2) hipLaunchKernelGGL((splitk_epilogue_gn_kernel<Tag, 2>), grid, block, 0, st, p, rl_n);
    else if (mr == 4) hipLaunchKernelGGL((splitk_epilogue_gn_kernel<Tag, 4>), grid, block, 0, st, p, rl_n);
    else hipLaunchKernelGGL((splitk_epilogue_gn_kernel<Tag, 8>), grid, block, 0, st, p, rl_n);
    return;
  }
  if (p.stats) {
    hipLaunchKernelGGL(splitk_epilogue_stats_kernel<Tag>, dim3(p.m / p.stat_rows, (p.n + 31) / 32), dim3(256), 0, st, p);
    return;
  }
  long blocks = ((long)p.m * (p.n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_epilogue_kernel<Tag>, dim3((unsigned)blocks), dim3(256), 0, st, p);
}

// ---- configurations: {BM, BN, BK, NST, WGM, WGN}
template <typename Tag, int BM, int BN, int BK, int NST, int WGM, int WGN, int MODE>
void launch_mode(const GemmP& p, hipStream_t st) {
  constexpr int NT_ = 64 * WGM * WGN, CPR_ = BK / Elem<Tag>::EPC;
  constexpr size_t lds = (size_t)NST * (((BM * CPR_ + NT_ - 1) / NT_) + ((BN * CPR_ + NT_ - 1) / NT_)) * NT_ * 16
                         + ((MODE & 7) == 4 ? WGM * WGN * 1024 : 0);    // + the per-wave column-scale arrays  (MODE + 8: split-fp16 products, TT_F32)
  static_assert(lds + (size_t)WGM * WGN * 2 * (BN / WGN) * sizeof(float) <= 160 * 1024, "LDS ring (+ the statistics staging rows) exceeds 160 KiB");
  static_assert((MODE & 7) != 4 || BN / WGN <= 256, "column-scale array: 1 KiB per wave");
  constexpr size_t stat_lds = (size_t)WGM * WGN * 2 * (BN / WGN) * sizeof(float);        // per wave: sum and sum-of-squares rows of its columns
  static unsigned long long attr_done = 0;     // per kernel instance, one bit per device (see tt_lds_opt_in)
  tt_lds_opt_in((const void*)gemm_kernel<Tag, BM, BN, BK, NST, WGM, WGN, MODE>, (int)(lds + stat_lds), &attr_done);
  hipLaunchKernelGGL((gemm_kernel<Tag, BM, BN, BK, NST, WGM, WGN, MODE>), dim3(p.tiles_m * p.tiles_n * p.splitk),
                     dim3(64 * WGM * WGN), (p.stats && p.splitk == 1) ? lds + stat_lds : lds, st, p);
  if (p.splitk > 1) launch_splitk_epilogue<Tag>(p, st);
}

// LNOK: also instantiate the fused-LayerNorm variants (only the tile shapes the planner picks; gemm.hip keeps LayerNorm
// problems on them)
// multiply-shift pairs of every launch-uniform divisor the kernel meets (tile order, K split, conv / frame geometry, row groups)
static inline void fill_fastdivs(GemmP& p) {
  p.fd_splitk = make_fastdiv(p.splitk);
  p.fd_per_group = make_fastdiv((long)p.group_m * p.tiles_n);
  p.fd_group_m = make_fastdiv(p.group_m);
  p.fd_last_rows = make_fastdiv(p.tiles_m % p.group_m ? p.tiles_m % p.group_m : p.group_m);
  p.fd_per_tap = make_fastdiv(p.nk0 + p.nk1);
  p.fd_hwo = make_fastdiv((long)p.hout * p.wout);
  p.fd_wout = make_fastdiv(p.wout);
  p.fd_hw = make_fastdiv(p.hw);
  p.fd_frames = make_fastdiv(p.frames);
  p.fd_rv_rows = make_fastdiv(p.rowvec ? p.rowvec_rows : 1);
  p.fd_rv_mod = make_fastdiv(p.rowvec_mod > 0 ? p.rowvec_mod : 1);
}

template <typename Tag, int BM, int BN, int BK, int NST, int WGM, int WGN, bool LNOK = false>
void launch_cfg(GemmP& p, hipStream_t st) {
  p.tiles_m = ceil_div(p.m, BM);
  p.tiles_n = ceil_div(p.n, BN);
  {
    // group height that balances the A rows and W columns of one XCD's resident window (32 CUs x blocks per CU)
    constexpr int NT_ = 64 * WGM * WGN, CPR_ = BK / Elem<Tag>::EPC;
    constexpr long lds_ = (long)NST * (((BM * CPR_ + NT_ - 1) / NT_) + ((BN * CPR_ + NT_ - 1) / NT_)) * NT_ * 16;
    const int window = 32 * (lds_ * 2 <= 160 * 1024 ? 2 : 1);
    int gm = p.group_m_override;
    if (gm <= 0) {
      gm = 1;
      // few columns: a row's tiles are all resident together anyway and row-major keeps neighbouring rows (conv halos) adjacent
      if (p.tiles_n > 8)
        while (gm * 2 * BM * gm * 2 <= (long)window * BN && gm * 2 <= p.tiles_m) gm *= 2;    // gm^2 * BM <= window * BN
    }
    p.group_m = gm < 1 ? 1 : (gm > p.tiles_m ? p.tiles_m : gm);
  }
  p.nk0 = ceil_div(p.k0, BK); p.nk1 = p.k1 ? ceil_div(p.k1, BK) : 0;
  p.kt_total = p.taps * (p.nk0 + p.nk1);
  fill_fastdivs(p);
  if constexpr (std::is_same<Tag, f32_tag>::value) {
    if (p.f32_split) {                       // tt_gemm_set_f32_split(1): the split-fp16 product variants (KMODE + 8)
      // KMODE + 8 (split products) + 16 (W pre-split: packed weights) / + 32 (A pre-split: the swapped V^T projection's weights)
      const int pre = p.presplit & 3;
      if (LNOK && p.ln_fold == 1) {          // statistics of the A rows: A is never pre-split here (tt_gemm refuses it)
        if (pre & 2) launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 11 + 16>(p, st); else launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 11>(p, st);
        return;
      }
      if (LNOK && p.ln_fold == 2) {          // statistics of the W rows: W is never pre-split here
        if (pre & 1) launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 12 + 32>(p, st); else launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 12>(p, st);
        return;
      }
      switch (p.mode * 4 + pre) {
        case 0: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 8>(p, st); break;
        case 1: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 8 + 32>(p, st); break;
        case 2: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 8 + 16>(p, st); break;
        case 3: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 8 + 48>(p, st); break;
        case 4: case 5: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 9>(p, st); break;            // (convs: only the weight is ever pre-split)
        case 6: case 7: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 9 + 16>(p, st); break;
        case 8: case 9: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 10>(p, st); break;
        default: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 10 + 16>(p, st); break;
      }
      return;
    }
  }
  if constexpr (LNOK) {
    if (p.ln_fold == 1) { launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 3>(p, st); return; }
    if (p.ln_fold == 2) { launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 4>(p, st); return; }
  }
  switch (p.mode) {
    case 0: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 0>(p, st); break;
    case 1: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 1>(p, st); break;
    default: launch_mode<Tag, BM, BN, BK, NST, WGM, WGN, 2>(p, st); break;
  }
}

template <typename Tag, bool HAS_RES, bool HAS_RV = false>
void launch_sq320(const GemmP& p, hipStream_t st) {
  constexpr size_t lds = (size_t)(HAS_RES ? 3 * 2 * SQ_TILE_BYTES : 5 * SQ_TILE_BYTES) + SQ_WAVES * 4096;
  static_assert(lds <= 160 * 1024, "sq320 LDS");
  static unsigned long long attr_done = 0;
  tt_lds_opt_in((const void*)sq320_kernel<Tag, HAS_RES, HAS_RV>, (int)lds, &attr_done);
  const int ntiles = (p.m + SQ_ROWS - 1) / SQ_ROWS;
  hipLaunchKernelGGL((sq320_kernel<Tag, HAS_RES, HAS_RV>), dim3(ntiles < 256 ? ntiles : 256), dim3(SQ_NT), lds, st, p);
}

template <typename Tag>
void launch(GemmP& p, int cfg, hipStream_t st) {          // cfg = index into kCfgs (gemm.hip)
  switch (cfg) {
    case 0: launch_cfg<Tag, 128, 128, 64, 2, 2, 2>(p, st); break;
    case 1: launch_cfg<Tag, 128, 64, 64, 3, 2, 2, true>(p, st); break;
    case 2: launch_cfg<Tag, 64, 64, 64, 4, 2, 2, true>(p, st); break;
    case 3: launch_cfg<Tag, 256, 128, 32, 3, 4, 2, true>(p, st); break;
    case 4: launch_cfg<Tag, 256, 256, 32, 3, 2, 4>(p, st); break;
    case 5: launch_cfg<Tag, 128, 128, 32, 3, 2, 2>(p, st); break;
    case 6: launch_cfg<Tag, 256, 128, 64, 3, 4, 2>(p, st); break;
    case 7: launch_cfg<Tag, 128, 160, 64, 2, 4, 1, true>(p, st); break;
    case 8: launch_cfg<Tag, 128, 320, 32, 3, 4, 2>(p, st); break;
    case 9: launch_cfg<Tag, 256, 256, 64, 2, 2, 4, true>(p, st); break;
    case 10: launch_cfg<Tag, 128, 128, 64, 4, 2, 2>(p, st); break;
    case 11: launch_cfg<Tag, 128, 128, 64, 2, 4, 2, true>(p, st); break;
    case 12: launch_cfg<Tag, 256, 160, 32, 3, 8, 1>(p, st); break;
    case 13: launch_cfg<Tag, 256, 128, 32, 4, 4, 2>(p, st); break;
    case 14: launch_cfg<Tag, 128, 128, 32, 5, 4, 2>(p, st); break;
    case 15: launch_cfg<Tag, 128, 128, 64, 3, 4, 2>(p, st); break;
    case 16: launch_cfg<Tag, 128, 128, 64, 4, 4, 2, true>(p, st); break;
    case 17: launch_cfg<Tag, 256, 256, 32, 4, 2, 4>(p, st); break;
    case 18: launch_cfg<Tag, 256, 128, 64, 2, 4, 2>(p, st); break;
    case 19: launch_cfg<Tag, 256, 128, 32, 5, 4, 2>(p, st); break;
    default: launch_cfg<Tag, 128, 128, 128, 2, 4, 2>(p, st); break;
  }
}

}  // namespace ttg
