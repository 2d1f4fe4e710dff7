// gemm_w320: the 256 x 320 x 64 big-tile member of the tt_gemm family (round 4), for the problems of the FINEST UNet level whose
// output is 320 channels wide (or a multiple): the ResnetBlock2D 3x3 convs, the (3,1,1) temporal convs, the 1x1 shortcuts over the
// skip concat, proj_in / proj_out, to_out, FF2 and the LayerNorm-folded Q projection at 32x56 latents (M = 28 x 1792 = 50176 rows =
// exactly 196 row tiles) -- reference svd/diffusion_arch/unet_3d_blocks.py:2094,2212,2311 (up blocks) and
// svd/diffusion_arch/transformer_temporal.py:323-376.  The tiled template serves them with 128 x 160 tiles (wave tile 32 x 160) whose
// K step needs 57 bytes of LDS fill per MFMA clock -- more than the ~54 B/clk/CU an LDS-DMA stream delivers (tools/dma_shape_test.hip),
// so it runs at 280-800 TFLOP/s; N = 320 does not tile by the persistent kernel's 256 columns either (gemm_pp.hip: 37 % column waste).
//
// Structure:
//   * 256 x 320 output tile (full width at N = 320: no column waste, 28.6 B of fill per MFMA clock), ONE tile per workgroup
//     (196 tiles: one round on 196 of the 256 CUs; the other CUs stay free for the concurrent branch of the step graph),
//     8 waves as 4 (rows) x 2 (columns): wave tile 64 x 160 = 2 x 5 MFMA fragments of 32 x 32 (160 accumulator registers);
//   * K slabs of 64 elements = 128-byte rows (full cache lines per LDS-DMA piece); two 72 KiB slots, each the A region (256 rows,
//     32 KiB) and five W regions (W_j = the j-th 32-column fragment of BOTH wave columns, 64 rows, 8 KiB);
//   * a slab is consumed in FIVE phases of 8 MFMAs: phase j multiplies the A half-tile of the wave (read once in phase 0 and
//     kept in 32 registers) with W_j.  Every region is read exactly once per slab and re-staged ONE phase after its read for
//     slab s+2 (two pieces per thread and phase), so 8 pieces per thread are always in flight and the only counted wait is
//     vmcnt(8) at the end of phase 4;
//   * the two groups of four waves (one of each per SIMD) run one barrier apart as in gemm_pp.hip: one issues its 8 MFMAs while
//     the other reads fragments and issues DMA;
//   * the producer walks (tap, source, k) like the tiled template: conv3x3 (stride 1) and temporal-conv taps shift the row
//     offsets of the 4 A pieces of a thread (recomputed once per tap), halo / ragged rows read zeros through the descriptor
//     bounds check, two channel sources implement the skip concat;
//   * epilogue after the K loop in the (then free) ring: accumulators transposed through a wave-private 8 KiB fp32 strip, every
//     load / store instruction covers 4 rows x 128 contiguous bytes; bias, scale, FiLM / frame-position row vector, residual,
//     AlphaBlender, 1/sigma of the fused LayerNorm (statistics gathered from the A fragments in phase 1's read interval).
#include <stdlib.h>
#include <type_traits>
#include "gemm_kernel.h"

#ifndef TT_W320_A_AUX
#define TT_W320_A_AUX 0      // cache policy of the A stream's LDS-DMA (aux: 2 = nt); measured: see DESIGN.md 6.R4
#endif

namespace ttg {

template <int N> __device__ __forceinline__ void w3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


// ---- epilogue of ONE fragment row of a wave (32 rows x 160 columns, acc[5] in the MFMA layout: lane (l31, hi) holds columns
// j*32 + 8g + 4hi + {0..3} of row l31), shared by both kernels of this file.  The arithmetic and its order are gemm_kernel.h's:
// v = (acc * rs + bias) * scale (+ row vector) ; out = alpha * blend + (1 - alpha) * (v + residual).  Per 64-column chunk the
// accumulators go through the wave's 8 KiB strip so that 16 consecutive lanes hold 64 consecutive columns of one row: every
// residual / blend load and every store instruction covers 4 rows x 128 contiguous bytes.
// Row offsets are 32-bit, without a per-row select: a row >= m lies beyond the extent its descriptor was built for ((m-1) * ld + n
// elements), so the bounds check returns 0 for its loads and drops its store (hipcc turns a `row < m ? offset : invalid` select
// around a 64-bit product into exec-masked branches with one load per side).  The function must not spill: a scratch reload is a
// VMEM operation behind the stores in flight (in-order vmcnt) and serialises the passes (measured with a 16-byte-per-lane variant
// that spilled 12-30 registers next to the 160 live accumulators: +4 us per tile).
// XCHG (gemm_w320h_kernel): the wave holds only one K half of the tile.  Per chunk it first writes the fragments of the row it
// gives away (`give`) lane-linearly into `xbuf`; after a workgroup barrier the partner's partial sums of THIS row are read from `pbuf`
// and added on the way into the strip (the accumulator tuples are never updated element-wise: hipcc spills them if they are).
template <typename Tag, int NI, bool FILM, bool RES, bool BLEND, bool XCHG, bool STATS>
__device__ __forceinline__ void w3_epilogue_rows(const GemmP& p, const f32x16_t (*accs)[5], const float* rss, char* ebuf, int mb0, int ncol0,
                                                 int lane, float alpha, const f32x16_t (&give)[5], char* xbuf, const char* pbuf, float* sstage) {
  typedef typename Elem<Tag>::quad_t quad_t;
  constexpr int ES = 2;
  const int hi = lane >> 5;
  const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(p.bias, p.bias_bytes);
  const __amdgpu_buffer_rsrc_t r_out = make_rsrc(p.out, p.out_bytes);
  const __amdgpu_buffer_rsrc_t r_rv = make_rsrc(p.rowvec, FILM ? p.rowvec_bytes : 0);
  const __amdgpu_buffer_rsrc_t r_res = make_rsrc(p.residual, RES ? p.res_bytes : 0);
  const __amdgpu_buffer_rsrc_t r_bl = make_rsrc(p.blend, BLEND ? p.blend_bytes : 0);
  const float one_m_alpha = 1.0f - alpha;
  const int rv_rows = FILM ? p.rowvec_rows : 1;
  // STATS (GemmP.stats): column sums of the stored values (gemm_kernel.h, colstat_strip).  Compile-time like the operand variants: a
  // uniform run-time `if` inside the pass loops turns the straight-line epilogue into basic blocks, and hipcc then drains vmcnt
  // between them (measured: +4 us on a 23 us launch with the run-time form)
  const bool parity = FILM && p.rowvec_mod == 2 && rv_rows == 1;             // even / odd rows: two vectors, the row's parity selects
  auto wrap = [&](int g) { return p.rowvec_mod > 0 ? g % p.rowvec_mod : g; };   // periodic row vector (uniform, once per fragment row)
  auto strip_off = [](int row, int quad, int nq) { return row * 256 + ((quad ^ (row & (nq - 1))) << 4); };
#pragma unroll
  for (int i = 0; i < NI; ++i) {
  const f32x16_t (&acc)[5] = accs[i];
  const float rs = rss[i];
  const int mb = mb0 + i * 32;
  // the strip addresses of the transposition are recomputed per fragment row (a few VALU): kept live from row 0 to row 1 they were the
  // registers hipcc spilled once the statistics code was added (3-8 per lane, reloaded behind a vmcnt(0))
  int l31 = lane & 31;
  asm volatile("" : "+v"(l31));
  const int grp_raw = mb / rv_rows, grp_split = (grp_raw + 1) * rv_rows;    // row group of the fragment's first row, first row of the next
  const int grp0 = parity ? 0 : wrap(grp_raw), grp1 = parity ? 1 : wrap(grp_raw + 1);
  const int sel_split = parity ? 0x7fffffff : grp_split, sel_odd = parity ? 1 : 0;
  // Residual only (the common case: to_out, FF2, proj_out, conv2 + shortcut): ALL 24 loads of the fragment row are issued before
  // the first chunk is processed.  Batch by batch (loads, wait, stores, next loads behind those stores: in-order vmcnt) a wave pays
  // one memory round trip per batch -- 12 per tile, the whole epilogue of a K = 320 problem; up front it pays one (two) per fragment row.
  constexpr bool PRELOAD = RES && !BLEND;
  // chunks loaded up front: what the registers next to the live accumulators hold without spilling (one chunk fewer with the statistics code)
  constexpr int NPRE = (NI == 2 ? 2 : 3) - (STATS ? 1 : 0);
  quad_t pre[PRELOAD ? NPRE : 1][PRELOAD ? 8 : 1];
  if constexpr (PRELOAD) {
#pragma unroll
    for (int c = 0; c < NPRE; ++c) {
      const int nfr = c < 2 ? 2 : 1, q_per_row = nfr * 8, rows_per_pass = 64 / q_per_row;
      const unsigned o = ((unsigned)(mb + lane / q_per_row) * (unsigned)p.ld_res + (unsigned)(ncol0 + c * 64 + (lane % q_per_row) * 4)) * ES;
      const unsigned st = (unsigned)(rows_per_pass * p.ld_res * ES);
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        pre[c][pass] = zero_quad<Tag>();
        if (pass * rows_per_pass < 32) pre[c][pass] = ldq<Tag>(r_res, (int)(o + pass * st));
      }
    }
  }
#pragma unroll
  for (int jc = 0; jc < 5; jc += 2) {
    const int nfr = (jc + 1 < 5) ? 2 : 1;
    const int q_per_row = nfr * 8, rows_per_pass = 64 / q_per_row;
    const int qq = lane % q_per_row, rq = lane / q_per_row;
    const int gn = ncol0 + jc * 32 + qq * 4;
    const float4 b4 = ld128f(r_bias, gn * 4);
    float4 film_lo = make_float4(0.f, 0.f, 0.f, 0.f), film_hi = film_lo;
    if constexpr (FILM) {                                                  // beyond the last group: out of range -> 0
      film_lo = ld128f(r_rv, (int)(((unsigned)grp0 * (unsigned)p.ld_rowvec + (unsigned)gn) * 4u));
      film_hi = ld128f(r_rv, (int)(((unsigned)grp1 * (unsigned)p.ld_rowvec + (unsigned)gn) * 4u));
    }
    if constexpr (XCHG) {
      if (jc) __syncthreads();                               // the partner has consumed the previous chunk's exchange buffer
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (jc + jj < 5) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(float4*)(xbuf + ((jj * 4 + g) * 64 + lane) * 16) =
                make_float4(give[jc + jj][g * 4], give[jc + jj][g * 4 + 1], give[jc + jj][g * 4 + 2], give[jc + jj][g * 4 + 3]);
        }
      __syncthreads();
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (jc + jj < 5) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (XCHG) o = *(const float4*)(pbuf + ((jj * 4 + g) * 64 + lane) * 16);
          *(float4*)(ebuf + strip_off(l31, jj * 8 + 2 * g + hi, nfr * 8)) =
              make_float4((acc[jc + jj][g * 4] + o.x) * rs, (acc[jc + jj][g * 4 + 1] + o.y) * rs, (acc[jc + jj][g * 4 + 2] + o.z) * rs,
                          (acc[jc + jj][g * 4 + 3] + o.w) * rs);
        }
      }
    const unsigned row0 = (unsigned)(mb + rq);
    const unsigned o_out = (row0 * (unsigned)p.ldo + (unsigned)gn) * ES, s_out = (unsigned)(rows_per_pass * p.ldo * ES);
    const unsigned o_res = (row0 * (unsigned)p.ld_res + (unsigned)gn) * ES, s_res = (unsigned)(rows_per_pass * p.ld_res * ES);
    const unsigned o_bl = (row0 * (unsigned)p.ld_blend + (unsigned)gn) * ES, s_bl = (unsigned)(rows_per_pass * p.ld_blend * ES);
    constexpr int PB = 4;                                    // passes per batch: loads first, stores last (in-order vmcnt)
#pragma unroll
    for (int pb = 0; pb < 8; pb += PB) {
      quad_t rqv[PB], blv[PB];
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        const int pass = pb + k;
        rqv[k] = blv[k] = zero_quad<Tag>();
        if (pass * rows_per_pass < 32) {
          if (PRELOAD && jc / 2 < NPRE) rqv[k] = pre[jc / 2 < NPRE ? jc / 2 : 0][pass];
          else if constexpr (RES) rqv[k] = ldq<Tag>(r_res, (int)(o_res + pass * s_res));
          if constexpr (BLEND) blv[k] = ldq<Tag>(r_bl, (int)(o_bl + pass * s_bl));
        }
      }
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        const int pass = pb + k;
        if (pass * rows_per_pass < 32) {
          const int r = pass * rows_per_pass + rq;
          const float4 t = *(const float4*)(ebuf + strip_off(r, qq, q_per_row));
          float v[4] = {(t.x + b4.x) * p.acc_scale, (t.y + b4.y) * p.acc_scale, (t.z + b4.z) * p.acc_scale, (t.w + b4.w) * p.acc_scale};
          if constexpr (FILM) {
            // branch-free selector (a `parity ? .. : ..` here became a uniform BRANCH per pass and broke the straight-line code):
            // groups of >= 32 rows: rows from grp_split on take the next group's vector; even / odd form: odd rows (mb is a multiple of 32)
            const float4 f = ((mb + r >= sel_split) | ((r & sel_odd) != 0)) ? film_hi : film_lo;
            v[0] += f.x; v[1] += f.y; v[2] += f.z; v[3] += f.w;
          }
          float r4[4], b4v[4];
          quad_to_f32<Tag>(rqv[k], r4);
          quad_to_f32<Tag>(BLEND ? blv[k] : rqv[k], b4v);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = alpha * b4v[e] + one_m_alpha * (v[e] + r4[e]);
          const quad_t packed = f32_to_quad<Tag>(v);
          st64(r_out, (int)(o_out + pass * s_out), packed.x, packed.y);
          if constexpr (STATS) {                             // (rows >= m never occur: stats are granted for m % tile rows == 0 only)
            float sv[4];
            quad_to_f32<Tag>(packed, sv);
            *(float4*)(ebuf + strip_off(r, qq, q_per_row)) = make_float4(sv[0], sv[1], sv[2], sv[3]);
          }
        }
      }
    }
    if constexpr (STATS) colstat_strip(ebuf, q_per_row, lane, sstage + jc * 32, sstage + 160 + jc * 32, i == 0);
  }
  }
}
// uniform dispatch over the operand variants (each straight-line); NI fragment rows starting at row mb0
template <typename Tag, int NI, bool XCHG>
__device__ __forceinline__ void w3_epilogue_dispatch(const GemmP& p, const f32x16_t (*accs)[5], const float* rss, char* ebuf, int mb0, int ncol0,
                                                     int lane, const f32x16_t (&give)[5], char* xbuf, const char* pbuf, float* sstage) {
  const float alpha = p.blend ? p.alpha : 0.0f;
  const bool blend_is_res = p.blend && p.blend == p.residual && p.ld_blend == p.ld_res;
  const bool film = p.rowvec != nullptr, res = p.residual != nullptr, bl = p.blend && !blend_is_res;
#define W3_RUN(F, R, B, S) w3_epilogue_rows<Tag, NI, F, R, B, XCHG, S>(p, accs, rss, ebuf, mb0, ncol0, lane, alpha, give, xbuf, pbuf, sstage)
  if (p.stats) {                                             // (never with a blend tensor distinct from the residual: tt_gemm_stats_rows)
    if (res) { if (film) W3_RUN(true, true, false, true); else W3_RUN(false, true, false, true); }
    else { if (film) W3_RUN(true, false, false, true); else W3_RUN(false, false, false, true); }
  }
  else if (bl) { if (film) W3_RUN(true, true, true, false); else W3_RUN(false, true, true, false); }
  else if (res) { if (film) W3_RUN(true, true, false, false); else W3_RUN(false, true, false, false); }
  else { if (film) W3_RUN(true, false, false, false); else W3_RUN(false, false, false, false); }
#undef W3_RUN
}

// ---- GemmP.stats, tile level: the waves' staging rows ([wave][2][160] floats at `st0`) of the waves that share a column half,
// added in a fixed order; thread c < 320 finishes column c.  `nw` waves per column half: wave index of (k, half) = wave_of(k, half).
// `parts` statistics tiles per output tile (1, or 2 on the 128-row kernel when GemmP.stat_rows asks for the 64 rows of one wave row: 448-row images).
template <typename F> __device__ __forceinline__ void w3_stats_tile(const GemmP& p, const float* st0, int tid, int tile_m, int n0, int nw, F wave_of, int parts = 1) {
  __syncthreads();
  if (tid < 320) {
    const int half = tid >= 160 ? 1 : 0, col = tid - half * 160, per = nw / parts;
    for (int part = 0; part < parts; ++part) {
      float a = 0.f, b = 0.f;
      for (int k = part * per; k < (part + 1) * per; ++k) {
        const float* row = st0 + wave_of(k, half) * 320;
        a += row[col]; b += row[160 + col];
      }
      const long srow = (long)tile_m * parts + part;
      p.stats[(srow * 2) * p.n + n0 + tid] = a;
      p.stats[(srow * 2 + 1) * p.n + n0 + tid] = b;
    }
  }
}

// ---- split-K form of the above for ONE fragment row of gemm_w320h_kernel (p.splitk > 1): the fp32 partial sums of K slice `split`
// (after the exchange of the two in-slab K halves) go to the workspace slab ws[split][m][n] through the same strip transposition,
// 16 bytes per lane and 4 rows x 256 contiguous bytes per store instruction; splitk_epilogue_kernel (gemm_kernel.h) adds the slabs
// in a fixed order and runs the full epilogue.  The slab's descriptor ends at row m: rows beyond it are dropped by the bounds check.
__device__ __forceinline__ void w3_partial_row(const GemmP& p, const f32x16_t (&acc)[5], char* ebuf, int mb, int ncol0, int lane, int split,
                                               const f32x16_t (&give)[5], char* xbuf, const char* pbuf) {
  const int l31 = lane & 31, hi = lane >> 5;
  const __amdgpu_buffer_rsrc_t r_ws = make_rsrc(p.ws + (long)split * p.m * p.n, (unsigned)((long)p.m * p.n * 4));
  auto strip_off = [](int row, int quad, int nq) { return row * 256 + ((quad ^ (row & (nq - 1))) << 4); };
#pragma unroll
  for (int jc = 0; jc < 5; jc += 2) {
    const int nfr = (jc + 1 < 5) ? 2 : 1;
    const int q_per_row = nfr * 8, rows_per_pass = 64 / q_per_row;
    const int qq = lane % q_per_row, rq = lane / q_per_row;
    const int gn = ncol0 + jc * 32 + qq * 4;
    if (jc) __syncthreads();                                 // the partner has consumed the previous chunk's exchange buffer
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (jc + jj < 5) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(xbuf + ((jj * 4 + g) * 64 + lane) * 16) =
              make_float4(give[jc + jj][g * 4], give[jc + jj][g * 4 + 1], give[jc + jj][g * 4 + 2], give[jc + jj][g * 4 + 3]);
      }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (jc + jj < 5) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 o = *(const float4*)(pbuf + ((jj * 4 + g) * 64 + lane) * 16);
          *(float4*)(ebuf + strip_off(l31, jj * 8 + 2 * g + hi, nfr * 8)) =
              make_float4(acc[jc + jj][g * 4] + o.x, acc[jc + jj][g * 4 + 1] + o.y, acc[jc + jj][g * 4 + 2] + o.z, acc[jc + jj][g * 4 + 3] + o.w);
        }
      }
    const unsigned o_ws = ((unsigned)(mb + rq) * (unsigned)p.n + (unsigned)gn) * 4u, s_ws = (unsigned)(rows_per_pass * p.n * 4);
#pragma unroll
    for (int pass = 0; pass < 8; ++pass)
      if (pass * rows_per_pass < 32)
        st128f(r_ws, (int)(o_ws + pass * s_ws), *(const float4*)(ebuf + strip_off(pass * rows_per_pass + rq, qq, q_per_row)));
  }
}

template <typename Tag, int MODE, int LNROWS>
__global__ __launch_bounds__(512, 2) void gemm_w320_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kernarg_touch<sizeof(GemmP)>();
  constexpr int BM = 256, BN = 320, ES = 2, CPR = 8, FM = 2, FN = 5;
  constexpr int A_BYTES = 32768, WJ = 8192, SLOT = A_BYTES + FN * WJ;          // 73728 bytes per slot
  static_assert(Elem<Tag>::ES == 2, "16-bit storage types only");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wc = wid >> 2, wr = wid & 3;
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-contiguous tile order (gemm_kernel.h): neighbouring row tiles (conv halos) and the column tiles of one row share an L2
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int S = p.kt_total;                                                    // slabs: taps * (k0 + k1) / 64

  const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(p.a0, p.a0_bytes);
  const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(p.a1 ? p.a1 : p.a0, p.a1_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  // ---- producer: thread t stages 16-byte chunk (t & 7) of row (t >> 3) of every 64-row pass: 4 passes of A, one per W region
  const int rr = tid >> 3, ch = (tid & 7) ^ tile_swz<CPR>(rr);                 // source chunk (the swizzle only depends on rr mod 16)
  const int w_v = (int)(((long)(n0 + (rr >> 5) * 160 + (rr & 31)) * p.ldw + ch * 8) * ES);     // n % 320 == 0: always in range
  const int w_jstride = __builtin_amdgcn_readfirstlane((int)(32 * p.ldw * ES));
  int a_pos[4];                                                                // MODE 1: y << 16 | x ; MODE 2: frame index
  int a_off[4];                                                                // byte offsets of the producer's (tap, source)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_pos[i] = 0;
    const int g = min(m0 + i * 64 + rr, p.m - 1);
    if constexpr (MODE == 1) {
      const int hw = p.hout * p.wout, rem = g - (g / hw) * hw, y = rem / p.wout;
      a_pos[i] = (y << 16) | (rem - y * p.wout);
    } else if constexpr (MODE == 2) {
      a_pos[i] = (g / p.hw) % p.frames;
    }
  }
  int p_tap = 0, p_src = 0, p_kc = 0, p_slab = 0;
  auto refresh = [&]() {
    const long lda = p_src ? p.lda1 : p.lda0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = m0 + i * 64 + rr;
      bool ok = g < p.m;
      long row = g;
      if constexpr (MODE == 1) {
        const int dy = p_tap / 3 - 1, dx = p_tap - (dy + 1) * 3 - 1;
        const int y = (a_pos[i] >> 16) + dy, x = (a_pos[i] & 0xffff) + dx;
        ok = ok && (unsigned)y < (unsigned)p.hin && (unsigned)x < (unsigned)p.win;
        row = (long)g + dy * p.win + dx;
      } else if constexpr (MODE == 2) {
        ok = ok && (unsigned)(a_pos[i] + p_tap - 1) < (unsigned)p.frames;
        row = (long)g + (long)(p_tap - 1) * p.hw;
      }
      a_off[i] = ok ? (int)((row * lda + ch * 8) * ES) : kInv;
    }
  };
  refresh();
  int p_soff_a = 0, p_soff_w = 0;
  bool p_ok = S > 0;
  auto advance = [&]() {                    // the producer moves to the next slab (called after the piece that closes a slab: W_4)
    if (++p_kc == (p_src ? p.nk1 : p.nk0)) {
      p_kc = 0;
      if (p_src == 0 && p.nk1 > 0) p_src = 1; else { p_src = 0; ++p_tap; }
      if (MODE != 0 || p.nk1 > 0) refresh();
    }
    ++p_slab;
    p_ok = p_slab < S;
    p_soff_a = __builtin_amdgcn_readfirstlane(p_kc * 64 * ES);
    p_soff_w = __builtin_amdgcn_readfirstlane((int)(((long)p_tap * (p.k0 + p.k1) + (p_src ? p.k0 : 0) + p_kc * 64) * ES));
  };
  auto stage_a = [&](int slot, auto i_tag) {
    constexpr int I = decltype(i_tag)::value;
    char* dst = smem + slot * SLOT + I * 8192 + wid * 1024;
    const int v = p_ok ? a_off[I] : kInv;
    if (__builtin_amdgcn_readfirstlane(p_src))
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra1, (__attribute__((address_space(3))) void*)dst, 16, v, p_soff_a, 0, TT_W320_A_AUX);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra0, (__attribute__((address_space(3))) void*)dst, 16, v, p_soff_a, 0, TT_W320_A_AUX);
  };
  auto stage_w = [&](int slot, auto j_tag) {
    constexpr int J = decltype(j_tag)::value;
    char* dst = smem + slot * SLOT + A_BYTES + J * WJ + wid * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, p_ok ? w_v : kInv,
                                             p_soff_w + J * w_jstride, 0, 0);
  };
  using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>; using C4 = std::integral_constant<int, 4>;

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ln_s[FM], ln_q[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) ln_s[i] = ln_q[i] = 0.f;

  // ---- consumer: fragment addresses inside a slot.  A fragment i: rows wr*64 + i*32 + l31 (+4096 bytes per i: same swizzle);
  // W fragment j: region j, rows wc*32 + l31 (+8192 bytes per j)
  const unsigned lds_base = lds_addr(smem);
  unsigned a_addr[4], b_addr[4];
  {
    const int ar = wr * 64 + l31, br = wc * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a_addr[ks] = ar * 128 + (((ks * 2 + hi) ^ tile_swz<CPR>(ar)) << 4);
      b_addr[ks] = A_BYTES + br * 128 + (((ks * 2 + hi) ^ tile_swz<CPR>(br)) << 4);
    }
  }
  raw_u32x4_t af[FM][4], bf[4];
  auto read_a = [&](unsigned sb) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      af[0][ks] = lds_read16_raw_off<0>(sb + a_addr[ks]);
      af[1][ks] = lds_read16_raw_off<4096>(sb + a_addr[ks]);
    }
  };
  auto read_b = [&](unsigned sb, auto j_tag) {
    constexpr int J = decltype(j_tag)::value;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bf[ks] = lds_read16_raw_off<J * WJ>(sb + b_addr[ks]);
  };
  auto mma = [&](auto j_tag) {
    constexpr int J = decltype(j_tag)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
        acc[i][J] = Cvt<Tag>::mfma32(make_uint4(bf[ks].x, bf[ks].y, bf[ks].z, bf[ks].w),
                                     make_uint4(af[i][ks].x, af[i][ks].y, af[i][ks].z, af[i][ks].w), acc[i][J]);
    __builtin_amdgcn_s_setprio(0);
  };
  // LayerNorm sums of the wave's A rows: in the read interval of phase 1 (only 4 W reads to wait for, the A fragments are in
  // registers), while the other group's wave on this SIMD issues its MFMAs.  Both wave columns hold the same A fragments: each
  // sums the K steps of its parity and the halves meet in the epilogue.
  auto stats = [&]() {
    if constexpr (LNROWS) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if ((ks & 1) == wc) {
#pragma unroll
          for (int i = 0; i < FM; ++i) ln_stat<Tag>(af[i][ks], ln_s[i], ln_q[i]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };

  // ---- prologue: slab 0 complete (slot 0) + everything of slab 1 but W_4 (slot 1), as the steady state would have issued them
  stage_w(0, C0{}); stage_a(0, C0{}); stage_w(0, C1{}); stage_a(0, C1{}); stage_w(0, C2{}); stage_a(0, C2{});
  stage_w(0, C3{}); stage_a(0, C3{}); stage_w(0, C4{});
  advance();
  stage_w(1, C0{}); stage_a(1, C0{}); stage_w(1, C1{}); stage_a(1, C1{}); stage_w(1, C2{}); stage_a(1, C2{});
  stage_w(1, C3{}); stage_a(1, C3{});
  w3_wait_vm<8>();
  bar();
  if (grp == 1) bar();                                       // group 1 runs one barrier behind group 0

  for (int s = 0; s < S; ++s) {
    const int slot = s & 1;
    const unsigned sb = lds_base + slot * SLOT;
    // ---- phase 0: A (kept for the slab) + W_0 ; DMA: W_4 of slab s+1 (other slot, read in phase 4 of slab s-1) closes that slab
    read_a(sb); read_b(sb, C0{});
    stage_w(slot ^ 1, C4{});
    advance();
    lds_wait<0>();
    bar();
    mma(C0{});
    bar();
    // ---- phase 1: W_1 ; DMA: W_0 and A rows 0-63 of slab s+2 (this slot, read in phase 0)
    read_b(sb, C1{});
    stage_w(slot, C0{}); stage_a(slot, C0{});
    stats();
    lds_wait<0>();
    bar();
    mma(C1{});
    bar();
    // ---- phase 2
    read_b(sb, C2{});
    stage_w(slot, C1{}); stage_a(slot, C1{});
    lds_wait<0>();
    bar();
    mma(C2{});
    bar();
    // ---- phase 3
    read_b(sb, C3{});
    stage_w(slot, C2{}); stage_a(slot, C2{});
    lds_wait<0>();
    bar();
    mma(C3{});
    bar();
    // ---- phase 4 ; everything of slab s+1 has landed once at most the 8 pieces of phases 1-4 are in flight
    read_b(sb, C4{});
    stage_w(slot, C3{}); stage_a(slot, C3{});
    w3_wait_vm<8>();
    lds_wait<0>();
    bar();
    mma(C4{});
    bar();
  }
  if (grp == 0) bar();                                       // both groups leave the loop together
  w3_wait_vm<0>();                                           // the zero-fill pieces staged past the last slab must not land in the strips
  bar();

  // ---- fused LayerNorm: 1/sigma of the wave's rows (lane <-> row l31 of fragment i); the two wave columns summed alternate K steps
  float rs[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) rs[i] = 1.0f;
  if constexpr (LNROWS) {
    float* xs = (float*)(smem + 8 * 8192);                   // [wave][lane][4] partial sums behind the strips
    *(float4*)(xs + (wid * 64 + lane) * 4) = make_float4(ln_s[0], ln_q[0], ln_s[1], ln_q[1]);
    __syncthreads();
    const float4 o = *(const float4*)(xs + ((wid ^ 4) * 64 + lane) * 4);      // the same rows in the other wave column
    const float inv_k = 1.0f / (float)p.k0;
    const float s0 = ln_s[0] + o.x, q0 = ln_q[0] + o.y, s1 = ln_s[1] + o.z, q1 = ln_q[1] + o.w;
    const float sm0 = (s0 + __shfl_xor(s0, 32)) * inv_k, sq0 = (q0 + __shfl_xor(q0, 32)) * inv_k;
    const float sm1 = (s1 + __shfl_xor(s1, 32)) * inv_k, sq1 = (q1 + __shfl_xor(q1, 32)) * inv_k;
    rs[0] = rsqrtf(fmaxf(sq0 - sm0 * sm0, 0.f) + p.ln_eps);
    rs[1] = rsqrtf(fmaxf(sq1 - sm1 * sm1, 0.f) + p.ln_eps);
  }

  // ---- epilogue in the (now free) ring: one 8 KiB strip per wave
  char* ebuf = smem + wid * 8192;
  float* sst = (float*)(smem + 9 * 8192);                    // statistics staging rows behind the strips and the LayerNorm partials: [wave][2][160]
  w3_epilogue_dispatch<Tag, FM, false>(p, acc, rs, ebuf, m0 + wr * 64, n0 + wc * 160, lane, acc[0], nullptr, nullptr, sst + wid * 320);
  if (p.stats) w3_stats_tile(p, sst, tid, tile_m, n0, 4, [](int k, int half) { return half * 4 + k; });      // wid = wc * 4 + wr
}

// ---- gemm_w320h: the half-height variant, 128 x 320 x 64 tiles, for problems with too few rows for a round of 256-row tiles
// (the second UNet level at 32x56 latents: M = 28 x 448 = 12544 rows, N = 640 = 2 x 320 -> 98 x 2 = 196 tiles; the live-row
// projections of the finest level: 25088 rows).  Same slots, regions and staging as above (the A region is 128 rows = 2 passes),
// but the two wave groups SPLIT K inside every slab instead of the rows: the four waves of group g (2 x 2 over the tile: wave tile
// 64 x 160, the same 160 accumulators) multiply K steps {2g, 2g+1} of each slab, 20 MFMAs per wave and slab in three phases
//     P0  A (4 reads, kept) x W_0, W_1 (8 MFMAs)     P1  W_2, W_3 (8)     P2  W_4 (4)
// and the two partial tiles meet in the epilogue: every wave hands the fragment row it does not finish to its partner (same
// (row, column) wave of the other group) through LDS -- group 0 finishes the upper 32 rows of each wave tile, group 1 the
// lower -- so all 8 waves share the transposition / residual / store work.  A region read in phase p is re-staged for slab s+2 in
// phase p+1 (7 pieces per thread and slab); the counted wait is vmcnt(6) at the end of P2.
template <typename Tag, int MODE, int LNROWS>
__global__ __launch_bounds__(512, 2) void gemm_w320h_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kernarg_touch<sizeof(GemmP)>();
  constexpr int BM = 128, BN = 320, ES = 2, CPR = 8, FM = 2, FN = 5;
  constexpr int A_BYTES = 16384, WJ = 8192, SLOT = A_BYTES + FN * WJ;          // 57344 bytes per slot
  static_assert(Elem<Tag>::ES == 2, "16-bit storage types only");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wr = (wid >> 1) & 1, wc = wid & 1;                 // K half, 64-row half, 160-column half
  const int l31 = lane & 31, hi = lane >> 5;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // split-K (p.splitk > 1): the K slices of one tile sit next to each other in the launch order; slice `split` covers slabs
  // [kt_lo, kt_lo + S) of the (tap, source, k) walk and ends in w3_partial_row instead of the epilogue
  int split = 0;
  if (p.splitk > 1) { split = bid % p.splitk; bid /= p.splitk; }
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt_lo = p.splitk > 1 ? (int)((long)p.kt_total * split / p.splitk) : 0;
  const int S = p.splitk > 1 ? (int)((long)p.kt_total * (split + 1) / p.splitk) - kt_lo : p.kt_total;

  const __amdgpu_buffer_rsrc_t ra0 = make_rsrc(p.a0, p.a0_bytes);
  const __amdgpu_buffer_rsrc_t ra1 = make_rsrc(p.a1 ? p.a1 : p.a0, p.a1_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  // ---- producer (as in gemm_w320_kernel, two A passes)
  const int rr = tid >> 3, ch = (tid & 7) ^ tile_swz<CPR>(rr);
  const int w_v = (int)(((long)(n0 + (rr >> 5) * 160 + (rr & 31)) * p.ldw + ch * 8) * ES);
  const int w_jstride = __builtin_amdgcn_readfirstlane((int)(32 * p.ldw * ES));
  int a_pos[2], a_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a_pos[i] = 0;
    const int g = min(m0 + i * 64 + rr, p.m - 1);
    if constexpr (MODE == 1) {
      const int hw = p.hout * p.wout, rem = g - (g / hw) * hw, y = rem / p.wout;
      a_pos[i] = (y << 16) | (rem - y * p.wout);
    } else if constexpr (MODE == 2) {
      a_pos[i] = (g / p.hw) % p.frames;
    }
  }
  int p_tap = 0, p_src = 0, p_kc = 0, p_slab = 0;
  if (kt_lo) {                                               // the walk starts inside the K range: (tap, source, k step) of slab kt_lo
    const int per_tap = p.nk0 + p.nk1;
    p_tap = kt_lo / per_tap;
    p_kc = kt_lo - p_tap * per_tap;
    if (p_kc >= p.nk0) { p_src = 1; p_kc -= p.nk0; }
  }
  auto refresh = [&]() {
    const long lda = p_src ? p.lda1 : p.lda0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = m0 + i * 64 + rr;
      bool ok = g < p.m;
      long row = g;
      if constexpr (MODE == 1) {
        const int dy = p_tap / 3 - 1, dx = p_tap - (dy + 1) * 3 - 1;
        const int y = (a_pos[i] >> 16) + dy, x = (a_pos[i] & 0xffff) + dx;
        ok = ok && (unsigned)y < (unsigned)p.hin && (unsigned)x < (unsigned)p.win;
        row = (long)g + dy * p.win + dx;
      } else if constexpr (MODE == 2) {
        ok = ok && (unsigned)(a_pos[i] + p_tap - 1) < (unsigned)p.frames;
        row = (long)g + (long)(p_tap - 1) * p.hw;
      }
      a_off[i] = ok ? (int)((row * lda + ch * 8) * ES) : kInv;
    }
  };
  refresh();
  int p_soff_a = __builtin_amdgcn_readfirstlane(p_kc * 64 * ES);
  int p_soff_w = __builtin_amdgcn_readfirstlane((int)(((long)p_tap * (p.k0 + p.k1) + (p_src ? p.k0 : 0) + p_kc * 64) * ES));
  bool p_ok = S > 0;
  auto advance = [&]() {
    if (++p_kc == (p_src ? p.nk1 : p.nk0)) {
      p_kc = 0;
      if (p_src == 0 && p.nk1 > 0) p_src = 1; else { p_src = 0; ++p_tap; }
      if (MODE != 0 || p.nk1 > 0) refresh();
    }
    ++p_slab;
    p_ok = p_slab < S;
    p_soff_a = __builtin_amdgcn_readfirstlane(p_kc * 64 * ES);
    p_soff_w = __builtin_amdgcn_readfirstlane((int)(((long)p_tap * (p.k0 + p.k1) + (p_src ? p.k0 : 0) + p_kc * 64) * ES));
  };
  auto stage_a = [&](int slot, auto i_tag) {
    constexpr int I = decltype(i_tag)::value;
    char* dst = smem + slot * SLOT + I * 8192 + wid * 1024;
    const int v = p_ok ? a_off[I] : kInv;
    if (__builtin_amdgcn_readfirstlane(p_src))
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra1, (__attribute__((address_space(3))) void*)dst, 16, v, p_soff_a, 0, TT_W320_A_AUX);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra0, (__attribute__((address_space(3))) void*)dst, 16, v, p_soff_a, 0, TT_W320_A_AUX);
  };
  auto stage_w = [&](int slot, auto j_tag) {
    constexpr int J = decltype(j_tag)::value;
    char* dst = smem + slot * SLOT + A_BYTES + J * WJ + wid * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, p_ok ? w_v : kInv,
                                             p_soff_w + J * w_jstride, 0, 0);
  };
  using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>; using C4 = std::integral_constant<int, 4>;

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ln_s[FM], ln_q[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) ln_s[i] = ln_q[i] = 0.f;

  // ---- consumer: the group's two K steps of a slab (chunks 4 grp + 2 ksl + hi of a 128-byte row)
  const unsigned lds_base = lds_addr(smem);
  unsigned a_addr[2], b_addr[2];
  {
    const int ar = wr * 64 + l31, br = wc * 32 + l31;
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      const int c = (grp * 2 + ksl) * 2 + hi;
      a_addr[ksl] = ar * 128 + ((c ^ tile_swz<CPR>(ar)) << 4);
      b_addr[ksl] = A_BYTES + br * 128 + ((c ^ tile_swz<CPR>(br)) << 4);
    }
  }
  raw_u32x4_t af[FM][2], bf[2][2];
  auto read_a = [&](unsigned sb) {
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      af[0][ksl] = lds_read16_raw_off<0>(sb + a_addr[ksl]);
      af[1][ksl] = lds_read16_raw_off<4096>(sb + a_addr[ksl]);
    }
  };
  auto read_b = [&](unsigned sb, auto j_tag, auto n_tag) {          // W_J (and W_{J+1} when N == 2) -> bf[0] (bf[1])
    constexpr int J = decltype(j_tag)::value, N = decltype(n_tag)::value;
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      bf[0][ksl] = lds_read16_raw_off<J * WJ>(sb + b_addr[ksl]);
      if constexpr (N == 2) bf[1][ksl] = lds_read16_raw_off<(J + 1) * WJ>(sb + b_addr[ksl]);
    }
  };
  auto mma = [&](auto j_tag, auto n_tag) {
    constexpr int J = decltype(j_tag)::value, N = decltype(n_tag)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl)
#pragma unroll
      for (int jj = 0; jj < N; ++jj)
#pragma unroll
        for (int i = 0; i < FM; ++i)
          acc[i][J + jj] = Cvt<Tag>::mfma32(make_uint4(bf[jj][ksl].x, bf[jj][ksl].y, bf[jj][ksl].z, bf[jj][ksl].w),
                                            make_uint4(af[i][ksl].x, af[i][ksl].y, af[i][ksl].z, af[i][ksl].w), acc[i][J + jj]);
    __builtin_amdgcn_s_setprio(0);
  };
  // LayerNorm sums: each group sums the K steps it multiplies; the two wave columns of a group hold the same A fragments, each
  // sums one of the group's two K steps; the four partial sums of a row meet in the epilogue
  auto stats = [&]() {
    if constexpr (LNROWS) {
#pragma unroll
      for (int ksl = 0; ksl < 2; ++ksl)
        if (ksl == wc) {
#pragma unroll
          for (int i = 0; i < FM; ++i) ln_stat<Tag>(af[i][ksl], ln_s[i], ln_q[i]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };

  // ---- prologue: slab 0 complete (slot 0) + everything of slab 1 but W_4 (slot 1)
  stage_w(0, C0{}); stage_w(0, C1{}); stage_a(0, C0{}); stage_a(0, C1{}); stage_w(0, C2{}); stage_w(0, C3{}); stage_w(0, C4{});
  advance();
  stage_w(1, C0{}); stage_w(1, C1{}); stage_a(1, C0{}); stage_a(1, C1{}); stage_w(1, C2{}); stage_w(1, C3{});
  w3_wait_vm<6>();
  bar();
  if (grp == 1) bar();

  for (int s = 0; s < S; ++s) {
    const int slot = s & 1;
    const unsigned sb = lds_base + slot * SLOT;
    // ---- P0: A (kept for the slab), W_0, W_1 ; DMA: W_4 of slab s+1 (other slot, read in P2 of slab s-1) closes that slab
    read_a(sb); read_b(sb, C0{}, C2{});
    stage_w(slot ^ 1, C4{});
    advance();
    lds_wait<0>();
    bar();
    mma(C0{}, C2{});
    bar();
    // ---- P1: W_2, W_3 ; DMA: W_0, W_1 and A of slab s+2 (this slot, read in P0)
    read_b(sb, C2{}, C2{});
    stage_w(slot, C0{}); stage_w(slot, C1{}); stage_a(slot, C0{}); stage_a(slot, C1{});
    stats();
    lds_wait<0>();
    bar();
    mma(C2{}, C2{});
    bar();
    // ---- P2: W_4 ; DMA: W_2, W_3 of slab s+2 ; slab s+1 has landed once at most the 6 pieces of P1 / P2 are in flight
    read_b(sb, C4{}, C1{});
    stage_w(slot, C2{}); stage_w(slot, C3{});
    w3_wait_vm<6>();
    lds_wait<0>();
    bar();
    mma(C4{}, C1{});
    bar();
  }
  if (grp == 0) bar();
  w3_wait_vm<0>();
  bar();

  // ---- fused LayerNorm: the partial sums of a row sit in four waves (two groups x two wave columns)
  float rs[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) rs[i] = 1.0f;
  if constexpr (LNROWS) {
    float* xs = (float*)(smem + 65536);                      // [wave][lane][4]: in the exchange buffers, before their first use
    *(float4*)(xs + (wid * 64 + lane) * 4) = make_float4(ln_s[0], ln_q[0], ln_s[1], ln_q[1]);
    __syncthreads();
    const float4 o1 = *(const float4*)(xs + ((wid ^ 1) * 64 + lane) * 4);     // other wave column, this group
    const float4 o2 = *(const float4*)(xs + ((wid ^ 4) * 64 + lane) * 4);     // this wave column, other group
    const float4 o3 = *(const float4*)(xs + ((wid ^ 5) * 64 + lane) * 4);
    __syncthreads();                                         // all partial sums read before the exchange buffers are written
    const float inv_k = 1.0f / (float)p.k0;
    const float s0 = (ln_s[0] + o1.x) + (o2.x + o3.x), q0 = (ln_q[0] + o1.y) + (o2.y + o3.y);
    const float s1 = (ln_s[1] + o1.z) + (o2.z + o3.z), q1 = (ln_q[1] + o1.w) + (o2.w + o3.w);
    const float sm0 = (s0 + __shfl_xor(s0, 32)) * inv_k, sq0 = (q0 + __shfl_xor(q0, 32)) * inv_k;
    const float sm1 = (s1 + __shfl_xor(s1, 32)) * inv_k, sq1 = (q1 + __shfl_xor(q1, 32)) * inv_k;
    rs[0] = rsqrtf(fmaxf(sq0 - sm0 * sm0, 0.f) + p.ln_eps);
    rs[1] = rsqrtf(fmaxf(sq1 - sm1 * sm1, 0.f) + p.ln_eps);
  }

  // ---- the two K halves meet in the epilogue: wave w and its partner w ^ 4 hold partial sums of the same 64 x 160 sub-tile.
  // Group 0 finishes fragment row 0, group 1 fragment row 1; per 64-column chunk each wave hands the fragments of the row it gives
  // away to its partner through an 8 KiB exchange buffer behind the strips (w3_epilogue_rows, XCHG).  Uniform branch on the
  // group: the accumulator indices stay compile-time.
  char* ebuf = smem + wid * 8192;
  char* xbuf = smem + 65536 + wid * 8192;
  const char* pbuf = smem + 65536 + (wid ^ 4) * 8192;
  const int mb = m0 + wr * 64, ncol0 = n0 + wc * 160;
  if (p.splitk > 1) {
    if (grp == 0) w3_partial_row(p, acc[0], ebuf, mb, ncol0, lane, split, acc[1], xbuf, pbuf);
    else w3_partial_row(p, acc[1], ebuf, mb + 32, ncol0, lane, split, acc[0], xbuf, pbuf);
    return;
  }
  float* sst = (float*)(smem + 16 * 8192);                   // statistics staging rows behind the strips and the exchange buffers
  if (grp == 0) w3_epilogue_dispatch<Tag, 1, true>(p, &acc[0], &rs[0], ebuf, mb, ncol0, lane, acc[1], xbuf, pbuf, sst + wid * 320);
  else w3_epilogue_dispatch<Tag, 1, true>(p, &acc[1], &rs[1], ebuf, mb + 32, ncol0, lane, acc[0], xbuf, pbuf, sst + wid * 320);
  // wid = grp * 4 + wr * 2 + wc: the four waves (K half, row half) of a column half, rows in order
  if (p.stats) w3_stats_tile(p, sst, tid, tile_m, n0, 4, [](int k, int half) { return (k & 1) * 4 + (k >> 1) * 2 + half; }, p.stat_rows == 64 ? 2 : 1);
}

template <typename Tag, int MODE, int LNROWS>
static void launch_w320_inst(GemmP& p, hipStream_t st) {
  constexpr int lds = 2 * 73728;                             // two slots (the epilogue strips, the LayerNorm partials and the statistics staging rows reuse them)
  static_assert(lds <= 160 * 1024 && 9 * 8192 + 8 * 320 * 4 <= lds, "w320 LDS");
  static unsigned long long attr_done = 0;
  tt_lds_opt_in((const void*)gemm_w320_kernel<Tag, MODE, LNROWS>, lds, &attr_done);
  hipLaunchKernelGGL((gemm_w320_kernel<Tag, MODE, LNROWS>), dim3(p.tiles_m * p.tiles_n), dim3(512), lds, st, p);
}
template <typename Tag>
static void launch_w320_tag(GemmP& p, hipStream_t st) {
  p.tiles_m = ceil_div(p.m, 256);
  p.tiles_n = p.n / 320;
  p.nk0 = p.k0 / 64; p.nk1 = p.k1 / 64;
  p.kt_total = p.taps * (p.nk0 + p.nk1);
  if (p.mode == 1) launch_w320_inst<Tag, 1, 0>(p, st);
  else if (p.mode == 2) launch_w320_inst<Tag, 2, 0>(p, st);
  else if (p.ln_fold) launch_w320_inst<Tag, 0, 1>(p, st);
  else launch_w320_inst<Tag, 0, 0>(p, st);
}
void launch_w320_bf16(GemmP& p, hipStream_t st) { launch_w320_tag<bf16_tag>(p, st); }
void launch_w320_f16(GemmP& p, hipStream_t st) { launch_w320_tag<f16_tag>(p, st); }

template <typename Tag, int MODE, int LNROWS>
static void launch_w320h_inst(GemmP& p, hipStream_t st) {
  constexpr int lds = 128 * 1024 + 8 * 320 * 4;              // two 56 KiB slots; the epilogue reuses them: 8 strips + 8 exchange buffers of 8 KiB, + the statistics staging rows
  static_assert(lds <= 160 * 1024 && 2 * 57344 <= lds && 16 * 8192 + 8 * 320 * 4 <= lds, "w320h LDS");
  static unsigned long long attr_done = 0;
  tt_lds_opt_in((const void*)gemm_w320h_kernel<Tag, MODE, LNROWS>, lds, &attr_done);
  hipLaunchKernelGGL((gemm_w320h_kernel<Tag, MODE, LNROWS>), dim3(p.tiles_m * p.tiles_n * p.splitk), dim3(512), lds, st, p);
  if (p.splitk > 1) launch_splitk_epilogue<Tag>(p, st);      // second pass: the slabs summed in a fixed order + the full epilogue (+ the tile sums)
}
template <typename Tag>
static void launch_w320h_tag(GemmP& p, hipStream_t st) {
  p.tiles_m = ceil_div(p.m, 128);
  p.tiles_n = p.n / 320;
  p.nk0 = p.k0 / 64; p.nk1 = p.k1 / 64;
  p.kt_total = p.taps * (p.nk0 + p.nk1);
  if (p.mode == 1) launch_w320h_inst<Tag, 1, 0>(p, st);
  else if (p.mode == 2) launch_w320h_inst<Tag, 2, 0>(p, st);
  else if (p.ln_fold) launch_w320h_inst<Tag, 0, 1>(p, st);
  else launch_w320h_inst<Tag, 0, 0>(p, st);
}
void launch_w320h_bf16(GemmP& p, hipStream_t st) { launch_w320h_tag<bf16_tag>(p, st); }
void launch_w320h_f16(GemmP& p, hipStream_t st) { launch_w320h_tag<f16_tag>(p, st); }

}  // namespace ttg
