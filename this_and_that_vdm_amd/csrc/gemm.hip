// tt_gemm: gather-GEMM on gfx950 MFMA (v_mfma_f32_32x32x16_{bf16,f16}).
//
//   out[m][n] = epilogue( sum_{tap,src,c} A_src[rowmap(m,tap)][c] * W[n][(tap,src,c)] )
//
// One kernel family covers nn.Linear, 1x1 conv, 3x3 conv (stride 1/2, fused nearest-x2 upsample),
// and the (3,1,1) temporal conv, with up to two channel sources (skip-concat without a concat
// buffer).  Structure (cdna guide section 5, "minimum 2-phase" form):
//   * BM x BN x 64 tiles, 4 waves, each wave owns (BM/WGM) x (BN/WGN) as 32x32 MFMA fragments;
//   * A and W tiles go HBM -> LDS by 16-byte LDS-DMA (global_load_lds), double buffered, one
//     barrier per K step; out-of-range rows / conv halo / K tails point their lane at a zero page;
//   * LDS image is lane-linear with the XOR swizzle applied on the SOURCE chunk and on the read
//     (conflict-free ds_read_b128, see common.h);
//   * MFMA operands are swapped (W fragment as the MFMA "A" operand) so each lane ends up with 4
//     consecutive output columns of one row: 8-byte epilogue loads/stores.
#include "common.h"

namespace {

struct GemmP {
  const char* a0; const char* a1;
  int k0, k1; long lda0, lda1;
  const char* w; long ldw;
  int m, n, mode;
  int nimg, hin, win, hout, wout, stride, upsample;
  int frames, hw;
  const float* bias; float acc_scale;
  const float* rowvec; int rowvec_rows; long ld_rowvec;
  int geglu;
  const char* residual; long ld_res;
  const char* blend; long ld_blend; float alpha;
  char* out; long ldo; int out_f32;
  int out_col_hw, out_col_hwp;
  int nk0, nk1, taps, kt_total;     // derived: K steps per source, taps, total K steps
  int tiles_m, tiles_n;
};

constexpr int BK = 64;

template <typename Tag, int BM, int BN, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, FM = WTM / 32, FN = WTN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;           // tile rows staged per thread

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware tile order: the dispatcher places block b on XCD b%8; give each XCD a contiguous run of
  // tiles (same A rows, neighbouring W columns) so its private L2 sees the reuse (guide T1, bijective form).
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread staging descriptors
  const int crow = tid >> 3, cchunk = tid & 7;
  int a_src_chunk[AR];          // source chunk (swizzled) per staged row
  long a_rowoff0[AR];           // MODE 0: element offset of the row in a0 (a1 uses lda1)
  int a_img[AR], a_y[AR], a_x[AR];
  bool a_valid[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = i * 32 + crow;
    a_src_chunk[i] = cchunk ^ tile_swz<8>(r);
    const int gm = m0 + r;
    a_valid[i] = gm < p.m;
    const int g = a_valid[i] ? gm : 0;
    if constexpr (MODE == 0) {
      a_rowoff0[i] = (long)g;
      a_img[i] = a_y[i] = a_x[i] = 0;
    } else if constexpr (MODE == 1) {
      const int hwo = p.hout * p.wout;
      a_img[i] = g / hwo;
      const int rem = g - a_img[i] * hwo;
      a_y[i] = rem / p.wout;
      a_x[i] = rem - a_y[i] * p.wout;
      a_rowoff0[i] = 0;
    } else {
      const int bf = g / p.hw;
      a_img[i] = bf % p.frames;   // frame index
      a_y[i] = a_x[i] = 0;
      a_rowoff0[i] = (long)g;
    }
  }
  int b_src_chunk[BR];
  long b_rowoff[BR];
  bool b_valid[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int r = i * 32 + crow;
    b_src_chunk[i] = cchunk ^ tile_swz<8>(r);
    const int gn = n0 + r;
    b_valid[i] = gn < p.n;
    b_rowoff[i] = (long)(b_valid[i] ? gn : 0) * p.ldw;
  }
  const char* zero = (const char*)tt_zero_page;

  // K-step iterator state for the NEXT tile to stage (uniform)
  int s_tap = 0, s_src = 0, s_kc = 0;
  auto stage = [&](int buf) {
    const int ksrc = s_src ? p.k1 : p.k0;
    const char* abase = s_src ? p.a1 : p.a0;
    const long lda = s_src ? p.lda1 : p.lda0;
    const int kbase = s_kc * BK;
    char* lds_a = smem + buf * STAGE + wid * 1024;
    char* lds_b = smem + buf * STAGE + A_BYTES + wid * 1024;
    int dy = 0, dx = 0;
    if constexpr (MODE == 1) { dy = s_tap / 3 - 1; dx = s_tap - (dy + 1) * 3 - 1; }
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int kk = kbase + a_src_chunk[i] * 8;
      bool ok = a_valid[i] && kk < ksrc;
      long row;
      if constexpr (MODE == 0) {
        row = a_rowoff0[i];
      } else if constexpr (MODE == 1) {
        int iy = a_y[i] * p.stride + dy, ix = a_x[i] * p.stride + dx;
        const int hv = p.upsample ? p.hin * 2 : p.hin, wv = p.upsample ? p.win * 2 : p.win;
        ok = ok && iy >= 0 && iy < hv && ix >= 0 && ix < wv;
        if (p.upsample) { iy >>= 1; ix >>= 1; }
        row = ((long)a_img[i] * p.hin + iy) * p.win + ix;
      } else {
        const int f = a_img[i] + s_tap - 1;
        ok = ok && f >= 0 && f < p.frames;
        row = a_rowoff0[i] + (long)(s_tap - 1) * p.hw;
      }
      const char* src = ok ? abase + (row * lda + kk) * 2 : zero;
      glds16(src, lds_a + i * 4096);
    }
    const long wcol = (long)s_tap * (p.k0 + p.k1) + (s_src ? p.k0 : 0) + kbase;
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int kk = b_src_chunk[i] * 8;
      const bool ok = b_valid[i] && (kbase + kk) < ksrc;
      const char* src = ok ? p.w + (b_rowoff[i] + wcol + kk) * 2 : zero;
      glds16(src, lds_b + i * 4096);
    }
    // advance
    if (++s_kc == (s_src ? p.nk1 : p.nk0)) {
      s_kc = 0;
      if (s_src == 0 && p.nk1 > 0) s_src = 1;
      else { s_src = 0; ++s_tap; }
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

  stage(0);
  for (int kt = 0; kt < p.kt_total; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < p.kt_total) stage((kt + 1) & 1);
    const char* sa = smem + (kt & 1) * STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + hi;
      uint4 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = lds_read16(sa, tile_off<8>(a_lds_row[i], chunk));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = lds_read16(sb, tile_off<8>(b_lds_row[j], chunk));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = Cvt<Tag>::mfma32(bf[j], af[i], acc[i][j]);
    }
  }

  // ---- epilogue: lane holds row m = .. + l31, columns n = .. + 8g + 4hi + {0..3} for g = 0..3
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int gm = m0 + wr * WTM + i * 32 + l31;
    if (gm >= p.m) continue;
    const float* rv = p.rowvec ? p.rowvec + (long)(gm / p.rowvec_rows) * p.ld_rowvec : nullptr;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nb = n0 + wc * WTN + j * 32;
      if (p.geglu) {
        // 16-row groups of packed W rows: [8 value | 8 gate]; regs g=0/2 value, g=1/3 gate
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int gn = nb + t * 16 + 4 * hi;          // packed column of the value quad
          if (gn >= p.n) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float val = acc[i][j][(2 * t) * 4 + e], gate = acc[i][j][(2 * t + 1) * 4 + e];
            if (p.bias) { val += p.bias[gn + e]; gate += p.bias[gn + 8 + e]; }
            v[e] = val * gelu_erf_f(gate);
          }
          const int oc = (gn >> 4) * 8 + (gn & 7);
          uint2 o = make_uint2(pack2<Tag>(v[0], v[1]), pack2<Tag>(v[2], v[3]));
          *(uint2*)(p.out + ((long)gm * p.ldo + oc) * 2) = o;
        }
        continue;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int gn = nb + 8 * g + 4 * hi;
        if (gn >= p.n) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e];
        if (p.bias) {
          const float4 b = *(const float4*)(p.bias + gn);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.acc_scale;
        if (rv) {
          const float4 b = *(const float4*)(rv + gn);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.residual) {
          float r4[4];
          unpack4<Tag>(*(const uint2*)(p.residual + ((long)gm * p.ld_res + gn) * 2), r4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += r4[e];
        }
        if (p.blend) {
          float r4[4];
          unpack4<Tag>(*(const uint2*)(p.blend + ((long)gm * p.ld_blend + gn) * 2), r4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = p.alpha * r4[e] + (1.0f - p.alpha) * v[e];
        }
        if (p.out_f32) {
          *(float4*)(p.out + ((long)gm * p.ldo + gn) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (p.out_col_hw > 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = gn + e;
            const long oc = (long)(c / p.out_col_hw) * p.out_col_hwp + (c % p.out_col_hw);
            *(unsigned short*)(p.out + ((long)gm * p.ldo + oc) * 2) = Cvt<Tag>::from_f32(v[e]);
          }
        } else {
          *(uint2*)(p.out + ((long)gm * p.ldo + gn) * 2) = make_uint2(pack2<Tag>(v[0], v[1]), pack2<Tag>(v[2], v[3]));
        }
      }
    }
  }
}

template <typename Tag, int BM, int BN, int WGM, int WGN, int MODE>
void launch_mode(const GemmP& p, hipStream_t st) {
  constexpr size_t lds = 2 * (BM + BN) * 128;
  static bool attr_done = false;     // one flag per kernel instance: opt in to the full dynamic-LDS size once
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<Tag, BM, BN, WGM, WGN, MODE>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_kernel<Tag, BM, BN, WGM, WGN, MODE>), dim3(p.tiles_m * p.tiles_n), dim3(256), lds, st, p);
}

template <typename Tag, int BM, int BN, int WGM, int WGN>
int launch_cfg(GemmP& p, hipStream_t st) {
  p.tiles_m = ceil_div(p.m, BM);
  p.tiles_n = ceil_div(p.n, BN);
  switch (p.mode) {
    case 0: launch_mode<Tag, BM, BN, WGM, WGN, 0>(p, st); break;
    case 1: launch_mode<Tag, BM, BN, WGM, WGN, 1>(p, st); break;
    default: launch_mode<Tag, BM, BN, WGM, WGN, 2>(p, st); break;
  }
  return 0;
}

// Tile choice: the largest tile that still gives the 256 CUs about two waves of workgroups.
void choose_tile(int m, int n, int* bm, int* bn) {
  const long b128 = (long)ceil_div(m, 128) * ceil_div(n, 128);
  const long b12864 = (long)ceil_div(m, 128) * ceil_div(n, 64);
  if (b128 >= 384) { *bm = 128; *bn = 128; }
  else if (b12864 >= 384) { *bm = 128; *bn = 64; }
  else { *bm = 64; *bn = 64; }
}

template <typename Tag>
int launch(GemmP& p, hipStream_t st) {
  int bm, bn;
  choose_tile(p.m, p.n, &bm, &bn);
  if (bm == 128 && bn == 128) return launch_cfg<Tag, 128, 128, 2, 2>(p, st);
  if (bm == 128) return launch_cfg<Tag, 128, 64, 2, 2>(p, st);
  return launch_cfg<Tag, 64, 64, 2, 2>(p, st);
}

}  // namespace

extern "C" int tt_gemm_plan(const TtGemmArgs* a, int32_t* bm, int32_t* bn) {
  if (!a || !bm || !bn || a->m <= 0 || a->n <= 0) TT_FAIL(TT_EINVAL, "tt_gemm_plan: bad arguments");
  int x, y;
  choose_tile(a->m, a->n, &x, &y);
  *bm = x; *bn = y;
  return TT_OK;
}

extern "C" int tt_gemm(const TtGemmArgs* a, tt_stream_t stream) {
  if (!a || !a->a0 || !a->w || !a->out) TT_FAIL(TT_EINVAL, "tt_gemm: null operand");
  if (a->m <= 0 || a->n <= 0 || a->k0 <= 0) TT_FAIL(TT_EINVAL, "tt_gemm: empty problem m=%d n=%d k0=%d", a->m, a->n, a->k0);
  if ((a->k0 & 7) || (a->k1 & 7) || (a->n & 3)) TT_FAIL(TT_EINVAL, "tt_gemm: k0/k1 must be multiples of 8 and n of 4");
  if ((a->lda0 & 7) || (a->k1 && (a->lda1 & 7)) || (a->ldw & 7)) TT_FAIL(TT_EINVAL, "tt_gemm: row strides must be multiples of 8 elements");
  if (a->k1 && !a->a1) TT_FAIL(TT_EINVAL, "tt_gemm: k1 > 0 without a1");
  if (a->mode < 0 || a->mode > 2) TT_FAIL(TT_EINVAL, "tt_gemm: bad mode %d", a->mode);
  if (a->dtype != TT_BF16 && a->dtype != TT_F16) TT_FAIL(TT_EINVAL, "tt_gemm: bad dtype");
  if (a->geglu && ((a->n & 15) || a->residual || a->blend || a->rowvec || a->out_f32 || a->out_col_hw))
    TT_FAIL(TT_EINVAL, "tt_gemm: geglu needs n %% 16 == 0 and no other epilogue terms");
  if (a->rowvec && a->rowvec_rows <= 0) TT_FAIL(TT_EINVAL, "tt_gemm: rowvec_rows");
  GemmP p;
  p.a0 = (const char*)a->a0; p.a1 = (const char*)a->a1; p.k0 = a->k0; p.k1 = a->k1;
  p.lda0 = a->lda0; p.lda1 = a->lda1; p.w = (const char*)a->w; p.ldw = a->ldw;
  p.m = a->m; p.n = a->n; p.mode = a->mode;
  p.nimg = a->nimg; p.hin = a->hin; p.win = a->win; p.hout = a->hout; p.wout = a->wout;
  p.stride = a->stride; p.upsample = a->upsample; p.frames = a->frames; p.hw = a->hw;
  p.bias = a->bias; p.acc_scale = a->acc_scale;
  p.rowvec = a->rowvec; p.rowvec_rows = a->rowvec_rows; p.ld_rowvec = a->ld_rowvec;
  p.geglu = a->geglu; p.residual = (const char*)a->residual; p.ld_res = a->ld_res;
  p.blend = (const char*)a->blend; p.ld_blend = a->ld_blend; p.alpha = a->alpha;
  p.out = (char*)a->out; p.ldo = a->ldo; p.out_f32 = a->out_f32;
  p.out_col_hw = a->out_col_hw; p.out_col_hwp = a->out_col_hwp;
  if (p.mode == 1) {
    if (p.nimg <= 0 || p.hin <= 0 || p.win <= 0 || p.hout <= 0 || p.wout <= 0 || p.stride < 1)
      TT_FAIL(TT_EINVAL, "tt_gemm: conv geometry");
    if ((long)p.nimg * p.hout * p.wout != p.m) TT_FAIL(TT_EINVAL, "tt_gemm: m != nimg*hout*wout");
  }
  if (p.mode == 2) {
    if (p.frames <= 0 || p.hw <= 0 || p.m % ((long)p.frames * p.hw)) TT_FAIL(TT_EINVAL, "tt_gemm: tconv geometry");
  }
  p.taps = p.mode == 1 ? 9 : (p.mode == 2 ? 3 : 1);
  p.nk0 = ceil_div(p.k0, BK); p.nk1 = p.k1 ? ceil_div(p.k1, BK) : 0;
  p.kt_total = p.taps * (p.nk0 + p.nk1);
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == TT_BF16) launch<bf16_tag>(p, st); else launch<f16_tag>(p, st);
  TT_CHECK_LAUNCH("tt_gemm");
  return TT_OK;
}
