// tt_conv3x3: 3x3 / stride 1 / pad 1 convolution of the ResBlocks with the GroupNorm (+SiLU) of its INPUT fused in and the
// input staged in LDS as a spatial patch with halo (BASELINE north_star: "conv2d + GroupNorm + SiLU fused ... LDS-staged 3x3
// tiles").  Replaces, per ResnetBlock2D conv:  gn_apply (read x, write a normalised copy)  +  the implicit-GEMM conv that
// re-fetched every input tile once per tap (tt_gemm mode 1).
//
//   out[pixel][n] = bias[n] (+ FiLM[batch][n]) (+ residual[pixel][n]) + sum_{tap, c} act(x[pixel + tap][c]) * W[n][tap][c]
//   act(v) = silu?(v * scale[image][c] + shift[image][c])        (scale/shift from tt_groupnorm_stats; identity if absent)
//
// Structure (256 threads = 4 waves, 2 workgroups per CU):
//   * an output tile is a TH x TW rectangle of ONE image (<= 128 pixels: MFMA rows), all BN output channels of a column
//     tile; wave w owns output rows 32w .. 32w+31, i.e. accumulators of 32 x BN;
//   * K is walked as (64-channel chunk) x (9 taps).  Per chunk the (TH+2) x (TW+2) input patch is fetched ONCE -- raw
//     16-byte global loads into registers, issued half a chunk ahead -- normalised + activated in registers (out-of-image
//     halo pixels become exact zeros AFTER the activation, as the reference's zero padding does) and written to LDS with
//     144-byte rows: the pad makes the 16 rows of a ds_read_b128 lane group hit disjoint banks WITHOUT an address swizzle,
//     so a tap is a constant byte offset (dy * (TW+2) + dx) * 144 on one per-lane base address;
//   * the weight tile of (tap, chunk) goes HBM/L2 -> LDS by LDS-DMA into a double buffer, one tap ahead;
//   * per tap: 4 k sub-steps x BN/32 MFMAs (v_mfma_f32_32x32x16), fragment reads as raw ds_read_b128 retired by counted
//     lgkmcnt waits (see gemm_kernel.h);
//   * epilogue: accumulators transposed through a wave-private LDS strip, then bias / FiLM / residual on 4 consecutive
//     columns per lane, 8-byte stores covering 128 contiguous bytes per pixel.
// LDS-fill traffic per 64 channels of K: one 23..26 KiB patch + 9 W tiles, against 9 x (16 KiB A tile + W tile) before.
#include <type_traits>
#include "gemm_kernel.h"

using namespace ttg;

namespace {

struct ConvP {
  const char* x0; const char* x1; int c0, c1; long ld0, ld1;
  const char* w; long ldw;
  const float* gn_scale; const float* gn_shift; int silu;
  const float* bias; const float* rowvec; int rowvec_rows; long ld_rowvec;
  const char* residual; long ld_res;
  char* out; long ldo;
  int nimg, h, w_, n;
  int tiles_y, tiles_x, tiles_n;
  unsigned x0_bytes, x1_bytes, w_bytes, out_bytes, res_bytes, bias_bytes, rowvec_bytes;
};

constexpr int CP_RS = 144;           // patch row stride: 64 channels x 2 B + 16 B (bank-conflict-free without swizzle)

constexpr int CP_NST = 3;            // W ring depth: two (tap, chunk) tiles in flight behind the one being multiplied

template <typename Tag, int TH, int TW, int BN>
__global__ __launch_bounds__(256, 2) void conv_patch_kernel(const ConvP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PW = TW + 2, PROWS = (TH + 2) * PW;
  constexpr int PATCH = (PROWS * CP_RS + 1023) / 1024 * 1024;
  constexpr int FN = BN / 32, BR = BN * 8 / 256, WST = BR * 4096;     // W stage: BN rows x 128 B
  constexpr int NPI = (PROWS + 31) / 32;                               // patch rows per thread
  constexpr int SLOTS = TH * TW;
  static_assert(SLOTS <= 128 && BN % 32 == 0 && (BN * 8) % 256 == 0, "tile shape");
  static_assert(4 * 8192 <= PATCH + CP_NST * WST, "epilogue strips do not fit");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int bid = blockIdx.x;
  {   // XCD-aware order: consecutive tiles (same patch, neighbouring column tiles / neighbouring patches) share an L2
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = bid % p.tiles_n;
  const int sp = bid / p.tiles_n;
  const int tx0 = (sp % p.tiles_x) * TW, ty0 = ((sp / p.tiles_x) % p.tiles_y) * TH, img = sp / (p.tiles_x * p.tiles_y);
  const int n0 = tile_n * BN;
  const int ctot = p.c0 + p.c1, nchunks = ctot >> 6;

  const __amdgpu_buffer_rsrc_t rx0 = make_rsrc(p.x0, p.x0_bytes);
  const __amdgpu_buffer_rsrc_t rx1 = make_rsrc(p.x1 ? p.x1 : p.x0, p.x1_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  // ---- patch staging geometry: thread = (row lane tid/8, 16-byte segment tid%8), rows row_lane + 32 i
  const int seg = tid & 7, prow0 = tid >> 3;
  int ppix[NPI];                      // pixel index of the patch row inside the tensor, -1 outside the image / patch
#pragma unroll
  for (int i = 0; i < NPI; ++i) {
    const int r = prow0 + 32 * i;
    const int py = r / PW, px = r - py * PW;
    const int y = ty0 - 1 + py, x = tx0 - 1 + px;
    ppix[i] = (r < PROWS && y >= 0 && y < p.h && x >= 0 && x < p.w_) ? (img * p.h + y) * p.w_ + x : -1;
  }
  u32x4_t pre[NPI];
  auto prefetch = [&](int ch) {
    const int kc = ch << 6;
    const bool s1 = kc >= p.c0;                                  // uniform: which source holds this chunk
    const long ld = s1 ? p.ld1 : p.ld0;
    const int coff = ((s1 ? kc - p.c0 : kc) + seg * 8) * 2;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int v = ppix[i] >= 0 ? (int)((long)ppix[i] * ld * 2 + coff) : kInv;
      pre[i] = s1 ? __builtin_amdgcn_raw_buffer_load_b128(rx1, v, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rx0, v, 0, 0);
    }
  };
  auto write_patch = [&](int ch) {
    float sc[8], sh[8];
    if (p.gn_scale) {
      const float* ps = p.gn_scale + (long)img * ctot + (ch << 6) + seg * 8;
      const float* ph = p.gn_shift + (long)img * ctot + (ch << 6) + seg * 8;
      const float4 a = *(const float4*)ps, b = *(const float4*)(ps + 4), c = *(const float4*)ph, d = *(const float4*)(ph + 4);
      sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
      sh[0] = c.x; sh[1] = c.y; sh[2] = c.z; sh[3] = c.w; sh[4] = d.x; sh[5] = d.y; sh[6] = d.z; sh[7] = d.w;
    }
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
      const int r = prow0 + 32 * i;
      if (r < PROWS) {
        uint4 v = make_uint4(pre[i].x, pre[i].y, pre[i].z, pre[i].w);
        if (p.gn_scale) {
          float f[8];
          unpack8<Tag>(v, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = fmaf(f[e], sc[e], sh[e]);
            f[e] = p.silu ? silu_f(t) : t;
          }
          v = ppix[i] >= 0 ? pack8<Tag>(f) : make_uint4(0, 0, 0, 0);      // zero padding is applied AFTER the activation
        }
        *(uint4*)(smem + r * CP_RS + seg * 16) = v;
      }
    }
  };

  // ---- W staging (as gemm_kernel: lane-linear LDS image, XOR swizzle on the source chunk)
  const int crow = tid >> 3, cchunk = tid & 7;
  int vb[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int r = i * 32 + crow, gn = n0 + r;
    vb[i] = gn < p.n ? (int)(((long)gn * p.ldw + (cchunk ^ tile_swz<8>(r)) * 8) * 2) : kInv;
  }
  auto stage_w = [&](int gt, int slot, bool live) {          // gt = chunk * 9 + tap; !live: past the end, issued only to keep vmcnt uniform
    const int ch = gt / 9, tap = gt - ch * 9;
    char* lds_w = smem + PATCH + slot * WST + wid * 1024;
    const int soff = __builtin_amdgcn_readfirstlane((int)(((long)tap * ctot + (ch << 6)) * 2));
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int v = live ? vb[i] : kInv;      // (hipcc: passing the captured array element itself makes the whole kernel template silently un-instantiable)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lds_w + i * 4096), 16, v, soff, 0, 0);
    }
  };

  // ---- fragment addresses
  const int slot = wid * 32 + l31;
  const bool slot_ok = slot < SLOTS;
  const int sy = slot_ok ? slot / TW : 0, sx = slot_ok ? slot - sy * TW : 0;
  const unsigned lds_base = lds_addr(smem);
  const unsigned a_base = lds_base + (sy * PW + sx) * CP_RS + hi * 16;         // tap (0,0) = patch row (sy, sx)
  f32x16_t acc[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int total = nchunks * 9;
  prefetch(0);
  stage_w(0, 0, true);
  stage_w(1, 1, total > 1);
  int ring = 0;                                  // ring slot of tile gt
  for (int gt = 0; gt < total; ++gt) {
    const int ch = gt / 9, tap = gt - ch * 9;
    // W tile gt has landed.  Younger VMEM traffic: tile gt + 1 (BR pieces) and, at taps 5 and 6, the patch prefetch issued at
    // tap 4 (the prefetch of THIS chunk is older than tile gt at tap 0, so the same wait covers write_patch's operands).
    if ((tap == 5 || tap == 6) && ch + 1 < nchunks) wait_vmcnt<BR + NPI>(); else wait_vmcnt<BR>();
    if (tap == 0) {
      __syncthreads();                          // every wave is done with the previous chunk's patch
      write_patch(ch);
    }
    __syncthreads();                            // tile gt visible to all waves; every wave is done with tile gt - 1
    asm volatile("" ::: "memory");
    const int free_slot = ring == 0 ? CP_NST - 1 : ring - 1;
    stage_w(gt + 2, free_slot, gt + 2 < total);
    asm volatile("" ::: "memory");
    if (tap == 4 && ch + 1 < nchunks) prefetch(ch + 1);      // lands under taps 4 .. 8
    const unsigned a_tap = a_base + (unsigned)(((tap / 3) * PW + (tap % 3)) * CP_RS);
    const unsigned w_tap = (unsigned)(ring * WST);
    ring = ring == CP_NST - 1 ? 0 : ring + 1;
    raw_u32x4_t af[2], bf[2][FN];
    auto read = [&](int ks, raw_u32x4_t& a, raw_u32x4_t (&b)[FN]) {
      a = lds_read16_raw(a_tap + ks * 32);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        // chunk (2 ks + hi) ^ swz(row): swz only touches bits that 2 ks + hi may carry, so recompute per ks
        b[j] = lds_read16_raw(lds_base + PATCH + w_tap + tile_off<8>(j * 32 + l31, ks * 2 + hi));
      }
    };
    read(0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) { read(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]); lds_wait<1 + FN>(); }
      else lds_wait<0>();
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const raw_u32x4_t b = bf[ks & 1][j], a = af[ks & 1];
        acc[j] = Cvt<Tag>::mfma32(make_uint4(b.x, b.y, b.z, b.w), make_uint4(a.x, a.y, a.z, a.w), acc[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: lane (row l31, columns j*32 + 8g + 4hi + e) -> wave-private fp32 strip -> lane (row, 4 consecutive columns)
  __syncthreads();                                // all waves are done with the patch / W ring
  const __amdgpu_buffer_rsrc_t r_bias = make_rsrc(p.bias, p.bias_bytes);
  const __amdgpu_buffer_rsrc_t r_rv = make_rsrc(p.rowvec, p.rowvec_bytes);
  const __amdgpu_buffer_rsrc_t r_res = make_rsrc(p.residual, p.res_bytes);
  const __amdgpu_buffer_rsrc_t r_out = make_rsrc(p.out, p.out_bytes);
  auto strip_off = [](int row, int quad, int nq) { return row * 256 + ((quad ^ (row & (nq - 1))) << 4); };
  char* ebuf = smem + wid * 8192;
  const int img_pix0 = img * p.h * p.w_;
  const int rv_row = p.rowvec ? (img_pix0 / p.rowvec_rows) : 0;       // FiLM row of this image's batch element (tile = one image)
#pragma unroll
  for (int jc = 0; jc < FN; jc += 2) {
    const int nfr = (jc + 1 < FN) ? 2 : 1;
    const int q_per_row = nfr * 8, rows_per_pass = 64 / q_per_row, npass = 32 / rows_per_pass;
    const int qq = lane % q_per_row, rr = lane / q_per_row;
    const int gn = n0 + jc * 32 + qq * 4;
    const bool col_ok = gn < p.n;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (jc + jj < FN) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(ebuf + strip_off(l31, jj * 8 + 2 * g + hi, nfr * 8)) =
              make_float4(acc[jc + jj][g * 4], acc[jc + jj][g * 4 + 1], acc[jc + jj][g * 4 + 2], acc[jc + jj][g * 4 + 3]);
      }
    float4 b4 = ld128f(r_bias, col_ok ? gn * 4 : kInv);
    const float4 f4 = ld128f(r_rv, col_ok ? (int)(((long)rv_row * p.ld_rowvec + gn) * 4) : kInv);
    b4.x += f4.x; b4.y += f4.y; b4.z += f4.z; b4.w += f4.w;
    uint2 res[8];
    int off[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      res[ps] = make_uint2(0, 0);
      off[ps] = -1;
      if (ps < npass) {
        const int s = wid * 32 + ps * rows_per_pass + rr;
        if (s < SLOTS && col_ok) {
          const int yy = s / TW, xx = s - yy * TW;
          off[ps] = img_pix0 + (ty0 + yy) * p.w_ + tx0 + xx;
        }
        res[ps] = ld64(r_res, off[ps] >= 0 ? (int)(((long)off[ps] * p.ld_res + gn) * 2) : kInv);
      }
    }
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      if (ps < npass) {
        const int r = ps * rows_per_pass + rr;
        const float4 t = *(const float4*)(ebuf + strip_off(r, qq, q_per_row));
        float r4[4];
        unpack4<Tag>(res[ps], r4);
        const float o0 = t.x + b4.x + r4[0], o1 = t.y + b4.y + r4[1], o2 = t.z + b4.z + r4[2], o3 = t.w + b4.w + r4[3];
        const int ooff = off[ps] >= 0 ? (int)(((long)off[ps] * p.ldo + gn) * 2) : kInv;
        st64(r_out, ooff, pack2<Tag>(o0, o1), pack2<Tag>(o2, o3));
      }
    }
  }
}

template <typename Tag, int TH, int TW, int BN>
void launch_conv(const ConvP& p, hipStream_t st) {
  constexpr int PW = TW + 2, PROWS = (TH + 2) * PW;
  constexpr size_t lds = (size_t)((PROWS * CP_RS + 1023) / 1024 * 1024) + CP_NST * (BN * 8 / 256) * 4096;
  static_assert(lds <= 160 * 1024, "LDS");
  static unsigned long long attr_done = 0;
  tt_lds_opt_in((const void*)conv_patch_kernel<Tag, TH, TW, BN>, (int)lds, &attr_done);
  hipLaunchKernelGGL((conv_patch_kernel<Tag, TH, TW, BN>), dim3(p.nimg * p.tiles_y * p.tiles_x * p.tiles_n), dim3(256), lds, st, p);
}

template <typename Tag, int TH, int TW>
void launch_conv_bn(ConvP& p, int bn, hipStream_t st) {
  p.tiles_y = p.h / TH; p.tiles_x = p.w_ / TW; p.tiles_n = (p.n + bn - 1) / bn;
  if (bn == 160) launch_conv<Tag, TH, TW, 160>(p, st); else launch_conv<Tag, TH, TW, 128>(p, st);
}

// tile shape for an h x w image: a TH x TW rectangle that divides it; 0 = unsupported (the caller falls back)
int pick_tile(int h, int w) {
  if (h % 16 == 0 && w % 8 == 0 && w % 16 != 0) return 1;     // 16 x 8   (32x56, 64x112: widths 56 / 112)
  if (h % 8 == 0 && w % 16 == 0) return 3;                    //  8 x 16
  if (h % 16 == 0 && w % 8 == 0) return 1;
  if (h % 8 == 0 && w % 14 == 0) return 2;                    //  8 x 14  (16x28, 8x14)
  return 0;
}

}  // namespace

extern "C" int tt_conv3x3_supported(int32_t h, int32_t w, int32_t c0, int32_t c1, int32_t n, int32_t dtype) {
  return (dtype == TT_BF16 || dtype == TT_F16) && pick_tile(h, w) != 0 && c0 > 0 && (c0 & 63) == 0 && (c1 & 63) == 0 && n > 0 && (n & 3) == 0;
}

extern "C" int tt_conv3x3(const TtConvArgs* a, tt_stream_t stream) {
  if (!a || !a->x0 || !a->w || !a->out) TT_FAIL(TT_EINVAL, "tt_conv3x3: null operand");
  if (!tt_conv3x3_supported(a->h, a->w_img, a->c0, a->c1, a->n, a->dtype))
    TT_FAIL(TT_EUNSUPPORTED, "tt_conv3x3: %dx%d images with %d+%d -> %d channels are served by tt_gemm mode 1 (+ tt_groupnorm_apply)", a->h, a->w_img, a->c0, a->c1, a->n);
  if (a->c1 && !a->x1) TT_FAIL(TT_EINVAL, "tt_conv3x3: c1 without x1");
  if ((a->gn_scale == nullptr) != (a->gn_shift == nullptr)) TT_FAIL(TT_EINVAL, "tt_conv3x3: gn_scale and gn_shift come together");
  if (a->rowvec && a->rowvec_rows <= 0) TT_FAIL(TT_EINVAL, "tt_conv3x3: rowvec_rows");
  if ((a->ld0 & 7) || (a->c1 && (a->ld1 & 7)) || (a->ldw & 7) || (a->ldo & 3) || (a->residual && (a->ld_res & 3)))
    TT_FAIL(TT_EINVAL, "tt_conv3x3: strides");
  ConvP p;
  p.x0 = (const char*)a->x0; p.x1 = (const char*)a->x1; p.c0 = a->c0; p.c1 = a->c1; p.ld0 = a->ld0; p.ld1 = a->c1 ? a->ld1 : a->ld0;
  p.w = (const char*)a->w; p.ldw = a->ldw;
  p.gn_scale = a->gn_scale; p.gn_shift = a->gn_shift; p.silu = a->silu;
  p.bias = a->bias; p.rowvec = a->rowvec; p.rowvec_rows = a->rowvec_rows; p.ld_rowvec = a->ld_rowvec;
  p.residual = (const char*)a->residual; p.ld_res = a->ld_res;
  p.out = (char*)a->out; p.ldo = a->ldo;
  p.nimg = a->nimg; p.h = a->h; p.w_ = a->w_img; p.n = a->n;
  const long rows = (long)a->nimg * a->h * a->w_img;
  const long x0b = ((rows - 1) * p.ld0 + p.c0) * 2, x1b = p.c1 ? ((rows - 1) * p.ld1 + p.c1) * 2 : 16;
  const long wb = ((long)(p.n - 1) * p.ldw + 9L * (p.c0 + p.c1)) * 2;
  const long outb = ((rows - 1) * p.ldo + p.n) * 2, resb = p.residual ? ((rows - 1) * p.ld_res + p.n) * 2 : 0;
  const long rvb = p.rowvec ? ((long)((rows - 1) / p.rowvec_rows) * p.ld_rowvec + p.n) * 4 : 0;
  if (x0b >= (1L << 31) || x1b >= (1L << 31) || wb >= (1L << 31) || outb >= (1L << 31) || resb >= (1L << 31) || rvb >= (1L << 31))
    TT_FAIL(TT_EUNSUPPORTED, "tt_conv3x3: operand larger than 2 GiB (32-bit buffer offsets)");
  p.x0_bytes = (unsigned)x0b; p.x1_bytes = (unsigned)x1b; p.w_bytes = (unsigned)wb; p.out_bytes = (unsigned)outb;
  p.res_bytes = (unsigned)resb; p.bias_bytes = p.bias ? (unsigned)p.n * 4u : 0u; p.rowvec_bytes = (unsigned)rvb;
  int bn = (p.n % 160 == 0 && p.n % 128 != 0) ? 160 : 128;
  if (const char* e = getenv("TT_CONV_BN")) bn = atoi(e) == 160 ? 160 : 128;      // tuning aid
  hipStream_t st = (hipStream_t)stream;
  const int tile = pick_tile(p.h, p.w_);
#define TT_CV(TAG) do { if (tile == 1) launch_conv_bn<TAG, 16, 8>(p, bn, st); else if (tile == 2) launch_conv_bn<TAG, 8, 14>(p, bn, st); \
                        else launch_conv_bn<TAG, 8, 16>(p, bn, st); } while (0)
  if (a->dtype == TT_BF16) TT_CV(bf16_tag); else TT_CV(f16_tag);
#undef TT_CV
  TT_CHECK_LAUNCH("tt_conv3x3");
  return TT_OK;
}
