// GroupNorm (statistics -> per-(image,channel) affine -> apply+SiLU) and LayerNorm for token-major tensors.
// All statistics in fp32 with fp64 block/group combination (inputs reach |x| ~ 1e2 after sigma scaling).
#include "common.h"

namespace {

constexpr int GN_GROUPS = 32;
// rows of one image a partial-statistics block covers: 8 per thread row-lane (two rounds of 4 loads in flight).  With a
// fixed 128 rows the wide levels (C = 1280: one row-lane per block) walked 128 rows serially and launched 28 blocks:
// 12 us per call whatever the tensor size.
__host__ __device__ inline int gn_rows_per_chunk(int C) {
  int rpb = 256 / (C >> 3);
  if (rpb < 1) rpb = 1;
  const int rows = rpb * 8;
  return rows > 128 ? 128 : rows;
}

// partial sums per (image, row-chunk, group): ws[((img*chunks + chunk)*32 + g)*2 + {0,1}] (double)
template <typename Tag>
__global__ void gn_partial_kernel(const char* x0, int c0, const char* x1, int c1, int hw, int chunks, double* ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // per-(row-lane, channel) partial sums, reduced in a fixed order: results are bit-reproducible
  const int C = c0 + c1, cv = C >> 3;    // 8-channel vectors per row
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int rpb = nthr / cv;             // rows handled in parallel
  float* psum = (float*)smem;            // [rpb][C]
  float* psq = psum + rpb * C;           // [rpb][C]
  const int myv = tid % cv, myr = tid / cv;
  const int rows_per_chunk = gn_rows_per_chunk(C);
  const int r0 = chunk * rows_per_chunk, r1 = min(hw, r0 + rows_per_chunk);
  if (myr < rpb) {
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    const int ch = myv * 8;
    const char* base; long ld; int coff;
    if (ch < c0) { base = x0; ld = c0; coff = ch; } else { base = x1; ld = c1; coff = ch - c0; }
    constexpr int ES = Elem<Tag>::ES;
    const char* pbase = base + (((long)img * hw) * ld + coff) * ES;
    const long rstride = ld * ES;
    int r = r0 + myr;
    for (; r + 3 * rpb < r1; r += 4 * rpb) {
      float f[4][8];                 // 4 independent row loads in flight per thread (the kernel is a pure HBM stream)
#pragma unroll
      for (int k = 0; k < 4; ++k) load8<Tag>(pbase + (long)(r + k * rpb) * rstride, f[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[k][e]; q[e] = fmaf(f[k][e], f[k][e], q[e]); }
      }
    }
    for (; r < r1; r += rpb) {
      float f[8];
      load8<Tag>(pbase + (long)r * rstride, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] = fmaf(f[e], f[e], q[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { psum[myr * C + ch + e] = s[e]; psq[myr * C + ch + e] = q[e]; }
  }
  __syncthreads();
  if (tid < GN_GROUPS) {
    const int cpg = C / GN_GROUPS;
    double a = 0.0, b = 0.0;
    for (int r = 0; r < rpb; ++r)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += (double)psum[r * C + c]; b += (double)psq[r * C + c]; }
    double* o = ws + (((long)img * chunks + chunk) * GN_GROUPS + tid) * 2;
    o[0] = a; o[1] = b;
  }
}

// one block per statistics group-set: (image-group ig) -> images [ig*fpg, (ig+1)*fpg).
// 1024 threads = 32 groups x 32 slices; each slice sums a strided subset of the (frame, chunk) partials, then the
// 32 slices are combined in a fixed order (deterministic).  (The cross-frame statistics of the temporal ResBlocks reduce
// frames x chunks ~ 500 partials in 2 blocks: the slice count is what bounds this kernel's latency.)
constexpr int GN_SLICES = 32;
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const double* ws, int chunks, int hw, int C, int fpg,
                                                          const float* gamma, const float* beta, float eps,
                                                          float* scale, float* shift) {
  __shared__ double s_a[GN_SLICES][GN_GROUPS], s_b[GN_SLICES][GN_GROUPS];
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  const int ig = blockIdx.x, tid = threadIdx.x;
  const int g = tid & 31, slice = tid >> 5;
  {
    double a = 0.0, b = 0.0;
    const int total = fpg * chunks;
    const double* base = ws + ((long)ig * fpg * chunks) * GN_GROUPS * 2;
    // this kernel is two blocks of pure latency in front of every temporal GroupNorm's apply pass: keep 8 partials (16-byte loads)
    // in flight per thread instead of one dependent round trip per partial; the summation order is unchanged
    int i = slice;
    for (; i + 7 * GN_SLICES < total; i += 8 * GN_SLICES) {
      double2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *(const double2*)(base + ((long)(i + k * GN_SLICES) * GN_GROUPS + g) * 2);
#pragma unroll
      for (int k = 0; k < 8; ++k) { a += v[k].x; b += v[k].y; }
    }
    for (; i < total; i += GN_SLICES) {
      const double2 v = *(const double2*)(base + ((long)i * GN_GROUPS + g) * 2);
      a += v.x; b += v.y;
    }
    s_a[slice][g] = a; s_b[slice][g] = b;
  }
  __syncthreads();
  if (tid < GN_GROUPS) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int s = 0; s < GN_SLICES; ++s) { a += s_a[s][tid]; b += s_b[s][tid]; }
    const double cnt = (double)fpg * hw * (C / GN_GROUPS);
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int cpg = C / GN_GROUPS;
  for (int c = tid; c < C; c += blockDim.x) {
    const int gg = c / cpg;
    const float sc = s_rstd[gg] * gamma[c];
    const float sh = beta[c] - s_mean[gg] * sc;
    for (int f = 0; f < fpg; ++f) {
      scale[((long)ig * fpg + f) * C + c] = sc;
      shift[((long)ig * fpg + f) * C + c] = sh;
    }
  }
}

// Small images (at most GN_ONE_BYTES per image, per-image statistics): ONE kernel, one 1024-thread block per image -- the
// partial + finalize pair costs two dependent launches (~5 us each in the step graph) for tensors a single CU streams in a
// few microseconds.  Thread = (row lane, 8-channel vector); fp32 sums per thread over its rows (8 loads in flight), combined
// per group in fp64 in a fixed order (bit-reproducible), then scale/shift for every channel of the image.
constexpr long GN_ONE_BYTES = 640 * 1024;
// With y != nullptr the kernel also APPLIES the norm (tt_groupnorm_small): gridDim.y = S blocks per image, every one of them
// computes the image's statistics (the same fixed-order arithmetic: identical results; the S-fold re-read is served by L2) and
// then normalises + activates its own 1/S of the rows -- one launch instead of three for tensors whose GroupNorm is launch-bound.
template <typename Tag>
__global__ __launch_bounds__(1024) void gn_stats_image_kernel(const char* x0, int c0, const char* x1, int c1, int hw, int rpb,
                                                              const float* gamma, const float* beta, float eps,
                                                              float* scale, float* shift, char* y, long ldy, int silu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kernarg_touch<104>();
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  const int C = c0 + c1, cv = C >> 3;
  const int img = blockIdx.x, tid = threadIdx.x;
  float* psum = (float*)smem;            // [rpb][C]
  float* psq = psum + rpb * C;           // [rpb][C]
  const int myv = tid % cv, myr = tid / cv;
  if (myr < rpb) {
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    const int ch = myv * 8;
    const char* base; long ld; int coff;
    if (ch < c0) { base = x0; ld = c0; coff = ch; } else { base = x1; ld = c1; coff = ch - c0; }
    constexpr int ES = Elem<Tag>::ES;
    const char* pbase = base + (((long)img * hw) * ld + coff) * ES;
    const long rstride = ld * ES;
    int r = myr;
    for (; r + 7 * rpb < hw; r += 8 * rpb) {
      float f[8][8];
#pragma unroll
      for (int k = 0; k < 8; ++k) load8<Tag>(pbase + (long)(r + k * rpb) * rstride, f[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[k][e]; q[e] = fmaf(f[k][e], f[k][e], q[e]); }
      }
    }
    for (; r < hw; r += rpb) {
      float f[8];
      load8<Tag>(pbase + (long)r * rstride, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] = fmaf(f[e], f[e], q[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { psum[myr * C + ch + e] = s[e]; psq[myr * C + ch + e] = q[e]; }
  }
  __syncthreads();
  const int cpg = C / GN_GROUPS;
  {
    // group g is summed by the 32 lanes tid = g*32 .. g*32+31 (slices of its rpb*cpg values), then combined by a wave-level
    // fixed-order tree (xor shuffles: the same association on every run)
    const int g = tid >> 5, sl = tid & 31;
    double a = 0.0, b = 0.0;
    const int n = rpb * cpg;
    for (int i = sl; i < n; i += 32) {
      const int r = i / cpg, c = g * cpg + (i - r * cpg);
      a += (double)psum[r * C + c]; b += (double)psq[r * C + c];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (sl == 0) {
      const double cnt = (double)hw * cpg;
      const double mean = a / cnt;
      double var = b / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  if (y == nullptr) {
    for (int c = tid; c < C; c += 1024) {
      const int gg = c / cpg;
      const float sc = s_rstd[gg] * gamma[c];
      scale[(long)img * C + c] = sc;
      shift[(long)img * C + c] = beta[c] - s_mean[gg] * sc;
    }
    return;
  }
  if (myr < rpb) {
    const int ch = myv * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int gg = (ch + e) / cpg;
      sc[e] = s_rstd[gg] * gamma[ch + e];
      sh[e] = beta[ch + e] - s_mean[gg] * sc[e];
    }
    const char* base; long ld; int coff;
    if (ch < c0) { base = x0; ld = c0; coff = ch; } else { base = x1; ld = c1; coff = ch - c0; }
    constexpr int ES = Elem<Tag>::ES;
    const char* pbase = base + (((long)img * hw) * ld + coff) * ES;
    char* ybase = y + (((long)img * hw) * ldy + ch) * ES;
    const int S = gridDim.y, part = blockIdx.y;
    const int r_lo = (int)((long)hw * part / S), r_hi = (int)((long)hw * (part + 1) / S);
    for (int r = r_lo + myr; r < r_hi; r += rpb) {
      float f[8];
      load8<Tag>(pbase + (long)r * ld * ES, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = fmaf(f[e], sc[e], sh[e]);
        f[e] = silu ? silu_f(t) : t;
      }
      store8<Tag>(ybase + (long)r * ldy * ES, f);
    }
  }
}

template <typename Tag>
__global__ void gn_apply_kernel(const char* x0, int c0, const char* x1, int c1, int hw, long total_vec,
                                const float* scale, const float* shift, int silu, char* y, long ldy) {
  const int C = c0 + c1, cv = C >> 3;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < total_vec; v += (long)gridDim.x * blockDim.x) {
    const long row = v / cv;
    const int ch = (int)(v - row * cv) * 8;
    const int img = (int)(row / hw);
    constexpr int ES = Elem<Tag>::ES;
    const char* src = ch < c0 ? x0 + (row * c0 + ch) * ES : x1 + (row * c1 + (ch - c0)) * ES;
    float f[8];
    load8<Tag>(src, f);
    const float4 s0 = *(const float4*)(scale + (long)img * C + ch), s1 = *(const float4*)(scale + (long)img * C + ch + 4);
    const float4 h0 = *(const float4*)(shift + (long)img * C + ch), h1 = *(const float4*)(shift + (long)img * C + ch + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = fmaf(f[e], sc[e], sh[e]);
      f[e] = silu ? silu_f(t) : t;
    }
    store8<Tag>(y + (row * ldy + ch) * ES, f);
  }
}

// LayerNorm: one wave per row, values kept in registers (C <= 64*8*MAXV)
template <typename Tag, int MAXV>
__global__ __launch_bounds__(256) void ln_kernel(const char* x, long ldx, int rows, int c, const float* gamma,
                                                 const float* beta, float eps, const float* rowvec, int rows_per_vec,
                                                 int nvec, char* xsum, char* y, long ldy) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cv = c >> 3;
  float v[MAXV][8];
  float s = 0.f;
  const float* rv = rowvec ? rowvec + (long)((row / rows_per_vec) % nvec) * c : nullptr;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + i * 64;
    if (vi < cv) {
      load8<Tag>(x + ((long)row * ldx + vi * 8) * Elem<Tag>::ES, v[i]);
      if (rv) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = round_store<Tag>(v[i][e] + rv[vi * 8 + e]);   // normalise what xsum stores
        store8<Tag>(xsum + ((long)row * ldx + vi * 8) * Elem<Tag>::ES, v[i]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + i * 64;
    if (vi < cv) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / c + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + i * 64;
    if (vi < cv) {
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = (v[i][e] - mean) * rstd * gamma[vi * 8 + e] + beta[vi * 8 + e];
      store8<Tag>(y + ((long)row * ldy + vi * 8) * Elem<Tag>::ES, o8);
    }
  }
}


// ---- per-image GroupNorm of ANY size in one launch, without any exchange between blocks: the statistics of a group need only
// that group's channels, so block (image, slice) owns `gpb` consecutive groups = `nv` 8-channel vectors of every row: it sums
// them (fp32 per thread and channel, fp64 per group in a fixed order: bit-reproducible), then re-reads its slice -- still in L2 --
// normalises, activates and stores it.  x crosses HBM once in each direction; statistics, finalize and apply are one launch.
// The slices of one image go to ONE XCD (linear block L: XCD L % 8 runs images L % 8 + 8 k), whose L2 then serves the sectors
// that neighbouring slices share (a 10-channel group is 20 bytes: slices are not sector-aligned).
template <typename Tag>
__global__ __launch_bounds__(512) void gn_group_kernel(const char* x0, int c0, const char* x1, int c1, int nimg, int hw, int gpb, int nslices,
                                                       int rlanes, const float* gamma, const float* beta, float eps, int silu,
                                                       char* y, long ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  kernarg_touch<88>();
  const int C = c0 + c1, cpg = C / GN_GROUPS;
  const int L = blockIdx.x, xcd = L & 7, t = L >> 3;
  const int slice = t % nslices, img = xcd + 8 * (t / nslices);
  if (img >= nimg) return;
  const int nch = gpb * cpg, nv = nch >> 3, ch_lo = slice * nch;
  const int tid = threadIdx.x;
  const int myv = tid % nv, myr = tid / nv;                 // vector of the slice, row lane
  float* psum = (float*)smem;                               // [rlanes][nch]
  float* psq = psum + rlanes * nch;
  constexpr int ES = Elem<Tag>::ES;
  const int ch = ch_lo + myv * 8;
  const char* base; long ld; int coff;
  if (ch < c0) { base = x0; ld = c0; coff = ch; } else { base = x1; ld = c1; coff = ch - c0; }
  const char* pbase = base + (((long)img * hw) * ld + coff) * ES;
  const long rstride = ld * ES;
  if (myr < rlanes) {
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    int r = myr;
    for (; r + 7 * rlanes < hw; r += 8 * rlanes) {
      float f[8][8];
#pragma unroll
      for (int k = 0; k < 8; ++k) load8<Tag>(pbase + (long)(r + k * rlanes) * rstride, f[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[k][e]; q[e] = fmaf(f[k][e], f[k][e], q[e]); }
      }
    }
    for (; r < hw; r += rlanes) {
      float f[8];
      load8<Tag>(pbase + (long)r * rstride, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] = fmaf(f[e], f[e], q[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { psum[myr * nch + myv * 8 + e] = s[e]; psq[myr * nch + myv * 8 + e] = q[e]; }
  }
  __syncthreads();
  {
    // group g of the slice <- 32 lanes (slices of its rlanes * cpg partial sums), combined by a fixed-order xor tree
    const int g = tid >> 5, sl = tid & 31;
    if (g < gpb) {
      double a = 0.0, b = 0.0;
      const int n = rlanes * cpg;
      for (int i = sl; i < n; i += 32) {
        const int r = i / cpg, c = g * cpg + (i - r * cpg);
        a += (double)psum[r * nch + c]; b += (double)psq[r * nch + c];
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      if (sl == 0) {
        const double cnt = (double)hw * cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[g] = (float)mean;
        s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
      }
    }
  }
  __syncthreads();
  if (myr < rlanes) {
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int gg = (myv * 8 + e) / cpg;
      sc[e] = s_rstd[gg] * gamma[ch + e];
      sh[e] = beta[ch + e] - s_mean[gg] * sc[e];
    }
    char* ybase = y + (((long)img * hw) * ldy + ch) * ES;
    const long ystride = ldy * ES;
    int r = myr;
    for (; r + 3 * rlanes < hw; r += 4 * rlanes) {
      float f[4][8];
#pragma unroll
      for (int k = 0; k < 4; ++k) load8<Tag>(pbase + (long)(r + k * rlanes) * rstride, f[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = fmaf(f[k][e], sc[e], sh[e]);
          f[k][e] = silu ? silu_f(v) : v;
        }
        store8<Tag>(ybase + (long)(r + k * rlanes) * ystride, f[k]);
      }
    }
    for (; r < hw; r += rlanes) {
      float f[8];
      load8<Tag>(pbase + (long)r * rstride, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = fmaf(f[e], sc[e], sh[e]);
        f[e] = silu ? silu_f(v) : v;
      }
      store8<Tag>(ybase + (long)r * ystride, f);
    }
  }
}
// ---- GroupNorm from the PRODUCER's tile sums (tt_gemm stats_out; round 5): the conv / Linear launch that wrote x also left, per tile
// of R rows and per channel, the sum and the sum of squares of what it stored.  A segment (one image, or the frames x hw rows of one
// video for the temporal ResBlock's cross-frame statistics) is a whole number of tiles, so its statistics are a sum over seg_rows / R
// tile rows -- a few KB instead of a pass over the tensor.  Block (row part of a segment, slice of gpb groups): adds the tile sums of its
// channels (fp64, fixed order: bit-reproducible), folds channels into groups, then streams ITS rows once: normalise, activate, store.
// x crosses HBM once in each direction and there is one launch per GroupNorm, for per-image and cross-frame statistics alike
// (before: two passes in one launch, or three launches).  Slices of one row part go to ONE XCD, as in gn_group_kernel.
template <typename Tag>
__global__ __launch_bounds__(512) void gn_tiles_kernel(const char* x, int C, const float* stats, int R, int seg_rows, int nunits, int parts,
                                                       int gpb, int nslices, int rlanes, const float* gamma, const float* beta, float eps,
                                                       int silu, char* y, long ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  kernarg_touch<96>();
  const int cpg = C / GN_GROUPS;
  const int L = blockIdx.x, xcd = L & 7, t = L >> 3;
  const int slice = t % nslices, unit = xcd + 8 * (t / nslices);
  if (unit >= nunits) return;
  const int seg = unit / parts, part = unit - seg * parts;
  const int nch = gpb * cpg, nv = nch >> 3, ch_lo = slice * nch;
  const int tid = threadIdx.x;
  constexpr int ES = Elem<Tag>::ES;
  // ---- statistics: tile rows of this segment, channels of this slice.  Thread (channel c, lane kl of KL) adds tiles kl, kl + KL, ..
  double* pd = (double*)smem;                                // [KL][nch][2], then the channel totals in row 0
  const int T = seg_rows / R, KL = 512 / nch;
  {
    const int c = tid % nch, kl = tid / nch;
    if (kl < KL) {
      double a = 0.0, b = 0.0;
      const float* base = stats + ((long)seg * T * 2) * C + ch_lo + c;
      for (int k = kl; k < T; k += KL) { a += (double)base[(long)k * 2 * C]; b += (double)base[(long)k * 2 * C + C]; }
      pd[(kl * nch + c) * 2] = a; pd[(kl * nch + c) * 2 + 1] = b;
    }
  }
  __syncthreads();
  if (tid < nch) {
    double a = 0.0, b = 0.0;
    for (int kl = 0; kl < KL; ++kl) { a += pd[(kl * nch + tid) * 2]; b += pd[(kl * nch + tid) * 2 + 1]; }
    pd[tid * 2] = a; pd[tid * 2 + 1] = b;                    // (row 0 of the table: only thread `tid` read these two entries)
  }
  __syncthreads();
  if (tid < gpb) {
    double a = 0.0, b = 0.0;
    for (int c = 0; c < cpg; ++c) { a += pd[(tid * cpg + c) * 2]; b += pd[(tid * cpg + c) * 2 + 1]; }
    const double cnt = (double)seg_rows * cpg, mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  // ---- apply: rows [r_lo, r_hi) of the segment, this slice's channels
  const int myv = tid % nv, myr = tid / nv;
  if (myr >= rlanes) return;
  const int ch = ch_lo + myv * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int gg = (myv * 8 + e) / cpg;
    sc[e] = s_rstd[gg] * gamma[ch + e];
    sh[e] = beta[ch + e] - s_mean[gg] * sc[e];
  }
  const int per = (seg_rows + parts - 1) / parts, r_lo = part * per, r_hi = min(seg_rows, r_lo + per);
  const char* pbase = x + (((long)seg * seg_rows) * C + ch) * ES;
  char* ybase = y + (((long)seg * seg_rows) * ldy + ch) * ES;
  const long rstride = (long)C * ES, ystride = ldy * ES;
  int r = r_lo + myr;
  for (; r + 3 * rlanes < r_hi; r += 4 * rlanes) {
    float f[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) load8<Tag>(pbase + (long)(r + k * rlanes) * rstride, f[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = fmaf(f[k][e], sc[e], sh[e]);
        f[k][e] = silu ? silu_f(v) : v;
      }
      store8<Tag>(ybase + (long)(r + k * rlanes) * ystride, f[k]);
    }
  }
  for (; r < r_hi; r += rlanes) {
    float f[8];
    load8<Tag>(pbase + (long)r * rstride, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = fmaf(f[e], sc[e], sh[e]);
      f[e] = silu ? silu_f(v) : v;
    }
    store8<Tag>(ybase + (long)r * ystride, f);
  }
}

// groups per block: the smallest count whose channels fill whole 8-channel vectors, doubled towards 128-byte row slices while an
// image keeps at least 8 slices (16 x 28 x 1280: 80-byte slices 47 us, 160-byte slices 26 us).  0: the channel counts do not allow it (a source boundary inside a vector cannot happen: c0 is a multiple of 8).
// Independent of the image count, so tt_groupnorm_small_supported can answer for the launch.
static int gn_group_gpb(int C, int es) {
  const int cpg = C / GN_GROUPS;
  int gpb = 1;
  while ((gpb * cpg) & 7) gpb *= 2;                          // <= 8
  while (gpb * 2 <= 4 && gpb * cpg * es < 128) gpb *= 2;
  if (gpb > 16 || (gpb * cpg) >> 3 > 512) return 0;          // 32 lanes per group in the combination; one thread per vector at least
  return gpb;
}
// rows: below, the per-group combination of the row lanes' sums dominates a block that owns three rows per lane (measured: 112 x 2560
// 41 us against 26 us for the one-block-per-image kernel, 28 x 2560 23 vs 11); above, an image's 32 / gpb blocks walk too many rows
// each and the multi-launch route with its thousands of blocks streams better (64 x 112 x 320: 112 us against 84 us; 32 x 56 x 320:
// 26 against 29 -- the slices' 80..240-byte row segments reach ~2.5 TB/s, full rows 3 to 4.5)
constexpr int GN_GROUPED_MIN_ROWS = 256, GN_GROUPED_MAX_ROWS = 4096;

// the per-image kernel keeps [rpb][C] x 2 fp32 partials in dynamic LDS (at most 96 KiB): ONE opt-in (per device) with that maximum
constexpr int GN_IMAGE_LDS_MAX = 96 * 1024;
template <typename Tag> static void gn_image_opt_in() {
  static unsigned long long done = 0;
  tt_lds_opt_in((const void*)gn_stats_image_kernel<Tag>, GN_IMAGE_LDS_MAX, &done);
}

}  // namespace

extern "C" size_t tt_groupnorm_ws_bytes(int32_t nimg, int32_t hw, int32_t c) {
  if (c < 8) c = 8;
  const int rows = gn_rows_per_chunk(c);
  const long chunks = (hw + rows - 1) / rows;
  return (size_t)nimg * chunks * GN_GROUPS * 2 * sizeof(double);
}

extern "C" int tt_groupnorm_stats(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t nimg, int32_t hw,
                                  int32_t fpg, const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                  void* ws, size_t ws_bytes, int32_t dtype, tt_stream_t stream) {
  const int C = c0 + c1;
  if (!x0 || !gamma || !beta || !scale || !shift || !ws) TT_FAIL(TT_EINVAL, "tt_groupnorm_stats: null operand");
  if (c1 && !x1) TT_FAIL(TT_EINVAL, "tt_groupnorm_stats: c1 without x1");
  if (nimg <= 0 || hw <= 0 || C <= 0 || (C % GN_GROUPS) || (c0 & 7) || (c1 & 7)) TT_FAIL(TT_EINVAL, "tt_groupnorm_stats: C=%d+%d must be a multiple of 32, sources of 8", c0, c1);
  if (fpg <= 0 || nimg % fpg) TT_FAIL(TT_EINVAL, "tt_groupnorm_stats: frames_per_group %d does not divide %d images", fpg, nimg);
  if (ws_bytes < tt_groupnorm_ws_bytes(nimg, hw, C)) TT_FAIL(TT_EINVAL, "tt_groupnorm_stats: workspace too small");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_groupnorm_stats: bad dtype");
  const int cv = C >> 3;
  if (cv > 1024) TT_FAIL(TT_EUNSUPPORTED, "tt_groupnorm_stats: C=%d too wide", C);
  const int rows_per_chunk = gn_rows_per_chunk(C);
  const int chunks = (hw + rows_per_chunk - 1) / rows_per_chunk;
  int rpb = 256 / cv; if (rpb < 1) rpb = 1;
  int threads = cv * rpb; if (threads < GN_GROUPS) threads = GN_GROUPS;
  threads = (threads + 63) / 64 * 64;
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype == TT_F32 ? 4 : 2;
  if (fpg == 1 && (long)hw * C * es <= GN_ONE_BYTES && cv <= 1024) {
    int rpb1 = 1024 / cv;
    while (rpb1 > 1 && (size_t)2 * rpb1 * C * sizeof(float) > 96 * 1024) --rpb1;     // [rpb][C] x 2 fp32 in LDS
    if (rpb1 > hw) rpb1 = hw;
    const size_t lds1 = (size_t)2 * rpb1 * C * sizeof(float);
#define TT_GN1(TAG, IDX) do { gn_image_opt_in<TAG>(); \
      hipLaunchKernelGGL(gn_stats_image_kernel<TAG>, dim3(nimg), dim3(1024), lds1, st, (const char*)x0, c0, (const char*)x1, c1, hw, rpb1, gamma, beta, eps, scale, shift, (char*)nullptr, 0L, 0); } while (0)
    if (dtype == TT_BF16) TT_GN1(bf16_tag, 0); else if (dtype == TT_F16) TT_GN1(f16_tag, 1); else TT_GN1(f32_tag, 2);
#undef TT_GN1
    TT_CHECK_LAUNCH("tt_groupnorm_stats");
    return TT_OK;
  }
  const size_t lds = (size_t)2 * rpb * C * sizeof(float);
  if (dtype == TT_BF16)
    hipLaunchKernelGGL(gn_partial_kernel<bf16_tag>, dim3(chunks, nimg), dim3(threads), lds, st, (const char*)x0, c0, (const char*)x1, c1, hw, chunks, (double*)ws);
  else if (dtype == TT_F16)
    hipLaunchKernelGGL(gn_partial_kernel<f16_tag>, dim3(chunks, nimg), dim3(threads), lds, st, (const char*)x0, c0, (const char*)x1, c1, hw, chunks, (double*)ws);
  else
    hipLaunchKernelGGL(gn_partial_kernel<f32_tag>, dim3(chunks, nimg), dim3(threads), lds, st, (const char*)x0, c0, (const char*)x1, c1, hw, chunks, (double*)ws);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(nimg / fpg), dim3(32 * GN_SLICES), 0, st, (const double*)ws, chunks, hw, C, fpg, gamma, beta, eps, scale, shift);
  TT_CHECK_LAUNCH("tt_groupnorm_stats");
  return TT_OK;
}

// ---- statistics + apply in one launch for small images (see gn_stats_image_kernel)
static int gn_small_rpb(int C, int hw) {
  const int cv = C >> 3;
  int rpb = 1024 / cv;
  while (rpb > 1 && (size_t)2 * rpb * C * sizeof(float) > 96 * 1024) --rpb;
  if (rpb > hw) rpb = hw;
  return rpb;
}
static int g_gn_grouped = -1;          // TT_GN_GROUPED=0: the round-2 routes (A/B)
static bool gn_grouped_on() {
  if (g_gn_grouped < 0) { const char* e = getenv("TT_GN_GROUPED"); g_gn_grouped = e ? atoi(e) : 1; }
  return g_gn_grouped != 0;
}
extern "C" int tt_groupnorm_small_supported(int32_t hw, int32_t c, int32_t dtype) {
  const int es = dtype == TT_F32 ? 4 : 2;
  if (hw <= 0 || c <= 0 || (c % GN_GROUPS) || (c & 7)) return 0;
  if (gn_grouped_on() && hw >= GN_GROUPED_MIN_ROWS && hw <= GN_GROUPED_MAX_ROWS && gn_group_gpb(c, es) > 0) return 1;   // one block per (image, group slice)
  return (c >> 3) <= 1024 && (long)hw * c * es <= GN_ONE_BYTES;
}
extern "C" int tt_groupnorm_small(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t nimg, int32_t hw,
                                  const float* gamma, const float* beta, float eps, int32_t silu, void* y, int64_t ldy,
                                  int32_t dtype, tt_stream_t stream) {
  const int C = c0 + c1;
  if (!x0 || !gamma || !beta || !y) TT_FAIL(TT_EINVAL, "tt_groupnorm_small: null operand");
  if (c1 && !x1) TT_FAIL(TT_EINVAL, "tt_groupnorm_small: c1 without x1");
  if (nimg <= 0 || (c0 & 7) || (c1 & 7) || (ldy & 7) || ldy < C) TT_FAIL(TT_EINVAL, "tt_groupnorm_small: channel counts/stride must be multiples of 8");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_groupnorm_small: bad dtype");
  if (!tt_groupnorm_small_supported(hw, C, dtype))
    TT_FAIL(TT_EUNSUPPORTED, "tt_groupnorm_small: %d x %d per image is served by tt_groupnorm_stats + tt_groupnorm_apply", hw, C);
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype == TT_F32 ? 4 : 2;
  if (const int gpb = gn_grouped_on() && hw >= GN_GROUPED_MIN_ROWS && hw <= GN_GROUPED_MAX_ROWS ? gn_group_gpb(C, es) : 0) {
    const int nslices = GN_GROUPS / gpb, nv = (gpb * (C / GN_GROUPS)) >> 3;
    int rlanes = 512 / nv;
    if (rlanes > hw) rlanes = hw;
    const size_t ldsg = (size_t)2 * rlanes * nv * 8 * sizeof(float);
    const int blocks = ((nimg + 7) / 8) * 8 * nslices;
#define TT_GNG(TAG) hipLaunchKernelGGL(gn_group_kernel<TAG>, dim3(blocks), dim3(512), ldsg, st, (const char*)x0, c0, (const char*)x1, c1, nimg, hw, gpb, nslices, \
                                       rlanes, gamma, beta, eps, (int)silu, (char*)y, (long)ldy)
    if (dtype == TT_BF16) TT_GNG(bf16_tag); else if (dtype == TT_F16) TT_GNG(f16_tag); else TT_GNG(f32_tag);
#undef TT_GNG
    TT_CHECK_LAUNCH("tt_groupnorm_small");
    return TT_OK;
  }
  const int rpb = gn_small_rpb(C, hw);
  const size_t lds = (size_t)2 * rpb * C * sizeof(float);
  // blocks per image: enough to put the apply phase on ~4 x 28..56 CUs without re-reading the image too often
  int parts = 4;
  if (hw < parts * rpb) parts = hw / rpb > 0 ? hw / rpb : 1;
#define TT_GNS(TAG, IDX) do { gn_image_opt_in<TAG>(); \
    hipLaunchKernelGGL(gn_stats_image_kernel<TAG>, dim3(nimg, parts), dim3(1024), lds, st, (const char*)x0, c0, (const char*)x1, c1, hw, rpb, \
                       gamma, beta, eps, (float*)nullptr, (float*)nullptr, (char*)y, (long)ldy, (int)silu); } while (0)
  if (dtype == TT_BF16) TT_GNS(bf16_tag, 0); else if (dtype == TT_F16) TT_GNS(f16_tag, 1); else TT_GNS(f32_tag, 2);
#undef TT_GNS
  TT_CHECK_LAUNCH("tt_groupnorm_small");
  return TT_OK;
}

extern "C" int tt_groupnorm_tiles_supported(int32_t seg_rows, int32_t c, int32_t stat_rows, int32_t dtype) {
  const int es = dtype == TT_F32 ? 4 : 2;
  if (seg_rows <= 0 || c <= 0 || stat_rows <= 0 || (c % GN_GROUPS) || (c & 7) || seg_rows % stat_rows) return 0;
  return gn_group_gpb(c, es) > 0 ? 1 : 0;
}
extern "C" int tt_groupnorm_tiles(const void* x, int32_t c, const float* stats, int32_t stat_rows, int32_t nseg, int32_t seg_rows,
                                  const float* gamma, const float* beta, float eps, int32_t silu, void* y, int64_t ldy, int32_t dtype,
                                  tt_stream_t stream) {
  if (!x || !stats || !gamma || !beta || !y) TT_FAIL(TT_EINVAL, "tt_groupnorm_tiles: null operand");
  if (nseg <= 0 || (ldy & 7) || ldy < c) TT_FAIL(TT_EINVAL, "tt_groupnorm_tiles: segments / output stride");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_groupnorm_tiles: bad dtype");
  if (!tt_groupnorm_tiles_supported(seg_rows, c, stat_rows, dtype))
    TT_FAIL(TT_EUNSUPPORTED, "tt_groupnorm_tiles: %d rows per segment must be a multiple of the %d statistics rows, C = %d a multiple of 32", seg_rows, stat_rows, c);
  const int es = dtype == TT_F32 ? 4 : 2;
  const int gpb = gn_group_gpb(c, es), nslices = GN_GROUPS / gpb, nch = gpb * (c / GN_GROUPS), nv = nch >> 3;
  // row parts per segment: ~2 blocks per CU over the launch, at least 128 rows each
  int parts = (512 + nseg * nslices - 1) / (nseg * nslices);
  if (parts > seg_rows / 128) parts = seg_rows / 128;
  if (parts < 1) parts = 1;
  const int per = (seg_rows + parts - 1) / parts;
  int rlanes = 512 / nv;
  if (rlanes > per) rlanes = per;
  const int nunits = nseg * parts, blocks = ((nunits + 7) / 8) * 8 * nslices;
  const size_t lds = (size_t)(512 / nch) * nch * 2 * sizeof(double);
  hipStream_t st = (hipStream_t)stream;
#define TT_GNT(TAG) hipLaunchKernelGGL(gn_tiles_kernel<TAG>, dim3(blocks), dim3(512), lds, st, (const char*)x, (int)c, stats, (int)stat_rows, (int)seg_rows, \
                                       nunits, parts, gpb, nslices, rlanes, gamma, beta, eps, (int)silu, (char*)y, (long)ldy)
  if (dtype == TT_BF16) TT_GNT(bf16_tag); else if (dtype == TT_F16) TT_GNT(f16_tag); else TT_GNT(f32_tag);
#undef TT_GNT
  TT_CHECK_LAUNCH("tt_groupnorm_tiles");
  return TT_OK;
}

extern "C" int tt_groupnorm_apply(const void* x0, int32_t c0, const void* x1, int32_t c1, int32_t nimg, int32_t hw,
                                  const float* scale, const float* shift, int32_t silu, void* y, int64_t ldy, int32_t dtype,
                                  tt_stream_t stream) {
  const int C = c0 + c1;
  if (!x0 || !scale || !shift || !y) TT_FAIL(TT_EINVAL, "tt_groupnorm_apply: null operand");
  if (c1 && !x1) TT_FAIL(TT_EINVAL, "tt_groupnorm_apply: c1 without x1");
  if ((c0 & 7) || (c1 & 7) || (ldy & 7) || ldy < C) TT_FAIL(TT_EINVAL, "tt_groupnorm_apply: channel counts/stride must be multiples of 8");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_groupnorm_apply: bad dtype");
  const long total = (long)nimg * hw * (C >> 3);
  long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TT_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<bf16_tag>, dim3((unsigned)blocks), dim3(256), 0, st, (const char*)x0, c0, (const char*)x1, c1, hw, total, scale, shift, silu, (char*)y, (long)ldy);
  else if (dtype == TT_F16)
    hipLaunchKernelGGL(gn_apply_kernel<f16_tag>, dim3((unsigned)blocks), dim3(256), 0, st, (const char*)x0, c0, (const char*)x1, c1, hw, total, scale, shift, silu, (char*)y, (long)ldy);
  else
    hipLaunchKernelGGL(gn_apply_kernel<f32_tag>, dim3((unsigned)blocks), dim3(256), 0, st, (const char*)x0, c0, (const char*)x1, c1, hw, total, scale, shift, silu, (char*)y, (long)ldy);
  TT_CHECK_LAUNCH("tt_groupnorm_apply");
  return TT_OK;
}

extern "C" int tt_layernorm(const void* x, int64_t ldx, int32_t rows, int32_t c, const float* gamma, const float* beta,
                            float eps, const float* rowvec, int32_t rows_per_vec, int32_t nvec, void* xsum_out, void* y,
                            int64_t ldy, int32_t dtype, tt_stream_t stream) {
  if (!x || !gamma || !beta || !y) TT_FAIL(TT_EINVAL, "tt_layernorm: null operand");
  if (rows <= 0 || c <= 0 || (c & 7) || (ldx & 7) || (ldy & 7)) TT_FAIL(TT_EINVAL, "tt_layernorm: c and strides must be multiples of 8");
  if (rowvec && (!xsum_out || rows_per_vec <= 0 || nvec <= 0)) TT_FAIL(TT_EINVAL, "tt_layernorm: rowvec needs xsum_out, rows_per_vec, nvec");
  if (c > 64 * 8 * 4) TT_FAIL(TT_EUNSUPPORTED, "tt_layernorm: c=%d > 2048", c);
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_layernorm: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((rows + 3) / 4), block(256);
#define TT_LN(TAG, MV) hipLaunchKernelGGL((ln_kernel<TAG, MV>), grid, block, 0, st, (const char*)x, (long)ldx, rows, c, gamma, beta, eps, rowvec, rows_per_vec, nvec, (char*)xsum_out, (char*)y, (long)ldy)
  const int mv = (c / 8 + 63) / 64;
  if (dtype == TT_BF16) { if (mv <= 1) TT_LN(bf16_tag, 1); else if (mv <= 2) TT_LN(bf16_tag, 2); else TT_LN(bf16_tag, 4); }
  else if (dtype == TT_F16) { if (mv <= 1) TT_LN(f16_tag, 1); else if (mv <= 2) TT_LN(f16_tag, 2); else TT_LN(f16_tag, 4); }
  else { if (mv <= 1) TT_LN(f32_tag, 1); else if (mv <= 2) TT_LN(f32_tag, 2); else TT_LN(f32_tag, 4); }
#undef TT_LN
  TT_CHECK_LAUNCH("tt_layernorm");
  return TT_OK;
}
