// Small dense layers on a handful of rows, sinusoidal embeddings, denoise-loop glue and layout plumbing.
#include <type_traits>
#include "common.h"

namespace {

// y[r][n] = act_out(sum_k act_in(x[r][k]) W[n][k] + bias[n]); one wave per output column n, RT rows at a time.  The RT rows of x are
// staged in LDS once per block with act_in applied (the FiLM projections of a request are 40 320 columns over the same 25 x 1280
// embedding rows: evaluated per column, the SiLU was 1.3 G transcendental pairs and 0.66 ms per launch), and a block walks its columns
// grid-stride over them.  Per column the lanes, the k order and the row order are what they were: results unchanged bit for bit.
template <typename Tag, int RT>
__global__ __launch_bounds__(256) void small_linear_kernel(const float* x, long ldx, int rows, int k, const char* w, long ldw,
                                                           int n, const float* bias, int act_in, int act_out, int accumulate,
                                                           float* y, long ldy) {
  extern __shared__ __attribute__((aligned(16))) float xs[];          // [RT][k]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kv = k >> 3, k4 = k >> 2;
  for (int r0 = 0; r0 < rows; r0 += RT) {
    __syncthreads();                                          // the previous chunk's rows are consumed
    for (int i = threadIdx.x; i < RT * k4; i += 256) {
      const int r = i / k4, c4 = i - r * k4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + r < rows) v = *(const float4*)(x + (long)(r0 + r) * ldx + c4 * 4);
      if (act_in) v = make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w));
      *(float4*)(xs + r * k + c4 * 4) = v;
    }
    __syncthreads();
    for (int col = blockIdx.x * 4 + wv; col < n; col += gridDim.x * 4) {
      float acc[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = 0.f;
      for (int v = lane; v < kv; v += 64) {
        float wf[8];
        load8<Tag>(w + ((long)col * ldw + v * 8) * Elem<Tag>::ES, wf);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const float* xp = xs + r * k + v * 8;
          const float4 a = *(const float4*)xp, b = *(const float4*)(xp + 4);
          const float xf[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[r] = fmaf(xf[e], wf[e], acc[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        float s = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0 && r0 + r < rows) {
          if (bias) s += bias[col];
          if (act_out) s = silu_f(s);
          float* yp = y + (long)(r0 + r) * ldy + col;
          *yp = accumulate ? *yp + s : s;
        }
      }
    }
  }
}

// packing.zero_sum_round on the device: one block per weight row (the host version is ~50 passes of fp64 tensor kernels over every
// LayerNorm-folded matrix: 2.4-2.9 s per process).  Same algorithm, same result bit for bit: q = w rounded to the storage type; r = sum(q)
// (fp64, exact: dyadic terms); binade by binade from the coarsest (`hi`, matrix-wide) down, up to round(|r| / u) elements of the binade --
// the first ones in index order -- move one ulp u against the sign of r; first descent: only elements whose own rounding went that way
// (their other rounding neighbour), second: any element not moved yet.  LDS: the row as fp32 (16-bit values are exact in fp32), its
// binade exponents and flags.  Thread t owns the contiguous elements [t E, (t + 1) E): "the first n" is a block-wide exclusive scan.
template <typename Tag>
__global__ __launch_bounds__(256) void zero_sum_round_kernel(const float* w, long ldw, int k, int hi, int lo, char* out, long ldo) {
  extern __shared__ __attribute__((aligned(16))) char zs_smem[];
  float* q = (float*)zs_smem;                                   // [k]
  short* ex = (short*)(q + k);                                  // [k] binade exponent
  unsigned char* fl = (unsigned char*)(ex + k);                 // [k] bit 0 rounded up, 1 rounded down, 2 non-zero, 3 moved
  __shared__ double red[4];
  __shared__ int cnts[5];
  constexpr int mant = std::is_same<Tag, bf16_tag>::value ? 7 : 10, emin = std::is_same<Tag, bf16_tag>::value ? -126 : -14;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long row = blockIdx.x;
  const int E = (k + 255) / 256, i0 = tid * E, i1 = min(k, i0 + E);
  double rs = 0.0;
  for (int i = i0; i < i1; ++i) {
    const float x = w[row * ldw + i];
    const float v = round_store<Tag>(x);
    q[i] = v;
    int e = v != 0.f ? ilogbf(fabsf(v)) : -1;
    if (e < emin) e = emin;
    ex[i] = (short)e;
    fl[i] = (unsigned char)((v > x ? 1 : 0) | (v < x ? 2 : 0) | (v != 0.f ? 4 : 0));
    rs += (double)v;
  }
  // exact row sum: lanes by shuffle, waves through LDS (all terms dyadic: the order does not matter)
  auto block_sum = [&](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  double r = block_sum(rs);
  const int stop = lo > hi - 48 ? lo : hi - 48;
  for (int pass = 0; pass < 2; ++pass) {
    for (int lvl = hi; lvl >= stop; --lvl) {
      if (r == 0.0) break;                                     // (uniform: every thread holds the same r)
      const double u = ldexp(1.0, lvl - mant);
      const double want = fabs(rint(r / u));
      if (want == 0.0) continue;
      const unsigned char need = pass == 0 ? (r > 0.0 ? 1 : 2) : 0;    // first descent: rounded up (r > 0) / down (r <= 0) elements only
      int c = 0;
      for (int i = i0; i < i1; ++i) {
        const unsigned char f = fl[i];
        c += ((f & 4) && !(f & 8) && ex[i] == lvl && (!need || (f & need))) ? 1 : 0;
      }
      // exclusive scan of the per-thread counts in thread order
      int incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
      __syncthreads();
      if (lane == 63) cnts[wv] = incl;
      __syncthreads();
      int base = incl - c;
      for (int w2 = 0; w2 < wv; ++w2) base += cnts[w2];
      const int total = cnts[0] + cnts[1] + cnts[2] + cnts[3];
      const double nabs = want < (double)total ? want : (double)total;
      const int n = (int)nabs;
      const float step = (float)(r > 0.0 ? u : -u);            // q -= sign(n) u
      int rank = base;
      for (int i = i0; i < i1; ++i) {
        const unsigned char f = fl[i];
        if ((f & 4) && !(f & 8) && ex[i] == lvl && (!need || (f & need))) {
          if (rank < n) { q[i] -= step; fl[i] = f | 8; }
          ++rank;
        }
      }
      r -= (r > 0.0 ? nabs : -nabs) * u;
    }
  }
  for (int i = i0; i < i1; ++i) store1<Tag>(out + (row * ldo + i) * 2, q[i]);
}

__global__ void timestep_embedding_kernel(const float* t, int rows, int dim, float* out, long ldo) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * half) return;
  const int r = i / half, j = i - r * half;
  // freq_j = exp(-ln(10000) * j / half)   (downscale_freq_shift = 0); flip_sin_to_cos -> [cos | sin]
  const float freq = expf(-9.210340371976184f * (float)j / (float)half);
  const float arg = t[r] * freq;
  out[(long)r * ldo + j] = cosf(arg);
  out[(long)r * ldo + half + j] = sinf(arg);
}

template <typename Tag>
__global__ void prep_input_kernel(const float* lat, const float* img, const float* cond, const float* sigmas, int step,
                                  int batch, int frames, int hw, int cpad, char* x) {
  const long total = (long)batch * frames * hw;
  const float sg = sigmas[step];
  const float c_in = 1.0f / sqrtf(sg * sg + 1.0f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % hw);
    const long bf = i / hw;
    const int f = (int)(bf % frames);
    const int b = (int)(bf / frames);
    constexpr int ES = Elem<Tag>::ES;
    char* o = x + i * cpad * ES;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      store1<Tag>(o + c * ES, lat[((long)f * 4 + c) * hw + p] * c_in);
      store1<Tag>(o + (4 + c) * ES, img[(((long)b * frames + f) * 4 + c) * hw + p]);
    }
    int c0 = 8;
    if (cond) {
#pragma unroll
      for (int c = 0; c < 4; ++c) store1<Tag>(o + (8 + c) * ES, cond[((long)f * 4 + c) * hw + p]);
      c0 = 12;
    }
    for (int c = c0; c < cpad; ++c) store1<Tag>(o + c * ES, 0.f);
  }
}

__global__ void cfg_euler_kernel(const float* eps, int ld_eps, float* lat, const float* guidance, const float* sigmas,
                                 int step, int batch, int frames, int hw, float image_guidance) {
  const long total = (long)frames * hw;
  const float sg = sigmas[step], sn = sigmas[step + 1];
  const float c_out = -sg / sqrtf(sg * sg + 1.0f), c_skip = 1.0f / (sg * sg + 1.0f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % hw);
    const int f = (int)(i / hw);
    const float g = guidance ? guidance[f] : 1.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v;
      if (batch == 3) {           // InstructPix2Pix order (reference :698-702): first-frame, cond, uncond
        const float e1 = eps[((long)f * hw + p) * ld_eps + c];
        const float cd = eps[(((long)frames + f) * hw + p) * ld_eps + c];
        const float u = eps[(((long)2 * frames + f) * hw + p) * ld_eps + c];
        v = u + g * (cd - u) + image_guidance * (cd - e1);
      } else if (batch == 2) {
        const float u = eps[((long)f * hw + p) * ld_eps + c];
        const float cd = eps[(((long)frames + f) * hw + p) * ld_eps + c];
        v = u + g * (cd - u);
      } else {
        v = eps[((long)f * hw + p) * ld_eps + c];
      }
      float* xp = lat + ((long)f * 4 + c) * hw + p;
      const float xv = *xp;
      const float x0 = v * c_out + xv * c_skip;
      *xp = xv + (xv - x0) / sg * (sn - sg);
    }
  }
}

// NCHW -> tokens through a 32x32 LDS transpose (coalesced both sides)
template <typename Tag, bool SRC_F32>
__global__ void nchw_to_tokens_kernel(const char* src, int c, int hw, char* dst, long ld_dst) {
  __shared__ float tile[32][33];
  const int img = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: ty 0..7
  for (int i = ty; i < 32; i += 8) {
    const int cc = c0 + i, pp = p0 + tx;
    float v = 0.f;
    if (cc < c && pp < hw) {
      const long idx = ((long)img * c + cc) * hw + pp;
      v = SRC_F32 ? ((const float*)src)[idx] : load1<Tag>(src + idx * Elem<Tag>::ES);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int pp = p0 + i, cc = c0 + tx;
    if (cc < c && pp < hw) store1<Tag>(dst + (((long)img * hw + pp) * ld_dst + cc) * Elem<Tag>::ES, tile[tx][i]);
  }
}

template <typename Tag, bool SRC_F32, bool DST_F32>
__global__ void tokens_to_nchw_kernel(const char* src, long ld_src, int c, int hw, char* dst) {
  __shared__ float tile[32][33];
  const int img = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int pp = p0 + i, cc = c0 + tx;
    float v = 0.f;
    if (cc < c && pp < hw) {
      const long idx = ((long)img * hw + pp) * ld_src + cc;
      v = SRC_F32 ? ((const float*)src)[idx] : load1<Tag>(src + idx * Elem<Tag>::ES);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int cc = c0 + i, pp = p0 + tx;
    if (cc < c && pp < hw) {
      const long idx = ((long)img * c + cc) * hw + pp;
      if (DST_F32) ((float*)dst)[idx] = tile[tx][i];
      else store1<Tag>(dst + idx * Elem<Tag>::ES, tile[tx][i]);
    }
  }
}

template <typename Tag>
__global__ void add_scaled_kernel(const char* a, const char* b, float scale, char* y, long nvec) {
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
    float fa[8], fb[8];
    load8<Tag>(a + v * 8 * Elem<Tag>::ES, fa);
    load8<Tag>(b + v * 8 * Elem<Tag>::ES, fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) fa[e] = fmaf(fb[e], scale, fa[e]);
    store8<Tag>(y + v * 8 * Elem<Tag>::ES, fa);
  }
}

template <typename Tag>
__global__ void add_rowvec_kernel(const char* x, long ldx, int rows, int cv, const float* rv, long ld_rv, int rows_per_vec,
                                  int nvec, char* y, long ldy) {
  const long total = (long)rows * cv;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
    const long row = v / cv;
    const int ch = (int)(v - row * cv) * 8;
    const float* r = rv + (long)((row / rows_per_vec) % nvec) * ld_rv + ch;
    float f[8];
    load8<Tag>(x + (row * ldx + ch) * Elem<Tag>::ES, f);
    const float4 a = *(const float4*)r, b = *(const float4*)(r + 4);
    f[0] += a.x; f[1] += a.y; f[2] += a.z; f[3] += a.w; f[4] += b.x; f[5] += b.y; f[6] += b.z; f[7] += b.w;
    store8<Tag>(y + (row * ldy + ch) * Elem<Tag>::ES, f);
  }
}


// row softmax of fp32 scores (tt_softmax_rows): one wave per row, the row passes through registers once when it fits
// (cols <= 64 * 4 * MAXV), otherwise three passes.  Output in the tag's storage type with `pad` zeroed columns behind it.
template <typename Tag>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, long ldx, int rows, int cols, char* y, long ldy, int cols_pad) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float mx = -INFINITY;
  for (int c = lane * 4; c < cols; c += 256) {
    const float4 v = *(const float4*)(xr + c);
    mx = fmaxf(fmaxf(mx, c + 0 < cols ? v.x : -INFINITY), fmaxf(c + 1 < cols ? v.y : -INFINITY, fmaxf(c + 2 < cols ? v.z : -INFINITY, c + 3 < cols ? v.w : -INFINITY)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const float4 v = *(const float4*)(xr + c);
    const float e[4] = {__expf(v.x - mx), __expf(v.y - mx), __expf(v.z - mx), __expf(v.w - mx)};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (c + k < cols) sum += e[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.0f / sum;
  constexpr int ES = Elem<Tag>::ES;
  char* yr = y + (long)row * ldy * ES;
  for (int c = lane * 4; c < cols_pad; c += 256) {
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) {
      const float4 v = *(const float4*)(xr + c);
      const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) if (c + k < cols) e[k] = __expf(t[k] - mx) * inv;
    }
    *(typename Elem<Tag>::quad_t*)(yr + c * ES) = f32_to_quad<Tag>(e);
  }
}

}  // namespace

extern "C" int tt_add_rowvec(const void* x, int64_t ldx, int32_t rows, int32_t c, const float* rowvec, int64_t ld_rowvec,
                             int32_t rows_per_vec, int32_t nvec, void* y, int64_t ldy, int32_t dtype, tt_stream_t stream) {
  if (!x || !rowvec || !y) TT_FAIL(TT_EINVAL, "tt_add_rowvec: null operand");
  if (rows <= 0 || c <= 0 || (c & 7) || (ldx & 7) || (ldy & 7) || (ld_rowvec & 3) || rows_per_vec <= 0 || nvec <= 0)
    TT_FAIL(TT_EINVAL, "tt_add_rowvec: c and strides must be multiples of 8 (rowvec stride of 4), rows_per_vec / nvec positive");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_add_rowvec: bad dtype");
  const long total = (long)rows * (c >> 3);
  long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
#define TT_ARV(TAG) hipLaunchKernelGGL(add_rowvec_kernel<TAG>, dim3((unsigned)blocks), dim3(256), 0, st, (const char*)x, (long)ldx, rows, c >> 3, rowvec, (long)ld_rowvec, rows_per_vec, nvec, (char*)y, (long)ldy)
  if (dtype == TT_BF16) TT_ARV(bf16_tag); else if (dtype == TT_F16) TT_ARV(f16_tag); else TT_ARV(f32_tag);
#undef TT_ARV
  TT_CHECK_LAUNCH("tt_add_rowvec");
  return TT_OK;
}


extern "C" int tt_softmax_rows(const float* x, int64_t ldx, int32_t rows, int32_t cols, void* y, int64_t ldy, int32_t cols_pad,
                               int32_t dtype, tt_stream_t stream) {
  if (!x || !y) TT_FAIL(TT_EINVAL, "tt_softmax_rows: null operand");
  if (rows <= 0 || cols <= 0 || cols_pad < cols || (cols_pad & 3) || (ldx & 3) || ldx < cols_pad || (ldy & 3) || ldy < cols_pad)
    TT_FAIL(TT_EINVAL, "tt_softmax_rows: cols_pad >= cols, cols_pad / ldx / ldy multiples of 4 and >= cols_pad (rows are read in float4)");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_softmax_rows: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((rows + 3) / 4), block(256);
#define TT_SM(TAG) hipLaunchKernelGGL(softmax_rows_kernel<TAG>, grid, block, 0, st, x, (long)ldx, rows, cols, (char*)y, (long)ldy, cols_pad)
  if (dtype == TT_BF16) TT_SM(bf16_tag); else if (dtype == TT_F16) TT_SM(f16_tag); else TT_SM(f32_tag);
#undef TT_SM
  TT_CHECK_LAUNCH("tt_softmax_rows");
  return TT_OK;
}

extern "C" int tt_small_linear(const float* x, int64_t ldx, int32_t rows, int32_t k, const void* w, int64_t ldw, int32_t n,
                               const float* bias, int32_t act_in, int32_t act_out, int32_t accumulate, float* y, int64_t ldy,
                               int32_t dtype, tt_stream_t stream) {
  if (!x || !w || !y) TT_FAIL(TT_EINVAL, "tt_small_linear: null operand");
  if (rows <= 0 || rows > 32 || n <= 0 || k <= 0 || (k & 7) || (ldx & 3) || (ldw & 7)) TT_FAIL(TT_EINVAL, "tt_small_linear: rows 1..32, k %% 8 == 0, ldx %% 4 == 0");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_small_linear: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  if (k > 8192) TT_FAIL(TT_EUNSUPPORTED, "tt_small_linear: k = %d > 8192 (four fp32 rows of x are staged in 128 KiB of LDS)", k);
  const int nb = (n + 3) / 4;
  const dim3 grid(nb < 1024 ? nb : 1024), block(256);
  const size_t lds = (size_t)4 * k * sizeof(float);
  static unsigned long long attr_done[3] = {0, 0, 0};        // (rows above 64 KiB: the temporal position embedding's 5120-wide hidden layer)
  if (dtype == TT_BF16) tt_lds_opt_in((const void*)small_linear_kernel<bf16_tag, 4>, 131072, &attr_done[0]);
  else if (dtype == TT_F16) tt_lds_opt_in((const void*)small_linear_kernel<f16_tag, 4>, 131072, &attr_done[1]);
  else tt_lds_opt_in((const void*)small_linear_kernel<f32_tag, 4>, 131072, &attr_done[2]);
  if (dtype == TT_BF16)
    hipLaunchKernelGGL((small_linear_kernel<bf16_tag, 4>), grid, block, lds, st, x, (long)ldx, rows, k, (const char*)w, (long)ldw, n, bias, act_in, act_out, accumulate, y, (long)ldy);
  else if (dtype == TT_F16)
    hipLaunchKernelGGL((small_linear_kernel<f16_tag, 4>), grid, block, lds, st, x, (long)ldx, rows, k, (const char*)w, (long)ldw, n, bias, act_in, act_out, accumulate, y, (long)ldy);
  else
    hipLaunchKernelGGL((small_linear_kernel<f32_tag, 4>), grid, block, lds, st, x, (long)ldx, rows, k, (const char*)w, (long)ldw, n, bias, act_in, act_out, accumulate, y, (long)ldy);
  TT_CHECK_LAUNCH("tt_small_linear");
  return TT_OK;
}

extern "C" int tt_zero_sum_round(const float* w, int64_t ldw, int32_t n, int32_t k, int32_t hi, int32_t lo, void* out, int64_t ldo,
                                 int32_t dtype, tt_stream_t stream) {
  if (!w || !out || n <= 0 || k <= 0) TT_FAIL(TT_EINVAL, "tt_zero_sum_round: bad arguments");
  if (dtype != TT_BF16 && dtype != TT_F16) TT_FAIL(TT_EINVAL, "tt_zero_sum_round: 16-bit storage types only");
  if (k > 16384) TT_FAIL(TT_EUNSUPPORTED, "tt_zero_sum_round: k = %d > 16384 (a row lives in LDS)", k);
  const size_t lds = (size_t)k * 7;                          // fp32 value + 16-bit exponent + flag byte
  hipStream_t st = (hipStream_t)stream;
  static unsigned long long attr_done[2] = {0, 0};
  if (dtype == TT_BF16) {
    tt_lds_opt_in((const void*)zero_sum_round_kernel<bf16_tag>, 7 * 16384, &attr_done[0]);
    hipLaunchKernelGGL(zero_sum_round_kernel<bf16_tag>, dim3(n), dim3(256), lds, st, w, (long)ldw, k, hi, lo, (char*)out, (long)ldo);
  } else {
    tt_lds_opt_in((const void*)zero_sum_round_kernel<f16_tag>, 7 * 16384, &attr_done[1]);
    hipLaunchKernelGGL(zero_sum_round_kernel<f16_tag>, dim3(n), dim3(256), lds, st, w, (long)ldw, k, hi, lo, (char*)out, (long)ldo);
  }
  TT_CHECK_LAUNCH("tt_zero_sum_round");
  return TT_OK;
}

extern "C" int tt_timestep_embedding(const float* t, int32_t rows, int32_t dim, float* out, int64_t ldo, tt_stream_t stream) {
  if (!t || !out || rows <= 0 || dim <= 0 || (dim & 1)) TT_FAIL(TT_EINVAL, "tt_timestep_embedding: bad arguments");
  const int total = rows * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, rows, dim, out, (long)ldo);
  TT_CHECK_LAUNCH("tt_timestep_embedding");
  return TT_OK;
}

extern "C" int tt_prep_model_input(const float* latents, const float* image_latents, const float* cond, const float* sigmas,
                                   int32_t step, int32_t batch, int32_t frames, int32_t h, int32_t w, int32_t cpad, void* x,
                                   int32_t dtype, tt_stream_t stream) {
  if (!latents || !image_latents || !sigmas || !x) TT_FAIL(TT_EINVAL, "tt_prep_model_input: null operand");
  if (cpad < (cond ? 12 : 8) || (cpad & 7) || batch <= 0 || frames <= 0 || step < 0) TT_FAIL(TT_EINVAL, "tt_prep_model_input: cpad/batch/frames");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_prep_model_input: bad dtype");
  const long total = (long)batch * frames * h * w;
  long blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TT_BF16) hipLaunchKernelGGL(prep_input_kernel<bf16_tag>, dim3((unsigned)blocks), dim3(256), 0, st, latents, image_latents, cond, sigmas, step, batch, frames, h * w, cpad, (char*)x);
  else if (dtype == TT_F16) hipLaunchKernelGGL(prep_input_kernel<f16_tag>, dim3((unsigned)blocks), dim3(256), 0, st, latents, image_latents, cond, sigmas, step, batch, frames, h * w, cpad, (char*)x);
  else hipLaunchKernelGGL(prep_input_kernel<f32_tag>, dim3((unsigned)blocks), dim3(256), 0, st, latents, image_latents, cond, sigmas, step, batch, frames, h * w, cpad, (char*)x);
  TT_CHECK_LAUNCH("tt_prep_model_input");
  return TT_OK;
}

extern "C" int tt_cfg_euler_step(const float* eps, int32_t ld_eps, float* latents, const float* guidance, const float* sigmas,
                                 int32_t step, int32_t batch, int32_t frames, int32_t h, int32_t w, tt_stream_t stream) {
  if (!eps || !latents || !sigmas) TT_FAIL(TT_EINVAL, "tt_cfg_euler_step: null operand");
  if (batch < 1 || batch > 2 || frames <= 0 || ld_eps < 4 || step < 0) TT_FAIL(TT_EINVAL, "tt_cfg_euler_step: batch must be 1 (no CFG) or 2 (uncond, cond)");
  const long total = (long)frames * h * w;
  long blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, eps, ld_eps, latents, guidance, sigmas, step, batch, frames, h * w, 0.0f);
  TT_CHECK_LAUNCH("tt_cfg_euler_step");
  return TT_OK;
}

extern "C" int tt_cfg3_euler_step(const float* eps, int32_t ld_eps, float* latents, const float* guidance, float image_guidance_scale,
                                  const float* sigmas, int32_t step, int32_t frames, int32_t h, int32_t w, tt_stream_t stream) {
  if (!eps || !latents || !sigmas || !guidance) TT_FAIL(TT_EINVAL, "tt_cfg3_euler_step: null operand");
  if (frames <= 0 || h <= 0 || w <= 0 || ld_eps < 4 || step < 0) TT_FAIL(TT_EINVAL, "tt_cfg3_euler_step: bad shape");
  const long total = (long)frames * h * w;
  long blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, eps, ld_eps, latents, guidance, sigmas, step, 3, frames, h * w, image_guidance_scale);
  TT_CHECK_LAUNCH("tt_cfg3_euler_step");
  return TT_OK;
}

extern "C" int tt_nchw_to_tokens(const void* src, int32_t src_f32, int32_t nimg, int32_t c, int32_t hw, void* dst, int64_t ld_dst,
                                 int32_t dtype, tt_stream_t stream) {
  if (!src || !dst || nimg <= 0 || c <= 0 || hw <= 0 || ld_dst < c) TT_FAIL(TT_EINVAL, "tt_nchw_to_tokens: bad arguments");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_nchw_to_tokens: bad dtype");
  const dim3 grid((hw + 31) / 32, (c + 31) / 32, nimg), block(256);
  hipStream_t st = (hipStream_t)stream;
#define TT_N2T(TAG, F) hipLaunchKernelGGL((nchw_to_tokens_kernel<TAG, F>), grid, block, 0, st, (const char*)src, c, hw, (char*)dst, (long)ld_dst)
  if (dtype == TT_BF16) { if (src_f32) TT_N2T(bf16_tag, true); else TT_N2T(bf16_tag, false); }
  else if (dtype == TT_F16) { if (src_f32) TT_N2T(f16_tag, true); else TT_N2T(f16_tag, false); }
  else TT_N2T(f32_tag, true);
#undef TT_N2T
  TT_CHECK_LAUNCH("tt_nchw_to_tokens");
  return TT_OK;
}

extern "C" int tt_tokens_to_nchw(const void* src, int32_t src_f32, int64_t ld_src, int32_t nimg, int32_t c, int32_t hw, void* dst,
                                 int32_t dst_f32, int32_t dtype, tt_stream_t stream) {
  if (!src || !dst || nimg <= 0 || c <= 0 || hw <= 0 || ld_src < c) TT_FAIL(TT_EINVAL, "tt_tokens_to_nchw: bad arguments");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_tokens_to_nchw: bad dtype");
  const dim3 grid((hw + 31) / 32, (c + 31) / 32, nimg), block(256);
  hipStream_t st = (hipStream_t)stream;
#define TT_T2N(TAG, S, D) hipLaunchKernelGGL((tokens_to_nchw_kernel<TAG, S, D>), grid, block, 0, st, (const char*)src, (long)ld_src, c, hw, (char*)dst)
  if (dtype == TT_BF16) {
    if (src_f32) { if (dst_f32) TT_T2N(bf16_tag, true, true); else TT_T2N(bf16_tag, true, false); }
    else { if (dst_f32) TT_T2N(bf16_tag, false, true); else TT_T2N(bf16_tag, false, false); }
  } else if (dtype == TT_F16) {
    if (src_f32) { if (dst_f32) TT_T2N(f16_tag, true, true); else TT_T2N(f16_tag, true, false); }
    else { if (dst_f32) TT_T2N(f16_tag, false, true); else TT_T2N(f16_tag, false, false); }
  } else {
    TT_T2N(f32_tag, true, true);
  }
#undef TT_T2N
  TT_CHECK_LAUNCH("tt_tokens_to_nchw");
  return TT_OK;
}

extern "C" int tt_add_scaled(const void* a, const void* b, float scale, void* y, int64_t n, int32_t dtype, tt_stream_t stream) {
  if (!a || !b || !y || n <= 0 || (n & 7)) TT_FAIL(TT_EINVAL, "tt_add_scaled: n must be a positive multiple of 8");
  if (dtype != TT_BF16 && dtype != TT_F16 && dtype != TT_F32) TT_FAIL(TT_EINVAL, "tt_add_scaled: bad dtype");
  const long nvec = n >> 3;
  long blocks = (nvec + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TT_BF16) hipLaunchKernelGGL(add_scaled_kernel<bf16_tag>, dim3((unsigned)blocks), dim3(256), 0, st, (const char*)a, (const char*)b, scale, (char*)y, nvec);
  else if (dtype == TT_F16) hipLaunchKernelGGL(add_scaled_kernel<f16_tag>, dim3((unsigned)blocks), dim3(256), 0, st, (const char*)a, (const char*)b, scale, (char*)y, nvec);
  else hipLaunchKernelGGL(add_scaled_kernel<f32_tag>, dim3((unsigned)blocks), dim3(256), 0, st, (const char*)a, (const char*)b, scale, (char*)y, nvec);
  TT_CHECK_LAUNCH("tt_add_scaled");
  return TT_OK;
}
