// Shared device/host helpers for libttvdm (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "ttvdm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct bf16_tag {};
struct f16_tag {};
struct f32_tag {};     // reference-precision mode (TT_F32): fp32 storage, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)

// storage geometry of a tag: bytes per element, elements per 16-byte chunk (the unit of LDS-DMA staging and of one
// MFMA operand read), and the register type holding 4 consecutive elements
template <typename Tag> struct Elem { static constexpr int ES = 2, EPC = 8; typedef uint2 quad_t; };
template <> struct Elem<f32_tag> { static constexpr int ES = 4, EPC = 4; typedef uint4 quad_t; };

// 16 zero bytes (x4 for safety) that out-of-range tile lanes point their global source at.
static __device__ __attribute__((aligned(64))) unsigned int tt_zero_page[64] = {0};   // per-TU copy: no -fgpu-rdc needed

void tt_set_error(const char* fmt, ...);
#define TT_FAIL(code, ...) do { tt_set_error(__VA_ARGS__); return (code); } while (0)
// Kernels that need more than the default 64 KiB of dynamic LDS opt in with hipFuncSetAttribute.  The attribute is per
// DEVICE, so the "done" flag is a per-device bit mask (one mask per kernel instance), and a failure is kept in
// tt_attr_err until the entry point's TT_CHECK_LAUNCH reports it (a launch without the opt-in would fail anyway).
static thread_local hipError_t tt_attr_err = hipSuccess;
static inline void tt_lds_opt_in(const void* fn, int bytes, unsigned long long* done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev >= 0 && dev < 64 && ((*done_mask >> dev) & 1ull)) return;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { tt_attr_err = e; return; }
  if (dev >= 0 && dev < 64) *done_mask |= 1ull << dev;
}
#define TT_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (tt_attr_err != hipSuccess) { e_ = tt_attr_err; tt_attr_err = hipSuccess; } \
    if (e_ != hipSuccess) TT_FAIL(TT_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

// ---------------------------------------------------------------- scalar conversions
template <typename Tag> struct Cvt;
template <> struct Cvt<bf16_tag> {
  static __device__ __forceinline__ float to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
  static __device__ __forceinline__ unsigned short from_f32(float f) {  // round-to-nearest-even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned short, (__bf16)f);
  }
  typedef float f32x4m_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4m_t mfma16(uint4 a, uint4 b, f32x4m_t c) {      // 16 x 16 x 32
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16_t mfma32(uint4 a, uint4 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Cvt<f16_tag> {
  static __device__ __forceinline__ float to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
  static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
  typedef float f32x4m_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x4m_t mfma16(uint4 a, uint4 b, f32x4m_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16_t mfma32(uint4 a, uint4 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

// fp32 "conversion" is the identity; one 16-byte chunk holds 4 k-values, consumed by 4 exact-fp32 MFMAs (K = 2 each:
// lane (l31, hi) supplies k-slot hi).  Bitwise an fmaf chain in k order (MI355X guide, f32-input MFMA).
template <> struct Cvt<f32_tag> {
  static __device__ __forceinline__ f32x16_t mfma32(uint4 a, uint4 b, f32x16_t c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

// two floats -> one dword of two 16-bit values, one instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, RNE)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <typename Tag> __device__ __forceinline__ unsigned pack2(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack2<bf16_tag>(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
template <> __device__ __forceinline__ unsigned pack2<f16_tag>(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){lo, hi}, f16x2_t));
}
template <typename Tag> __device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = Cvt<Tag>::to_f32((unsigned short)(w[i] & 0xffffu));
    f[2 * i + 1] = Cvt<Tag>::to_f32((unsigned short)(w[i] >> 16));
  }
}
template <typename Tag> __device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2<Tag>(f[0], f[1]), pack2<Tag>(f[2], f[3]), pack2<Tag>(f[4], f[5]), pack2<Tag>(f[6], f[7]));
}
template <typename Tag> __device__ __forceinline__ void unpack4(const uint2& v, float* f) {
  f[0] = Cvt<Tag>::to_f32((unsigned short)(v.x & 0xffffu));
  f[1] = Cvt<Tag>::to_f32((unsigned short)(v.x >> 16));
  f[2] = Cvt<Tag>::to_f32((unsigned short)(v.y & 0xffffu));
  f[3] = Cvt<Tag>::to_f32((unsigned short)(v.y >> 16));
}

// ---- tag-generic element access (16-bit tags: packed halves; f32_tag: plain floats)
template <typename Tag> __device__ __forceinline__ void quad_to_f32(const typename Elem<Tag>::quad_t& q, float* f) { unpack4<Tag>(q, f); }
template <> __device__ __forceinline__ void quad_to_f32<f32_tag>(const uint4& q, float* f) {
  f[0] = __uint_as_float(q.x); f[1] = __uint_as_float(q.y); f[2] = __uint_as_float(q.z); f[3] = __uint_as_float(q.w);
}
template <typename Tag> __device__ __forceinline__ typename Elem<Tag>::quad_t f32_to_quad(const float* f) {
  return make_uint2(pack2<Tag>(f[0], f[1]), pack2<Tag>(f[2], f[3]));
}
template <> __device__ __forceinline__ uint4 f32_to_quad<f32_tag>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
// one 16-byte chunk (EPC elements) <-> floats
template <typename Tag> __device__ __forceinline__ uint4 pack_chunk(const float* f) { return pack8<Tag>(f); }
template <> __device__ __forceinline__ uint4 pack_chunk<f32_tag>(const float* f) { return f32_to_quad<f32_tag>(f); }
template <typename Tag> __device__ __forceinline__ void unpack_chunk(const uint4& v, float* f) { unpack8<Tag>(v, f); }
template <> __device__ __forceinline__ void unpack_chunk<f32_tag>(const uint4& v, float* f) { quad_to_f32<f32_tag>(v, f); }
// 8 consecutive elements at byte address p (16-byte aligned)
template <typename Tag> __device__ __forceinline__ void load8(const char* p, float* f) { unpack8<Tag>(*(const uint4*)p, f); }
template <> __device__ __forceinline__ void load8<f32_tag>(const char* p, float* f) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 16);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename Tag> __device__ __forceinline__ void store8(char* p, const float* f) { *(uint4*)p = pack8<Tag>(f); }
template <> __device__ __forceinline__ void store8<f32_tag>(char* p, const float* f) {
  *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
  *(float4*)(p + 16) = make_float4(f[4], f[5], f[6], f[7]);
}
template <typename Tag> __device__ __forceinline__ float load1(const char* p) { return Cvt<Tag>::to_f32(*(const unsigned short*)p); }
template <> __device__ __forceinline__ float load1<f32_tag>(const char* p) { return *(const float*)p; }
template <typename Tag> __device__ __forceinline__ void store1(char* p, float v) { *(unsigned short*)p = Cvt<Tag>::from_f32(v); }
template <> __device__ __forceinline__ void store1<f32_tag>(char* p, float v) { *(float*)p = v; }
// the value a store of v followed by a load would return (storage rounding; identity for fp32)
template <typename Tag> __device__ __forceinline__ float round_store(float v) { return Cvt<Tag>::to_f32(Cvt<Tag>::from_f32(v)); }
template <> __device__ __forceinline__ float round_store<f32_tag>(float v) { return v; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below fp16/bf16 output rounding):
//   gelu(x) = x Phi(x),  Phi(x >= 0) = 1 - u,  Phi(x < 0) = u,  u = 0.5 (a1 t + .. + a5 t^5) exp(-x^2 / 2),  t = 1 / (1 + p |x| / sqrt 2)
// written for the issue slots it costs -- the GEGLU epilogue of the persistent GEMM is VALU-bound on it (64 per lane and tile):
// the 1/sqrt 2, 0.5 and log2(e) factors are folded into constants, exp is one v_exp_f32 of a product, the sign is one select:
// 12 plain VALU + v_rcp + v_exp instead of 18 + 2 (libm erff: ~40).
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));   // v_rcp_f32 (1 ulp): erf error stays ~1e-7
  float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  poly = fmaf(poly, t, 0.5f * 1.421413741f);
  poly = fmaf(poly, t, 0.5f * -0.284496736f);
  poly = fmaf(poly, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);      // exp(-x^2 / 2): argument <= 0, no range fix-ups needed
  const float u = poly * t * e;                                                // 0.5 erfc(|x| / sqrt 2)
  return x * (x >= 0.f ? 1.0f - u : u);
}

// The same function on two values at once, written on 2-vectors so that the arithmetic issues as packed fp32 (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: two lanes' worth per instruction); only |x|, the two transcendentals and the sign transfer are
// per element.  Phi(x) = 0.5 + copysign(0.5 - u, x): the same u as above, no compare / select.  Bitwise it may differ from
// gelu_erf_f in the last ulp (1 - u is formed as 0.5 + (0.5 - u)); the erf error bound is unchanged (tests: 1.5e-7 absolute).
typedef float f32pk_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32pk_t gelu_erf_pk(f32pk_t x) {
  const f32pk_t ax = {fabsf(x.x), fabsf(x.y)};
  const f32pk_t d = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
  const f32pk_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32pk_t poly = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
  poly = poly * t + (0.5f * 1.421413741f);
  poly = poly * t + (0.5f * -0.284496736f);
  poly = poly * t + (0.5f * 0.254829592f);
  const f32pk_t a = x * x * -0.72134752044448170f;
  const f32pk_t e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  const f32pk_t h = 0.5f - poly * t * e;                                     // 0.5 - u  (>= 0)
  const f32pk_t sg = {__builtin_copysignf(h.x, x.x), __builtin_copysignf(h.y, x.y)};
  return x * (sg + 0.5f);
}

// GELU for the GEGLU epilogue of the persistent kernel (16-bit storage only), round 5: the normal CDF as a logistic function of an odd
// polynomial,   Phi(x) ~ 1 / (1 + exp(-x P(x^2))),  P of degree 4 in x^2 (minimax fit over |x| <= 6.5: max |dPhi| = 1.44e-6, i.e.
// |x dPhi| <= 6.3e-6 in fp32 arithmetic -- three orders below bf16 / fp16 storage rounding of the outputs it feeds; the exact-erf
// form above stays in the tiled template and therefore in the TT_F32 reference-precision mode).  No |x|, no sign transfer, no second
// polynomial:  x^2, four FMAs, one product, exp2 (log2 e folded into the coefficients), 1 + e, rcp, one product = 8 VALU + 2
// transcendentals per element against 16 + 2 -- the epilogue is VALU-bound on it (64 evaluations per lane and 256 x 256 tile).
// Saturates cleanly: x -> +inf gives exp2(-inf) = 0 -> x; x -> -inf gives 1 / inf = 0 -> -0; the leading coefficient is positive, so
// x P(x^2) is monotone and no inf - inf can form.
__device__ __forceinline__ f32pk_t gelu_sig_pk(f32pk_t x) {
  const f32pk_t x2 = x * x;
  f32pk_t q = x2 * -4.111726866540266e-06f + 1.0587536235107109e-04f;
  q = q * x2 + 2.534117375034839e-04f;
  q = q * x2 + -0.10500594228506088f;
  q = q * x2 + -2.3021652698516846f;                        // -log2(e) P(x^2)
  const f32pk_t a = x * q;
  const f32pk_t d = (f32pk_t){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + 1.0f;
  return x * (f32pk_t){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

// GELU without transcendentals (round 6; bf16 storage in the persistent GEGLU kernel only): Phi(x) ~ 0.5 + xc P(xc^2), xc = clamp(x, +-3.75),
// P of degree 6 in x^2 (least-squares / Lawson fit on Chebyshev nodes: max |dPhi| = 5.6e-5, max |x dPhi| = 3.9e-4 over |x| <= 12 in fp32
// arithmetic -- an order below bf16's 2^-9 output rounding wherever the output exceeds 0.1, absolute 4e-4 at most elsewhere; beyond the clamp
// Phi stays at 0.99997 / 3.2e-5, a relative 3e-5 on gelu(x) ~ x).  v_exp_f32 and v_rcp_f32 issue at a quarter of the VALU rate: the sigmoid
// form above costs 4 packed-equivalent VALU + 2 transcendentals per value = 48 clocks, this one 9 packed instructions + 2 clamps per PAIR = 22.
__device__ __forceinline__ f32pk_t gelu_poly_pk(f32pk_t x) {
  const f32pk_t xc = {__builtin_amdgcn_fmed3f(x.x, -3.75f, 3.75f), __builtin_amdgcn_fmed3f(x.y, -3.75f, 3.75f)};
  const f32pk_t t = xc * xc;
  f32pk_t q = t * 3.9124383732769275e-08f + -2.3762543150951387e-06f;
  q = q * t + 6.234781903913245e-05f;
  q = q * t + -0.0009441798320040107f;
  q = q * t + 0.009362553246319294f;
  q = q * t + -0.06578987091779709f;
  q = q * t + 0.39870646595954895f;
  return x * (xc * q + 0.5f);
}

// ---------------------------------------------------------------- LDS tile staging (global -> LDS DMA)
// A tile is ROWS x (CPR chunks of 16 B).  The LDS image is lane-linear (DMA requirement): slot s holds
// the chunk (row = s / CPR, chunk' = s % CPR); the data stored there is source chunk  c = c' ^ swz(row),
// so a reader of (row, c) looks at chunk' = c ^ swz(row).  swz spreads the 16 rows a ds_read_b128 lane
// group touches over all 16 slots of the 256-B bank row (cdna guide section 6, guideline 4).
template <int CPR> __device__ __forceinline__ int tile_swz(int row) {
  if constexpr (CPR == 8) return (row >> 1) & 7;        // 128-B rows: two rows per bank row
  else if constexpr (CPR == 16 || CPR == 32) return row & 15;   // 256-B / 512-B rows (the XOR stays inside a 256-B half)
  else return (row >> 2) & 3;                           // 64-B rows
}
template <int CPR> __device__ __forceinline__ int tile_off(int row, int chunk) {
  return (row * CPR + (chunk ^ tile_swz<CPR>(row))) << 4;
}
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ uint4 lds_read16(const char* smem, int off) { return *(const uint4*)(smem + off); }

// ---- LDS access the compiler cannot see.  hipcc guards every LDS access that may alias an in-flight LDS-DMA with
// `s_waitcnt vmcnt(0)`, which drains the prefetch ring before each tile and makes counted vmcnt(N) pipelines pointless.
// These wrappers issue the ds instruction as opaque asm; the CALLER orders them: lds_wait<N>() (s_waitcnt lgkmcnt(N) +
// sched_barrier, cdna guide rule 18) before the first use of a raw read, and the usual vmcnt/barrier protocol against
// the DMA.  LDS instructions of one wave complete in order, so lds_wait<N> retires all but the N most recent.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long)((__attribute__((address_space(3))) const char*)p);
}
typedef unsigned raw_u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned raw_u32x2_t __attribute__((ext_vector_type(2)));
typedef float raw_f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ raw_u32x4_t lds_read16_raw(unsigned addr) {
  raw_u32x4_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// same with a compile-time byte offset in the instruction's 16-bit offset field (no address arithmetic per read)
template <int OFF> __device__ __forceinline__ raw_u32x4_t lds_read16_raw_off(unsigned addr) {
  static_assert(OFF >= 0, "offset");
  raw_u32x4_t v;
  if constexpr (OFF < 65536) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  else asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr + (unsigned)OFF));     // beyond the 16-bit field (fp32, D = 128)
  return v;
}
// gfx950's transposing read (ds_read_b64_tr_b16): a 16-lane group reads a [4 rows][16 columns] block of 16-bit elements, lane i supplying the
// address of the 8-byte run (row i >> 2, columns 4 (i & 3) ..), and lane i receives column i, rows 0..3 (tools/tr_read_probe.hip)
template <int OFF> __device__ __forceinline__ raw_u32x2_t lds_read8_tr_off(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "offset");
  raw_u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ raw_u32x2_t lds_read8_raw(unsigned addr) {
  raw_u32x2_t v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ void lds_write16_raw(unsigned addr, float a, float b, float c, float d) {
  const raw_f32x4_t v = {a, b, c, d};
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------- kernel-argument warm-up
// hipcc loads the kernel arguments where they are first needed: a kernel with a few hundred bytes of them meets 3-5 SERIALISED
// scalar-cache misses (s_load ... s_waitcnt lgkmcnt(0), ~0.3-0.5 us each when the line comes from L2 / the fabric) before its first
// global load -- measured with the TT_GEMM_TIMELINE stamps: 2.0 us of "prologue" for ~140 scalar instructions.
// kernarg_touch<BYTES>() reads one dword of every 64-byte line of the argument segment in ONE batch at kernel entry; the first wait
// then covers all of them and every later s_load hits the scalar cache.
template <int BYTES> __device__ __forceinline__ void kernarg_touch() {
  typedef const __attribute__((address_space(4))) unsigned* kptr_t;
  kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned acc = 0;
#pragma unroll
  for (int off = 0; off < BYTES; off += 64) acc |= ka[off / 4];
  asm volatile("" ::"s"(acc));
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- split-fp16 operands (TT_F32 "split16", tt_gemm_set_f32_split)
// x = 2^8 h + 2^-3 l with h = fp16(x 2^-8), l = fp16((x - 2^8 h) 2^3) = fp16(8 x - 2048 h): 4 fp32 values -> two fp16 pairs each (10 VALU;
// the fp16 operand of v_fma_mix is widened by the instruction, the fused result is rounded once).  See gemm_kernel.h (`mma`) for the scales.
int tt_internal_f32_split();         // gemm.hip: the process-wide switch (attention.hip follows it)
__device__ __forceinline__ void split_f16x4(unsigned x0, unsigned x1, unsigned x2, unsigned x3, uint2& h, uint2& l) {
  const f32x2_t v01 = (f32x2_t){__uint_as_float(x0), __uint_as_float(x1)}, v23 = (f32x2_t){__uint_as_float(x2), __uint_as_float(x3)};
  const f16x2_t h01 = __builtin_convertvector(v01 * 0.00390625f, f16x2_t), h23 = __builtin_convertvector(v23 * 0.00390625f, f16x2_t);
  h = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
  const f32x2_t e01 = v01 * 8.0f, e23 = v23 * 8.0f;
  const float m2048 = -2048.0f;
  unsigned l01, l23;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h.x), "v"(m2048), "v"(e01.x));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h.x), "v"(m2048), "v"(e01.y));
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h.y), "v"(m2048), "v"(e23.x));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h.y), "v"(m2048), "v"(e23.y));
  l = make_uint2(l01, l23);
}
