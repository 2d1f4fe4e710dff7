// Probe for the split-fp16 route to fp32-class products ("split16"):  x = hi + lo with hi = fp16(x), lo = fp16(x - hi),
//   a * b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi        (three v_mfma_f32_32x32x16_f16 into one fp32 accumulator)
// Questions: (1) does the f16 MFMA keep DENORMAL inputs (lo of |x| < 0.125 is below 2^-14) or flush them; (2) what relative error
// does the three-term product reach against an fp64 dot product, and against the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_split_probe.hip -o tools/mfma_split_probe.bin && tools/mfma_split_probe.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one wave: C[32][32] = A[32][K] B[32][K]^T, K = 16 * steps; lane (l31, hi) owns k = 16 s + 8 hi .. + 7 of row l31
__global__ void k_split(const float* a, const float* b, float* c_split, float* c_hi, float* c_f32, int K) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  f32x16 acc = {0}, acch = {0}, accf = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    f16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const float x = a[l31 * K + k0 + 8 * hi + j], y = b[l31 * K + k0 + 8 * hi + j];
      ah[j] = (_Float16)x; al[j] = (_Float16)(x - (float)ah[j]);
      bh[j] = (_Float16)y; bl[j] = (_Float16)(y - (float)bh[j]);
    }
    acch = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acch, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);       // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    for (int j = 0; j < 8; ++j) {       // exact-fp32 MFMA, K = 2 per instruction: lane half hi supplies k-slot hi
      accf = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l31 * K + k0 + 2 * j + hi], b[l31 * K + k0 + 2 * j + hi], accf, 0, 0, 0);
    }
  }
  // D layout of 32x32 MFMA: lane (l31, hi) register r = 4 g + j holds D[row 8 g + 4 hi + j][col l31]  (A rows = rows, B rows = cols)
  for (int r = 0; r < 16; ++r) {
    const int row = 8 * (r >> 2) + 4 * hi + (r & 3);
    c_split[row * 32 + l31] = acc[r]; c_hi[row * 32 + l31] = acch[r]; c_f32[row * 32 + l31] = accf[r];
  }
}

// denormal test: A = 2^-20 (an fp16 denormal: below 2^-14), B = 1024: the product 2^-10 survives only if the input is not flushed
__global__ void k_denorm(float* out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.f; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}

int main() {
  float* dd; hipMalloc(&dd, 4);
  hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, dd);
  float hd; hipMemcpy(&hd, dd, 4, hipMemcpyDeviceToHost);
  printf("denormal fp16 inputs (2^-20 x 1024, 16 terms): got %.9g, kept = %.9g, flushed = 0  -> %s\n", hd, 16 * 9.5367431640625e-07 * 1024,
         hd != 0.f ? "KEPT" : "FLUSHED");
  const int K = 1280;
  float *ha = (float*)malloc(32 * K * 4), *hb = (float*)malloc(32 * K * 4);
  for (int scale_case = 0; scale_case < 3; ++scale_case) {
    const float sa = scale_case == 0 ? 1.f : (scale_case == 1 ? 100.f : 1e-3f), sb = scale_case == 0 ? 0.03f : (scale_case == 1 ? 0.03f : 0.03f);
    srand(1 + scale_case);
    for (int i = 0; i < 32 * K; ++i) {
      ha[i] = sa * ((rand() / (float)RAND_MAX) * 2.f - 1.f) + (scale_case == 1 ? 50.f : 0.f);     // case 1: large row mean (pre-LayerNorm rows)
      hb[i] = sb * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
    }
    float *da, *db, *dc[3];
    hipMalloc(&da, 32 * K * 4); hipMalloc(&db, 32 * K * 4);
    for (int i = 0; i < 3; ++i) hipMalloc(&dc[i], 32 * 32 * 4);
    hipMemcpy(da, ha, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, 32 * K * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, da, db, dc[0], dc[1], dc[2], K);
    float hc[3][1024];
    for (int i = 0; i < 3; ++i) hipMemcpy(hc[i], dc[i], 4096, hipMemcpyDeviceToHost);
    double err[3] = {0, 0, 0}, ref2 = 0, mx[3] = {0, 0, 0};
    for (int r = 0; r < 32; ++r)
      for (int c = 0; c < 32; ++c) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)ha[r * K + k] * (double)hb[c * K + k];
        ref2 += s * s;
        for (int i = 0; i < 3; ++i) { const double d = hc[i][r * 32 + c] - s; err[i] += d * d; if (fabs(d) > mx[i]) mx[i] = fabs(d); }
      }
    printf("case %d (|a| ~ %g%s, |b| ~ %g, K = %d): rel-L2 vs fp64  split 3-term %.3e (max abs %.3e) | hi-only %.3e | exact-fp32 MFMA %.3e (max abs %.3e)\n",
           scale_case, sa, scale_case == 1 ? " + 50" : "", sb, K, sqrt(err[0] / ref2), mx[0], sqrt(err[1] / ref2), sqrt(err[2] / ref2), mx[2]);
  }
  return 0;
}
