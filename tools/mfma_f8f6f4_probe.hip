// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (gfx950): operand layout check against a CPU product and issue rate
// against the non-scaled 32x32x16 fp8 form.   hipcc --offload-arch=gfx950 -O3 tools/mfma_f8f6f4_probe.hip -o tools/mfma_f8f6f4_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// assumed layout: A is 32 x 64 (row i, k), lane l supplies row l & 31, k = 32 * (l >> 5) + byte index; B likewise (column l & 31);
// C/D as every 32x32 form: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__global__ void one(const unsigned char* a, const unsigned char* b, float* c, int scale) {
  const int l = threadIdx.x;
  v8i fa, fb;
  memcpy(&fa, a + (l & 31) * 64 + 32 * (l >> 5), 32);
  memcpy(&fb, b + (l & 31) * 64 + 32 * (l >> 5), 32);
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc, 0, 0, 0, scale, 0, scale);
  for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
template <int WIDE>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  v8i fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); fb[i] = 0x30303030 ^ (threadIdx.x << (i & 3)); }
  v16f acc[4];
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (WIDE) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[k], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else {
        const long a8 = ((long)fa[1] << 32) | (unsigned)fa[0], b8 = ((long)fb[1] << 32) | (unsigned)fb[0];
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc[k], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static float e4m3_to_f(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}
int main() {
  unsigned char ha[32 * 64], hb[32 * 64];
  srand(7);
  for (int i = 0; i < 32 * 64; ++i) { ha[i] = (unsigned char)((rand() % 2 ? 0x80 : 0) | (0x28 + rand() % 24)); hb[i] = (unsigned char)((rand() % 2 ? 0x80 : 0) | (0x28 + rand() % 24)); }
  unsigned char *da, *db; float* dc;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, 32 * 32 * 4);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  for (int scale : {0x7f7f7f7f, 0}) {
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, da, db, dc, scale);
    float hc[32 * 32];
    hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    double worst = 0, ratio = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0;
      for (int k = 0; k < 64; ++k) ref += (double)e4m3_to_f(ha[i * 64 + k]) * e4m3_to_f(hb[j * 64 + k]);
      worst = fmax(worst, fabs(hc[i * 32 + j] - ref));
      if (i == 3 && j == 5) ratio = hc[i * 32 + j] / ref;
    }
    printf("scale bytes 0x%02x: max |C - A B^T| = %.3g   (C[3][5] / ref = %.6g)\n", scale & 255, worst, ratio);
  }
  float* dout; hipMalloc(&dout, 1024 * 256 * 4);
  for (int wide = 0; wide < 2; ++wide) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (wide) hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, dout, iters); else hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * (wide ? 64 : 16) * 4.0 * iters * 1024 * 4;
    printf("%s: %.1f TFLOP/s\n", wide ? "v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)" : "v_mfma_f32_32x32x16_fp8_fp8", flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
