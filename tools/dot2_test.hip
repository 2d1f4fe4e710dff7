// standalone check of v_dot2c_f32_{bf16,f16} as used by ln_stat (gemm_kernel.h): hipcc --offload-arch=gfx950 -O3 tools/dot2_test.hip -o gpurun_out/dot2_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 b2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const u4* xb, const u4* xh, float* o) {
  const int t = threadIdx.x;
  u4 f = xb[t];
  float s = 0.f, q = 0.f;
#ifdef TT_WORKAROUND
  unsigned w0[4] = {f.x, f.y, f.z, f.w};
#define F(d) w0[d]
#else
#define F(d) f[d]      // miscompiled by hipcc 7.2: every iteration reads f[0]
#endif
  const b2 one = __builtin_bit_cast(b2, 0x3F803F80u);
  for (int d = 0; d < 4; ++d) { b2 x = __builtin_bit_cast(b2, F(d)); s = __builtin_amdgcn_fdot2_f32_bf16(x, one, s, false); q = __builtin_amdgcn_fdot2_f32_bf16(x, x, q, false); }
  o[t * 4] = s; o[t * 4 + 1] = q;
  f = xh[t]; s = 0.f; q = 0.f;
#ifdef TT_WORKAROUND
  w0[0] = f.x; w0[1] = f.y; w0[2] = f.z; w0[3] = f.w;
#endif
  const h2 oneh = __builtin_bit_cast(h2, 0x3C003C00u);
  for (int d = 0; d < 4; ++d) { h2 x = __builtin_bit_cast(h2, F(d)); s = __builtin_amdgcn_fdot2(x, oneh, s, false); q = __builtin_amdgcn_fdot2(x, x, q, false); }
  o[t * 4 + 2] = s; o[t * 4 + 3] = q;
}
int main() {
  const int n = 64;
  unsigned short hb[n * 8], hh[n * 8]; float ref[n * 4] = {0};
  for (int i = 0; i < n * 8; ++i) {
    float v = (float)((i * 37) % 23 - 11) * 0.25f;             // exactly representable in bf16 and f16
    unsigned u; memcpy(&u, &v, 4); hb[i] = u >> 16;
    _Float16 h = (_Float16)v; memcpy(&hh[i], &h, 2);
    ref[(i / 8) * 4] += v; ref[(i / 8) * 4 + 1] += v * v; ref[(i / 8) * 4 + 2] += v; ref[(i / 8) * 4 + 3] += v * v;
  }
  void *db, *dh; float* dout; float out[n * 4];
  hipMalloc(&db, sizeof hb); hipMalloc(&dh, sizeof hh); hipMalloc(&dout, sizeof out);
  hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice); hipMemcpy(dh, hh, sizeof hh, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, (const u4*)db, (const u4*)dh, dout);
  hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < n * 4; ++i) e = fmax(e, fabs(out[i] - ref[i]));
  printf("max |err| = %g ; lane0: bf16 s=%g q=%g (ref %g %g)  f16 s=%g q=%g (ref %g %g)\n", e, out[0], out[1], ref[0], ref[1], out[2], out[3], ref[2], ref[3]);
  return 0;
}
