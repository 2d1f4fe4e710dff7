// LDS-DMA rate vs the SHAPE of a 1 KiB piece (MI355X): a GEMM K slab is staged as rows of ROWB bytes at a row stride of
// LD bytes, so one wave-instruction (64 lanes x 16 B) touches 1024/ROWB different cache lines.  No MFMAs, no LDS reads.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_shape_test.hip -o tools/dma_shape_test.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
// block = 512 threads; slab = ROWS rows x ROWB bytes (ROWS*ROWB = 32 KiB); ring of NST slabs; K loop of KT slabs along a row,
// then the next ROWS rows.  The matrix has `rows` rows of LD bytes; blocks of one XCD share row panels (block b starts at panel b/8 % ..).
template <int ROWB, int NST>
__global__ __launch_bounds__(512) void k(const char* src, unsigned bytes, int rows, int ld, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SLAB = 32768, ROWS = SLAB / ROWB, LPR = ROWB / 16, PIECES = SLAB / (512 * 16);   // 4 pieces per thread
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t r = rsrc(src, bytes);
  const int KT = ld / ROWB, panels = rows / ROWS;
  int voff[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) { const int c = i * 512 + tid, row = c / LPR, ch = c % LPR; voff[i] = row * ld + ch * 16; }
  int panel = (blockIdx.x >> 3) % panels, kt = 0;
  for (int it = 0; it < iters; ++it) {
    char* dst = smem + (it % NST) * SLAB + wid * 1024;
    const int soff = __builtin_amdgcn_readfirstlane(panel * ROWS * ld + kt * ROWB);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int v = voff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, v, soff, 0, 0);
    }
    if (++kt == KT) { kt = 0; panel = (panel + 32) % panels; }
    if (NST == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES));
    else if (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES));
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES));
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)");
  __syncthreads();
  const unsigned a = *(unsigned*)(smem + (tid * 4) % SLAB);
  if (a == 0x12345678u) sink[0] = a;
}
template <int ROWB, int NST>
void run(const char* src, unsigned bytes, int rows, int ld, unsigned* sink) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)k<ROWB, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * 32768);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ROWB, NST>), dim3(256), dim3(512), NST * 32768, 0, src, bytes, rows, ld, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tb = 256.0 * iters * 32768 / (ms * 1e-3) / 1e12;
  printf("  rows of %4d B (ld %5d B, %5d rows = %4.0f MiB), ring %d: %6.2f TB/s = %5.1f B/clk/CU at 2.4 GHz, %.2f us per 32 KiB slab\n",
         ROWB, ld, rows, (double)rows * ld / 1048576.0, NST, tb, tb * 1e12 / 256 / 2.4e9, ms * 1e3 / iters);
}
int main() {
  char* src; unsigned* sink;
  const unsigned cap = 1u << 30;
  hipMalloc(&src, cap); hipMalloc(&sink, 64); hipMemset(src, 1, cap);
  struct { int rows, ld; } cfgs[] = {{4096, 640}, {50176, 640}, {8192, 16384}, {12544, 1280}, {50176, 2560}};
  for (auto& c : cfgs) {
    const unsigned bytes = (unsigned)((long)c.rows * c.ld);
    printf("matrix %d rows x %d B:\n", c.rows, c.ld);
    run<64, 4>(src, bytes, c.rows, c.ld, sink);
    run<128, 4>(src, bytes, c.rows, c.ld, sink);
    if (c.ld % 256 == 0) run<256, 4>(src, bytes, c.rows, c.ld, sink);
    run<64, 2>(src, bytes, c.rows, c.ld, sink);
    run<128, 2>(src, bytes, c.rows, c.ld, sink);
  }
  return 0;
}
