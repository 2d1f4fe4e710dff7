// Does an XCD's L2 keep what its CUs WROTE in the previous kernel?  (MI355X: 8 XCDs, 4 MiB L2 each, not coherent with each other.)
// Kernel A: block b writes region b.  Kernel B: block b reads region (b + shift) -- shift 0: the XCD that wrote the lines reads
// them; shift 1: the neighbouring XCD does (blocks are placed round-robin on XCDs).  If shift 0 is faster, a consistent
// rows -> XCD mapping across the kernels of a layer chain would turn producer -> consumer traffic into L2 hits.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_retention_test.hip -o tools/l2_retention_test.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void writer(u4* buf, int vec_per_block) {
  u4* p = buf + (size_t)blockIdx.x * vec_per_block;
  for (int i = threadIdx.x; i < vec_per_block; i += 256) p[i] = (u4){(unsigned)i, blockIdx.x, 3u, 4u};
}
__global__ __launch_bounds__(256) void reader(const u4* buf, int vec_per_block, int shift, int nblocks, unsigned* sink) {
  const u4* p = buf + (size_t)((blockIdx.x + shift) % nblocks) * vec_per_block;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < vec_per_block; i += 256) { const u4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv) {
  const int mib = argc > 1 ? atoi(argv[1]) : 16;            // total buffer; 16 MiB = 2 MiB per XCD
  const int nblocks = 2048;
  const size_t bytes = (size_t)mib << 20;
  const int vec_per_block = (int)(bytes / 16 / nblocks);
  u4* buf; unsigned* sink;
  (void)hipMalloc(&buf, bytes); (void)hipMalloc(&sink, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("buffer %d MiB, %d blocks x %d KiB\n", mib, nblocks, vec_per_block * 16 / 1024);
  for (int shift = 0; shift < 3; ++shift) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(writer, dim3(nblocks), dim3(256), 0, 0, buf, vec_per_block);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(reader, dim3(nblocks), dim3(256), 0, 0, buf, vec_per_block, shift, nblocks, sink);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("  reader after writer, shift %d: %.1f us  %.2f TB/s\n", shift, best * 1e3, bytes / (best * 1e-3) / 1e12);
  }
  // reference: reader twice in a row (its own previous read left the lines in L2)
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(reader, dim3(nblocks), dim3(256), 0, 0, buf, vec_per_block, 0, nblocks, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(reader, dim3(nblocks), dim3(256), 0, 0, buf, vec_per_block, 0, nblocks, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("  reader after reader (same mapping): %.1f us  %.2f TB/s\n", best * 1e3, bytes / (best * 1e-3) / 1e12);
  return 0;
}
