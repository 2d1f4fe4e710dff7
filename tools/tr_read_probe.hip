// ds_read_b64_tr_b16 on gfx950: which element does lane l get?  One wave, an LDS image of 64 rows (keys) x 64 columns (d) of u16 = row * 64 + col,
// 128-byte rows.  Hypothesis (the attention V recipe): a 16-lane group reads a [4 rows][16 columns] block, lane i of the group supplying the address
// of the 8-byte run (row i >> 2, columns (i & 3) * 4 ..), and receives column i of the block, rows 0..3.
//   hipcc --offload-arch=gfx950 -O3 tools/tr_read_probe.hip -o tools/tr_read_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[64 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 64; i += 64) img[i] = (unsigned short)i;
  __syncthreads();
  const int g = lane >> 4, i = lane & 15, hi = g >> 1;
  const int key0 = hi * 16, d0 = (g & 1) * 16;
  const unsigned addr = (unsigned)(size_t)img + (unsigned)(((key0 + (i >> 2)) * 64 + d0 + (i & 3) * 4) * 2);
  u2v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = (unsigned short)(v.x & 0xffff); out[lane * 4 + 1] = (unsigned short)(v.x >> 16);
  out[lane * 4 + 2] = (unsigned short)(v.y & 0xffff); out[lane * 4 + 3] = (unsigned short)(v.y >> 16);
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, i = l & 15, hi = g >> 1, l31 = (g & 1) * 16 + i;
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      const int want = (hi * 16 + j) * 64 + l31;
      printf(" (row %2d col %2d)%s", h[l * 4 + j] / 64, h[l * 4 + j] % 64, h[l * 4 + j] == want ? "" : "!");
      bad += h[l * 4 + j] != want;
    }
    printf("\n");
  }
  printf("%s: %d mismatches against  lane (group g, i) elem j = image[16 (g >> 1) + j][16 (g & 1) + i]\n", bad ? "DIFFERENT" : "AS EXPECTED", bad);
  return 0;
}
