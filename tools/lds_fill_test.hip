// LDS fill-rate probe (MI355X): how fast can a CU bring L2-resident data into LDS, by LDS-DMA (buffer_load ... lds, what the GEMM
// K loop does) and by global_load_dwordx4 -> VGPR -> ds_write_b128?  No MFMAs, no reads of the LDS data: this is the ceiling
// of the staging path alone (the GEMM K loop is bound by it).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_fill_test.hip -o tools/lds_fill_test.bin ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}

// Each block of NT threads streams `iters` tiles of TILE bytes (its own slice of a buffer that fits L2) into a NSTAGE-deep LDS
// ring, waiting with counted vmcnt like the GEMM does.  MODE 0: LDS-DMA.  MODE 1: load to VGPRs, ds_write one stage later.
template <int NT, int TILE, int NSTAGE, int MODE>
__global__ __launch_bounds__(NT) void fill_kernel(const char* src, unsigned bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PIECES = TILE / (NT * 16);              // 16-byte chunks per thread per tile
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t r = rsrc(src, bytes);
  // every block walks the whole buffer (so all blocks of an XCD share the lines in L2), offset by its id
  unsigned pos = (unsigned)(((unsigned long)blockIdx.x * 7919u * TILE) % bytes);
  unsigned acc = 0;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      char* dst = smem + (it % NSTAGE) * TILE + wid * 1024;
#pragma unroll
      for (int i = 0; i < PIECES; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + i * (NT * 16)), 16,
                                                 (int)(tid * 16 + i * NT * 16), (int)pos, 0, 0);
      pos += TILE; if (pos + TILE > bytes) pos = 0;
      // keep NSTAGE-1 tiles in flight
      if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES));
      else if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PIECES));
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * PIECES));
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)");
  } else {
    u4 regs[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) regs[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(tid * 16 + i * NT * 16), (int)pos, 0);
    for (int it = 0; it < iters; ++it) {
      pos += TILE; if (pos + TILE > bytes) pos = 0;
      u4 next[PIECES];
#pragma unroll
      for (int i = 0; i < PIECES; ++i) next[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(tid * 16 + i * NT * 16), (int)pos, 0);
      char* dst = smem + (it % NSTAGE) * TILE;
#pragma unroll
      for (int i = 0; i < PIECES; ++i) *(u4*)(dst + tid * 16 + i * NT * 16) = regs[i];
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < PIECES; ++i) regs[i] = next[i];
    }
  }
  // make the LDS contents observable so nothing is dropped
  __syncthreads();
  acc = *(unsigned*)(smem + (tid * 4) % TILE);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NT, int TILE, int NSTAGE, int MODE>
void run(const char* name, const char* src, unsigned bytes, int blocks_per_cu, unsigned* sink) {
  const int iters = 4000;
  const size_t lds = (size_t)NSTAGE * TILE;
  hipFuncSetAttribute((const void*)fill_kernel<NT, TILE, NSTAGE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<NT, TILE, NSTAGE, MODE>), dim3(blocks), dim3(NT), lds, 0, src, bytes, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tb = (double)blocks * iters * TILE / (ms * 1e-3) / 1e12;
  printf("%-44s %d block(s)/CU  %7.3f ms  %6.2f TB/s  = %5.1f B/clk/CU at 2.4 GHz\n", name, blocks_per_cu, ms, tb, tb * 1e12 / 256 / 2.4e9);
  if (hipGetLastError() != hipSuccess) printf("  launch error\n");
}

int main(int argc, char** argv) {
  // buffer size in MiB: 2 fits every XCD's 4 MiB L2 (all blocks walk the whole buffer), 16 / 64 are served by the Infinity Cache
  const unsigned mib = argc > 1 ? (unsigned)atoi(argv[1]) : 2u;
  const unsigned bytes = mib << 20;
  printf("source buffer %u MiB\n", mib);
  char* src; unsigned* sink;
  hipMalloc(&src, bytes); hipMalloc(&sink, 64); hipMemset(src, 1, bytes);
  run<256, 32768, 2, 0>("LDS-DMA   32 KiB tile, 2 stages, 4 waves", src, bytes, 2, sink);
  run<512, 32768, 2, 0>("LDS-DMA   32 KiB tile, 2 stages, 8 waves", src, bytes, 2, sink);
  run<512, 24576, 3, 0>("LDS-DMA   24 KiB tile, 3 stages, 8 waves", src, bytes, 2, sink);
  run<512, 16384, 4, 0>("LDS-DMA   16 KiB tile, 4 stages, 8 waves", src, bytes, 2, sink);
  run<512, 65536, 2, 0>("LDS-DMA   64 KiB tile, 2 stages, 8 waves", src, bytes, 1, sink);
  run<256, 32768, 2, 1>("load+ds_write 32 KiB tile, 2 stages, 4 waves", src, bytes, 2, sink);
  run<512, 65536, 2, 1>("load+ds_write 64 KiB tile, 2 stages, 8 waves", src, bytes, 1, sink);
  return 0;
}
