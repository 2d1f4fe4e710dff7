// Probe for the round-3 GEMM main loop (not part of libttvdm): persistent workgroups, 256-wide tiles, a 4-slot ring of
// 32-deep K slabs filled by LDS-DMA that keeps streaming across tile boundaries, and two groups of 4 waves that take turns
// on the matrix pipe (one group issues its MFMAs while the other reads fragments / issues DMA; they swap at every barrier).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ithis_and_that_vdm_amd/csrc tools/gemm_pp_probe.hip -o tools/gemm_pp_probe.bin
//   tools/gemm_pp_probe.bin            (prints TFLOP/s per shape and variant; checks sampled outputs against a naive kernel)
#include <vector>
#include <string>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <type_traits>
#include "common.h"

void tt_set_error(const char*, ...) {}

struct PP {
  const char* a; const char* w; char* out; const float* bias; const char* res;
  int m, n, k; long lda, ldw, ldo, ldr;
  unsigned a_bytes, w_bytes, out_bytes, bias_bytes, res_bytes;
  int tiles_m, tiles_n, ntiles, group_m;
  unsigned long long* tl;
};

constexpr int kInv = (int)0x80000000;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, (int)bytes, 0x00020000);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PPMODE 0: all 8 waves in lock step (read -> barrier -> MFMA -> barrier); 1: the two groups of 4 waves run one barrier apart
template <int BM, int BN, int WGM, int WGN, int PPMODE, int HAS_RES, int VAR, bool TIMING, int NSLOT = 4>
__global__ __launch_bounds__(512, 2) void gemm_pp(const PP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = 512, CPR = 4;
  constexpr int AR = BM * CPR / NT, BR = BN * CPR / NT;
  static_assert(BM * CPR % NT == 0 && BN * CPR % NT == 0 && WGM * WGN == 8, "tile");
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, SLOT = A_BYTES + B_BYTES;
  constexpr int D = NSLOT - 1;                                // prefetch distance in slabs
  constexpr int GL = AR + BR;
  constexpr int STRIP_OFF = NSLOT == 5 ? 0 : NSLOT * SLOT;    // 8 x 4 KiB strips behind the ring (5 slots: TIMING ONLY, strips alias slot 0)
  constexpr int WTM = BM / WGM, WTN = BN / WGN, FM = WTM / 32, FN = WTN / 32;
  static_assert(WTN == 64, "bias DMA: one dword instruction per wave covers 64 columns");
  constexpr int EST = FM * FN * 4;                            // epilogue stores per lane

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;                                   // waves 0-3 / 4-7: one wave of each group per SIMD
  const int wr = wid / WGN, wc = wid - wr * WGN;
  const int l31 = lane & 31, hi = lane >> 5;

  const __amdgpu_buffer_rsrc_t ra = mk_rsrc(p.a, p.a_bytes), rw = mk_rsrc(p.w, p.w_bytes);
  const __amdgpu_buffer_rsrc_t rb = mk_rsrc(p.bias, p.bias_bytes), ro = mk_rsrc(p.out, p.out_bytes);
  const __amdgpu_buffer_rsrc_t rres = mk_rsrc(p.res, HAS_RES ? p.res_bytes : 0);

  const int KS = p.k >> 5;                                    // slabs per tile (K % 32 == 0, K >= 128)
  const int nwg = gridDim.x;
  const int my_tiles = (int)blockIdx.x < p.ntiles ? (p.ntiles - 1 - (int)blockIdx.x) / nwg + 1 : 0;
  const int S = my_tiles * KS;

  // virtual block id -> tile (XCD-contiguous, grouped order as in gemm_kernel.h)
  auto tile_of = [&](int it, int& m0, int& n0) {
    int bid = it * nwg + (int)blockIdx.x;
    const int q = p.ntiles >> 3, r = p.ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gm = p.group_m, per_group = gm * p.tiles_n;
    const int g = bid / per_group, first = g * gm, rows = min(gm, p.tiles_m - first), rem = bid - g * per_group;
    const int tn = rem / rows, tm = first + (rem - tn * rows);
    m0 = tm * BM; n0 = tn * BN;
  };

  // ---- producer (DMA) cursor: slab `ps` of this block's slab sequence; per-lane row offsets of the tile it belongs to
  const int crow = tid >> 2, cchunk = tid & 3;                // 128 rows x 4 chunks per staging pass
  int pva[AR], pvb[BR];
  int p_it = 0, p_ks = 0, p_n0 = 0;
  auto producer_tile = [&](int it) {
    int m0 = 0, n0 = 0;
    const bool ok = it < my_tiles;
    if (ok) tile_of(it, m0, n0);
    p_n0 = n0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int r = i * 128 + crow, gmr = m0 + r;
      const int ch = cchunk ^ tile_swz<CPR>(r);
      pva[i] = (ok && gmr < p.m) ? (int)(((long)gmr * p.lda + ch * 8) * 2) : kInv;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int r = i * 128 + crow, gn = n0 + r;
      const int ch = cchunk ^ tile_swz<CPR>(r);
      pvb[i] = (ok && gn < p.n) ? (int)(((long)gn * p.ldw + ch * 8) * 2) : kInv;
    }
  };
  producer_tile(0);
  // half = 0: the A part of slab (p_it, p_ks) into `slot`; half = 1: the W part (+ the bias of the tile with slab 3), then advance
  auto stage = [&](int slot, int half) {
    char* base = smem + slot * SLOT + wid * 1024;
    const int soff = __builtin_amdgcn_readfirstlane(p_ks) * 64;
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int v = pva[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + i * 8192), 16, v, soff, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int v = pvb[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + A_BYTES + i * 8192), 16, v, soff, 0, 0);
      }
      if (VAR == 2 || p_ks == 3) {                           // this wave's 64 bias values into its own strip (VAR 2: a dropped load on the other slabs, uniform count)
        const int gn = p_n0 + wc * 64 + lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(smem + STRIP_OFF + wid * 4096), 4,
                                                 (p_ks == 3 && p_it < my_tiles && gn < p.n) ? gn * 4 : kInv, 0, 0, 0);
      }
      if (++p_ks == KS) { p_ks = 0; ++p_it; producer_tile(p_it); }
    }
  };

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds_base = lds_addr(smem);
  int a_off[FM], b_off[FN];                                   // byte offset of (row, chunk 0^swz) ; chunk c adds ((c ^ swz) << 4)
  int a_swz[FM], b_swz[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) { const int r = wr * WTM + i * 32 + l31; a_off[i] = r * 64; a_swz[i] = tile_swz<CPR>(r); }
#pragma unroll
  for (int j = 0; j < FN; ++j) { const int r = wc * WTN + j * 32 + l31; b_off[j] = A_BYTES + r * 64; b_swz[j] = tile_swz<CPR>(r); }
  raw_u32x4_t af[FM], bf[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) af[i] = (raw_u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)lane};
#pragma unroll
  for (int j = 0; j < FN; ++j) bf[j] = (raw_u32x4_t){0x3f803f80u, 0x3f803f80u, (unsigned)lane, 0x3f803f80u};
  auto read_frags = [&](unsigned sa, int ks, raw_u32x4_t (&fa)[FM], raw_u32x4_t (&fb)[FN]) {
    const int chunk = ks * 2 + hi;
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[i] = lds_read16_raw(sa + a_off[i] + ((chunk ^ a_swz[i]) << 4));
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[j] = lds_read16_raw(sa + b_off[j] + ((chunk ^ b_swz[j]) << 4));
  };
  auto mma1 = [&](const raw_u32x4_t (&fa)[FM], const raw_u32x4_t (&fb)[FN], int i, int j) {
    acc[i][j] = Cvt<bf16_tag>::mfma32(make_uint4(fb[j].x, fb[j].y, fb[j].z, fb[j].w), make_uint4(fa[i].x, fa[i].y, fa[i].z, fa[i].w), acc[i][j]);
  };
  constexpr bool PRIO = VAR < 20;
  auto mma = [&](const raw_u32x4_t (&fa)[FM], const raw_u32x4_t (&fb)[FN]) {
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) mma1(fa, fb, i, j);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  // the first fragment row, then (caller) the next phase's LDS reads, then the rest
  auto mma_head = [&](const raw_u32x4_t (&fa)[FM], const raw_u32x4_t (&fb)[FN]) {
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < FN; ++j) mma1(fa, fb, 0, j);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma_tail = [&](const raw_u32x4_t (&fa)[FM], const raw_u32x4_t (&fb)[FN]) {
#pragma unroll
    for (int i = 1; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) mma1(fa, fb, i, j);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };

  // ---- prologue: slabs 0..2 in flight, slab 0 landed
#pragma unroll
  for (int i = 0; i < D; ++i) { stage(i, 0); stage(i, 1); }
  if constexpr (VAR == 2) wait_vm<(D - 1) * (GL + 1)>(); else wait_vm<2 * GL>();
  bar();

  int c_ks = 0, c_it = 0;
  const unsigned strip = lds_base + STRIP_OFF + wid * 4096;
  auto strip_off = [](int row, int quad) { return row * 128 + ((quad ^ (row & 7)) << 4); };
  bool after_epi = false;                                     // the epilogue's stores are younger than slabs 1, 2 of the next tile
  // optional timeline: waves 0 and 4 of block 0 accumulate the s_memtime deltas between the 4 stamps of a slab in SGPRs
  unsigned long long tl_prev = 0, tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int tl_n = 0;
  auto stamp = [&]() {
    if constexpr (TIMING) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      if (tl_n > 16) tl_acc[(tl_n - 1) & 7] += t - tl_prev;       // skip the first two slabs; delta q = stamp q+1 - stamp q
      tl_prev = t;
      ++tl_n;
    }
  };

  auto epilogue = [&]() {
      int m0, n0;
      tile_of(c_it, m0, n0);
      ++c_it;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA results -> raw ds_write (no hazard recogniser in asm)
      // bias of this lane's 4 columns in each 32-column fragment: the DMA put the wave's 64 values at the strip's start
      float4 b4[FN];
      {
        raw_u32x4_t t[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) t[j] = lds_read16_raw(strip + j * 128 + (lane & 7) * 16);
        lds_wait<0>();
#pragma unroll
        for (int j = 0; j < FN; ++j) b4[j] = make_float4(__uint_as_float(t[j].x), __uint_as_float(t[j].y), __uint_as_float(t[j].z), __uint_as_float(t[j].w));
      }
      const int qq = lane & 7, rr = lane >> 3;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int gn = n0 + wc * WTN + j * 32 + qq * 4;
          uint2 rq[4];
          if (HAS_RES) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
              const int gm = m0 + wr * WTM + i * 32 + ps * 8 + rr;
              const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rres, (gm < p.m && gn < p.n) ? (int)(((long)gm * p.ldr + gn) * 2) : kInv, 0, 0);
              rq[ps] = make_uint2(v.x, v.y);
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g)
            lds_write16_raw(strip + strip_off(l31, 2 * g + hi), acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
          raw_u32x4_t tq[4];
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) tq[ps] = lds_read16_raw(strip + strip_off(ps * 8 + rr, qq));
          lds_wait<0>();
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int gm = m0 + wr * WTM + i * 32 + ps * 8 + rr;
            float v[4] = {__uint_as_float(tq[ps].x) + b4[j].x, __uint_as_float(tq[ps].y) + b4[j].y,
                          __uint_as_float(tq[ps].z) + b4[j].z, __uint_as_float(tq[ps].w) + b4[j].w};
            if (HAS_RES) {
              float r4[4];
              unpack4<bf16_tag>(rq[ps], r4);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += r4[e];
            }
            __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){pack2<bf16_tag>(v[0], v[1]), pack2<bf16_tag>(v[2], v[3])}, ro,
                                                  (gm < p.m && gn < p.n) ? (int)(((long)gm * p.ldo + gn) * 2) : kInv, 0, 0);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      after_epi = true;
  };

  if constexpr (VAR == 0 || (VAR >= 10 && VAR != 21)) {
    // ======== variant 0: read -> wait -> barrier -> MFMA -> barrier, one fragment set
    if (PPMODE == 1 && grp == 1) bar();                       // stagger: group 1 runs one barrier behind group 0
    for (int s = 0; s < S; ++s) {
      const int slot = s & 3, fill = (s + 3) & 3;
      const unsigned sa = lds_base + slot * SLOT;
      stamp();
      if constexpr (VAR != 11 && VAR < 13) read_frags(sa, 0, af, bf);
      if constexpr (VAR != 10 && VAR < 13) stage(fill, 0);                                         // A part of slab s+3 (its slot was last read in slab s-1)
      lds_wait<0>();
      stamp();
      if constexpr (VAR != 15) bar();
      stamp();
      if constexpr (VAR != 12) mma(af, bf); else { asm volatile("" :: "v"(af[0]), "v"(bf[0]), "v"(af[FM-1]), "v"(bf[FN-1])); }
      stamp();
      if constexpr (VAR != 15) bar();
      stamp();
      if constexpr (VAR != 11 && VAR < 13) read_frags(sa, 1, af, bf);
      if constexpr (VAR != 10 && VAR < 13) stage(fill, 1);
      stamp();
      // slab s+1 must have landed (own loads); younger: slabs s+2, s+3 (+ the bias load when c_ks < 2, + the last epilogue's stores)
      if constexpr (VAR != 10 && VAR < 13) { if (c_ks < 2) { if (after_epi) wait_vm<2 * GL + 1 + EST>(); else wait_vm<2 * GL + 1>(); }
      else wait_vm<2 * GL>(); }
      stamp();
      lds_wait<0>();
      if constexpr (VAR != 15) bar();
      stamp();
      if constexpr (VAR != 12) mma(af, bf); else { asm volatile("" :: "v"(af[0]), "v"(bf[0]), "v"(af[FM-1]), "v"(bf[FN-1])); }
      if constexpr (VAR != 15) bar();
      if (++c_ks == KS) {
        c_ks = 0;
        if (PPMODE == 1 && grp == 0) bar();                   // both groups run the epilogue together ...
        epilogue();
        if (PPMODE == 1 && grp == 1) bar();                   // ... and group 1 falls one barrier behind again
      }
    }
  } else if constexpr (VAR == 2) {
    // ======== variant 2: variant 0 with an NSLOT-deep ring (prefetch distance D) and a uniform load count per slab
    constexpr int G1 = GL + 1;
    if (PPMODE == 1 && grp == 1) bar();
    int slot = 0, fill = D;
    for (int s = 0; s < S; ++s) {
      const unsigned sa = lds_base + slot * SLOT;
      read_frags(sa, 0, af, bf);
      stage(fill, 0);
      lds_wait<0>();
      bar();
      mma(af, bf);
      bar();
      read_frags(sa, 1, af, bf);
      stage(fill, 1);
      if (after_epi && c_ks <= D - 2) wait_vm<(D - 1) * G1 + EST>(); else wait_vm<(D - 1) * G1>();
      lds_wait<0>();
      bar();
      mma(af, bf);
      bar();
      slot = slot + 1 == NSLOT ? 0 : slot + 1;
      fill = fill + 1 == NSLOT ? 0 : fill + 1;
      if (++c_ks == KS) {
        c_ks = 0;
        if (PPMODE == 1 && grp == 0) bar();
        epilogue();
        if (PPMODE == 1 && grp == 1) bar();
      }
    }
  } else {
    // ======== variant 1: two fragment sets; the reads of the NEXT phase are issued under this phase's MFMAs, so the
    // non-MFMA segment of a wave is only: DMA issue + counted waits
    raw_u32x4_t af2[FM], bf2[FN];
    read_frags(lds_base, 0, af, bf);
    if (PPMODE == 1 && grp == 1) bar();
    for (int s = 0; s < S; ++s) {
      const int slot = s & 3, fill = (s + 3) & 3;
      const unsigned sa = lds_base + slot * SLOT;
      stamp();
      // ---- R(s,0): A part of slab s+3; slab s+1 landed (it is read from M(s,1) on); fragments (s,0) landed
      stage(fill, 0);
      stamp();
      if (c_ks == 0) { if (after_epi) wait_vm<GL + AR + EST>(); else wait_vm<GL + AR>(); }
      else if (c_ks == 1) { if (after_epi) wait_vm<GL + AR + 1 + EST>(); else wait_vm<GL + AR + 1>(); }
      else wait_vm<GL + AR>();
      stamp();
      lds_wait<0>();
      bar();
      stamp();
      // ---- M(s,0)
      mma_head(af, bf);
      read_frags(sa, 1, af2, bf2);
      __builtin_amdgcn_sched_barrier(0);
      mma_tail(af, bf);
      stamp();
      bar();
      stamp();
      // ---- R(s,1): W part (+ bias) of slab s+3
      stage(fill, 1);
      lds_wait<0>();
      bar();
      stamp();
      // ---- M(s,1): fragments (s+1, 0) are read under the MFMAs (past the last slab: stale data, never used)
      mma_head(af2, bf2);
      read_frags(lds_base + ((s + 1) & 3) * SLOT, 0, af, bf);
      __builtin_amdgcn_sched_barrier(0);
      mma_tail(af2, bf2);
      bar();
      if (++c_ks == KS) {
        c_ks = 0;
        if (PPMODE == 1 && grp == 0) bar();
        epilogue();
        if (PPMODE == 1 && grp == 1) bar();
      }
    }
  }
  if constexpr (TIMING) {
    if (blockIdx.x == 0 && (wid == 0 || wid == 4) && lane == 0) {
      unsigned long long* dst = p.tl + (wid == 4 ? 16 : 0);
      for (int q = 0; q < 8; ++q) dst[q] = tl_acc[q];
      dst[8] = tl_n;
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// =====================================================================================================================
// gemm_q: 256x256 tile, K slabs of 64 (128-byte rows = full cache lines per DMA piece), TWO 64 KiB slots.  A slab is
// consumed in 4 quadrant phases (row half i2 x column fragment j of the wave's 128x64 tile, all 64 k each = 8 MFMAs):
//   p0 (0,0): reads A-lo (8) + B-lo (4)   p1 (0,1): reads B-hi (4)   p2 (1,1): reads A-hi (8)   p3 (1,0): reads B-lo (4)
// so the four 16 KiB regions of a slot are released one per phase and re-staged (2 DMA pieces per thread) one phase later
// for slab s+2: three to four regions are always in flight and the only counted wait is vmcnt(6) at the end of p3.
template <int PPMODE, int HAS_RES, int ABL>
__global__ __launch_bounds__(512, 2) void gemm_q(const PP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 256, WGN = 4, WTM = 128, WTN = 64, FM = 4, FN = 2, CPR = 8;
  constexpr int REG = 16384, SLOT = 4 * REG, STRIP_OFF = 2 * SLOT;      // regions: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi
  constexpr int EST = FM * FN * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wr = wid >> 2, wc = wid & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  const __amdgpu_buffer_rsrc_t ra = mk_rsrc(p.a, p.a_bytes), rw = mk_rsrc(p.w, p.w_bytes);
  const __amdgpu_buffer_rsrc_t rb = mk_rsrc(p.bias, p.bias_bytes), ro = mk_rsrc(p.out, p.out_bytes);
  const __amdgpu_buffer_rsrc_t rres = mk_rsrc(p.res, HAS_RES ? p.res_bytes : 0);
  const int KS = p.k >> 6;
  const int nwg = gridDim.x;
  const int my_tiles = (int)blockIdx.x < p.ntiles ? (p.ntiles - 1 - (int)blockIdx.x) / nwg + 1 : 0;
  const int S = my_tiles * KS;
  auto tile_of = [&](int it, int& m0, int& n0) {
    int bid = it * nwg + (int)blockIdx.x;
    const int q = p.ntiles >> 3, r = p.ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gm = p.group_m, per_group = gm * p.tiles_n;
    const int g = bid / per_group, first = g * gm, rows = min(gm, p.tiles_m - first), rem = bid - g * per_group;
    const int tn = rem / rows, tm = first + (rem - tn * rows);
    m0 = tm * BM; n0 = tn * BN;
  };
  // ---- producer: per-lane source offsets of the 2 pieces of each region for the producer's tile
  // region row rr (0..127) -> tile row: A-lo (rr/64)*128 + rr%64, A-hi +64 ; B-lo (rr/32)*64 + rr%32, B-hi +32
  int pv[4][2];
  int p_it = 0, p_ks = 0;
  auto producer_tile = [&](int it) {
    int m0 = 0, n0 = 0;
    const bool ok = it < my_tiles;
    if (ok) tile_of(it, m0, n0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = i * 512 + tid, rr = c >> 3, ch = (c & 7) ^ tile_swz<CPR>(rr);
      const int ar = (rr >> 6) * 128 + (rr & 63), br = (rr >> 5) * 64 + (rr & 31);
      pv[0][i] = (ok && m0 + ar < p.m) ? (int)(((long)(m0 + ar) * p.lda + ch * 8) * 2) : kInv;
      pv[1][i] = (ok && m0 + ar + 64 < p.m) ? (int)(((long)(m0 + ar + 64) * p.lda + ch * 8) * 2) : kInv;
      pv[2][i] = (ok && n0 + br < p.n) ? (int)(((long)(n0 + br) * p.ldw + ch * 8) * 2) : kInv;
      pv[3][i] = (ok && n0 + br + 32 < p.n) ? (int)(((long)(n0 + br + 32) * p.ldw + ch * 8) * 2) : kInv;
    }
  };
  producer_tile(0);
  // stage region `reg` (compile-time) of the producer's slab into `slot`; region 2 (B-lo) is the last of a slab: advance
  auto stage = [&](int slot, auto reg_tag) {
    constexpr int R = decltype(reg_tag)::value;
    if constexpr (ABL == 1 || ABL == 6) return;
    char* base = smem + slot * SLOT + R * REG + wid * 1024;
    const int soff = __builtin_amdgcn_readfirstlane(p_ks) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = pv[R][i];
      if (R < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + i * 8192), 16, v, soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + i * 8192), 16, v, soff, 0, 0);
    }
    if constexpr (R == 2) { if (++p_ks == KS) { p_ks = 0; ++p_it; producer_tile(p_it); } }
  };
  using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>; using R3 = std::integral_constant<int, 3>;

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds_base = lds_addr(smem);
  // fragment addresses inside a slot: A half i2 -> region i2, rows wr*64 + i*32 + l31 ; B frag j -> region 2+j, rows wc*32 + l31
  unsigned a_addr[2][4], b_addr[4];                            // [frag i][ks] / [ks]: offsets without the region / slot base
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = wr * 64 + i * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_addr[i][ks] = rr * 128 + (((ks * 2 + hi) ^ tile_swz<CPR>(rr)) << 4);
  }
  {
    const int rr = wc * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_addr[ks] = rr * 128 + (((ks * 2 + hi) ^ tile_swz<CPR>(rr)) << 4);
  }
  raw_u32x4_t af[2][4], bf[4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) af[i][ks] = (raw_u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)lane};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bf[ks] = (raw_u32x4_t){0x3f803f80u, 0x3f803f80u, (unsigned)lane, 0x3f803f80u};
  if constexpr (ABL == 6) {
    unsigned x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return (x & 0x7fff7fffu) % 0x3f803f80u | (x & 0x80008000u); };   // two bf16 in (-1, 1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[i][ks] = (raw_u32x4_t){rnd(), rnd(), rnd(), rnd()};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bf[ks] = (raw_u32x4_t){rnd(), rnd(), rnd(), rnd()};
  }
  auto read_a = [&](unsigned sbase, int i2) {
    if constexpr (ABL == 2 || ABL == 6) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[i][ks] = lds_read16_raw(sbase + i2 * REG + a_addr[i][ks]);
  };
  auto read_b = [&](unsigned sbase, int j) {
    if constexpr (ABL == 2 || ABL == 6) return;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bf[ks] = lds_read16_raw(sbase + (2 + j) * REG + b_addr[ks]);
  };
  auto mma = [&](auto i2_tag, auto j_tag) {
    constexpr int I2 = decltype(i2_tag)::value, J = decltype(j_tag)::value;
    if constexpr (ABL == 3) { asm volatile("" ::"v"(af[0][0]), "v"(af[1][3]), "v"(bf[0]), "v"(bf[3])); return; }
    if constexpr (ABL != 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[I2 * 2 + i][J] = Cvt<bf16_tag>::mfma32(make_uint4(bf[ks].x, bf[ks].y, bf[ks].z, bf[ks].w),
                                                   make_uint4(af[i][ks].x, af[i][ks].y, af[i][ks].z, af[i][ks].w), acc[I2 * 2 + i][J]);
    if constexpr (ABL != 4) __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  // ---- prologue: slab 0 (slot 0) complete + A-lo, B-hi, A-hi of slab 1 (slot 1), as the steady state would have issued them
  // (slab 0's regions must all belong to the same producer slab: B-lo advances the cursor, so it goes last)
  stage(0, R0{}); stage(0, R3{}); stage(0, R1{}); stage(0, R2{});
  stage(1, R0{}); stage(1, R3{}); stage(1, R1{});
  wait_vm<6>();
  bar();
  if (PPMODE == 1 && grp == 1) bar();

  int c_ks = 0, c_it = 0;
  const unsigned strip = lds_base + STRIP_OFF + wid * 4096;
  auto strip_off = [](int row, int quad) { return row * 128 + ((quad ^ (row & 7)) << 4); };

  auto epilogue = [&]() {
    int m0, n0;
    tile_of(c_it, m0, n0);
    ++c_it;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float4 b4[FN];
    {
      raw_u32x4_t t[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) t[j] = lds_read16_raw(strip + j * 128 + (lane & 7) * 16);
      lds_wait<0>();
#pragma unroll
      for (int j = 0; j < FN; ++j) b4[j] = make_float4(__uint_as_float(t[j].x), __uint_as_float(t[j].y), __uint_as_float(t[j].z), __uint_as_float(t[j].w));
    }
    const int qq = lane & 7, rr = lane >> 3;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int gn = n0 + wc * WTN + j * 32 + qq * 4;
        uint2 rq[4];
        if (HAS_RES) {
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int gm = m0 + wr * WTM + i * 32 + ps * 8 + rr;
            const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rres, (gm < p.m && gn < p.n) ? (int)(((long)gm * p.ldr + gn) * 2) : kInv, 0, 0);
            rq[ps] = make_uint2(v.x, v.y);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
          lds_write16_raw(strip + strip_off(l31, 2 * g + hi), acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        raw_u32x4_t tq[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) tq[ps] = lds_read16_raw(strip + strip_off(ps * 8 + rr, qq));
        lds_wait<0>();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int gm = m0 + wr * WTM + i * 32 + ps * 8 + rr;
          float v[4] = {__uint_as_float(tq[ps].x) + b4[j].x, __uint_as_float(tq[ps].y) + b4[j].y,
                        __uint_as_float(tq[ps].z) + b4[j].z, __uint_as_float(tq[ps].w) + b4[j].w};
          if (HAS_RES) {
            float r4[4];
            unpack4<bf16_tag>(rq[ps], r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r4[e];
          }
          __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){pack2<bf16_tag>(v[0], v[1]), pack2<bf16_tag>(v[2], v[3])}, ro,
                                                (gm < p.m && gn < p.n) ? (int)(((long)gm * p.ldo + gn) * 2) : kInv, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int tn = 0;
  unsigned long long clk0 = 0, rt0 = 0;
  if constexpr (ABL == 5) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0), "=s"(rt0)::"memory");
  for (int s = 0; s < S; ++s) {
    const int slot = s & 1;
    const unsigned sb = lds_base + slot * SLOT;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, t8 = 0;
#define STAMP(t) do { if constexpr (ABL == 5) asm volatile("s_memtime %0" : "=s"(t)::"memory"); } while (0)
    STAMP(t0);
    // ---- p0 (0,0): A-lo + B-lo ; DMA: [bias of this tile] + B-lo of slab s+1 (the other slot; released in p3 of slab s-1)
    read_a(sb, 0); read_b(sb, 0);
    if (c_ks == 0 && ABL != 1 && ABL != 6) {
      const int it = c_it < my_tiles ? c_it : 0;
      int m0, n0;
      tile_of(it, m0, n0);
      const int gn = n0 + wc * 64 + lane;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(smem + STRIP_OFF + wid * 4096), 4,
                                               gn < p.n ? gn * 4 : kInv, 0, 0, 0);
    }
    stage(slot ^ 1, R2{});
    lds_wait<0>();
    STAMP(t1);
    bar();
    STAMP(t2);
    mma(I0{}, I0{});
    STAMP(t3);
    bar();
    STAMP(t4);
    // ---- p1 (0,1): B-hi ; DMA: A-lo of slab s+2 (this slot; released in p0)
    read_b(sb, 1);
    stage(slot, R0{});
    lds_wait<0>();
    STAMP(t5);
    bar();
    STAMP(t6);
    mma(I0{}, I1{});
    STAMP(t7);
    bar();
    STAMP(t8);
    // ---- p2 (1,1): A-hi ; DMA: B-hi of slab s+2 (released in p1)
    read_a(sb, 1);
    stage(slot, R3{});
    lds_wait<0>();
    bar();
    mma(I1{}, I1{});
    bar();
    // ---- p3 (1,0): B-lo again ; DMA: A-hi of slab s+2 (released in p2) ; everything up to B-lo of slab s+1 has landed
    read_b(sb, 0);
    stage(slot, R1{});
    if constexpr (ABL != 1 && ABL != 6) wait_vm<6>();
    lds_wait<0>();
    if constexpr (ABL == 5) {       // every s_memtime above has returned (lgkmcnt(0)); accumulate the 8 deltas
      asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(t4), "+s"(t5), "+s"(t6), "+s"(t7), "+s"(t8));
      if (s >= 4) { tacc[0] += t1 - t0; tacc[1] += t2 - t1; tacc[2] += t3 - t2; tacc[3] += t4 - t3; tacc[4] += t5 - t4; tacc[5] += t6 - t5; tacc[6] += t7 - t6; tacc[7] += t8 - t7; ++tn; }
    }
    bar();
    mma(I1{}, I0{});
    bar();
    if (++c_ks == KS) {
      c_ks = 0;
      if (PPMODE == 1 && grp == 0) bar();
      epilogue();
      if (PPMODE == 1 && grp == 1) bar();
    }
  }
  if constexpr (ABL == 5) {
    if (blockIdx.x == 0 && (wid == 0 || wid == 4 || wid == 3 || wid == 7) && lane == 0) {
      unsigned long long* dst = p.tl + (wid == 0 ? 0 : wid == 4 ? 16 : wid == 3 ? 32 : 48);
      for (int q = 0; q < 8; ++q) dst[q] = tacc[q];
      dst[8] = tn;
      unsigned long long clk1, rt1;
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1), "=s"(rt1)::"memory");
      dst[9] = clk1 - clk0; dst[10] = rt1 - rt0;
    }
  }
}

template <int PPMODE, int HAS_RES, int ABL>
__global__ __launch_bounds__(512, 2) void gemm_q2(const PP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 256, WGN = 4, WTM = 128, WTN = 64, FM = 4, FN = 2, CPR = 8;
  constexpr int REG = 16384, SLOT = 4 * REG, STRIP_OFF = 2 * SLOT;      // regions: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi
  constexpr int EST = FM * FN * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wr = wid >> 2, wc = wid & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  const __amdgpu_buffer_rsrc_t ra = mk_rsrc(p.a, p.a_bytes), rw = mk_rsrc(p.w, p.w_bytes);
  const __amdgpu_buffer_rsrc_t rb = mk_rsrc(p.bias, p.bias_bytes), ro = mk_rsrc(p.out, p.out_bytes);
  const __amdgpu_buffer_rsrc_t rres = mk_rsrc(p.res, HAS_RES ? p.res_bytes : 0);
  const int KS = p.k >> 6;
  const int nwg = gridDim.x;
  const int my_tiles = (int)blockIdx.x < p.ntiles ? (p.ntiles - 1 - (int)blockIdx.x) / nwg + 1 : 0;
  const int S = my_tiles * KS;
  auto tile_of = [&](int it, int& m0, int& n0) {
    int bid = it * nwg + (int)blockIdx.x;
    const int q = p.ntiles >> 3, r = p.ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gm = p.group_m, per_group = gm * p.tiles_n;
    const int g = bid / per_group, first = g * gm, rows = min(gm, p.tiles_m - first), rem = bid - g * per_group;
    const int tn = rem / rows, tm = first + (rem - tn * rows);
    m0 = tm * BM; n0 = tn * BN;
  };
  // ---- producer: per-lane source offsets of the 2 pieces of each region for the producer's tile
  // region row rr (0..127) -> tile row: A-lo (rr/64)*128 + rr%64, A-hi +64 ; B-lo (rr/32)*64 + rr%32, B-hi +32
  int pv[4][2];
  int p_it = 0, p_ks = 0;
  auto producer_tile = [&](int it) {
    int m0 = 0, n0 = 0;
    const bool ok = it < my_tiles;
    if (ok) tile_of(it, m0, n0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = i * 512 + tid, rr = c >> 3, ch = (c & 7) ^ tile_swz<CPR>(rr);
      const int ar = (rr >> 6) * 128 + (rr & 63), br = (rr >> 5) * 64 + (rr & 31);
      pv[0][i] = (ok && m0 + ar < p.m) ? (int)(((long)(m0 + ar) * p.lda + ch * 8) * 2) : kInv;
      pv[1][i] = (ok && m0 + ar + 64 < p.m) ? (int)(((long)(m0 + ar + 64) * p.lda + ch * 8) * 2) : kInv;
      pv[2][i] = (ok && n0 + br < p.n) ? (int)(((long)(n0 + br) * p.ldw + ch * 8) * 2) : kInv;
      pv[3][i] = (ok && n0 + br + 32 < p.n) ? (int)(((long)(n0 + br + 32) * p.ldw + ch * 8) * 2) : kInv;
    }
  };
  producer_tile(0);
  // stage region `reg` (compile-time) of the producer's slab into `slot`; region 2 (B-lo) is the last of a slab: advance
  auto stage = [&](int slot, auto reg_tag) {
    constexpr int R = decltype(reg_tag)::value;
    if constexpr (ABL == 1 || ABL == 6) return;
    char* base = smem + slot * SLOT + R * REG + wid * 1024;
    const int soff = __builtin_amdgcn_readfirstlane(p_ks) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = pv[R][i];
      if (R < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + i * 8192), 16, v, soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + i * 8192), 16, v, soff, 0, 0);
    }
    if constexpr (R == 2) { if (++p_ks == KS) { p_ks = 0; ++p_it; producer_tile(p_it); } }
  };
  using R0 = std::integral_constant<int, 0>; using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>; using R3 = std::integral_constant<int, 3>;

  f32x16_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds_base = lds_addr(smem);
  // fragment addresses inside a slot: A half i2 -> region i2, rows wr*64 + i*32 + l31 ; B frag j -> region 2+j, rows wc*32 + l31
  unsigned a_addr[2][4], b_addr[4];                            // [frag i][ks] / [ks]: offsets without the region / slot base
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = wr * 64 + i * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_addr[i][ks] = rr * 128 + (((ks * 2 + hi) ^ tile_swz<CPR>(rr)) << 4);
  }
  {
    const int rr = wc * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_addr[ks] = rr * 128 + (((ks * 2 + hi) ^ tile_swz<CPR>(rr)) << 4);
  }
  raw_u32x4_t af[2][4], bf[4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) af[i][ks] = (raw_u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)lane};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bf[ks] = (raw_u32x4_t){0x3f803f80u, 0x3f803f80u, (unsigned)lane, 0x3f803f80u};
  if constexpr (ABL == 6) {
    unsigned x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return (x & 0x7fff7fffu) % 0x3f803f80u | (x & 0x80008000u); };   // two bf16 in (-1, 1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[i][ks] = (raw_u32x4_t){rnd(), rnd(), rnd(), rnd()};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bf[ks] = (raw_u32x4_t){rnd(), rnd(), rnd(), rnd()};
  }
  auto read_a = [&](unsigned sbase, int i2) {
    if constexpr (ABL == 2 || ABL == 6) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[i][ks] = lds_read16_raw(sbase + i2 * REG + a_addr[i][ks]);
  };
  auto read_b = [&](unsigned sbase, int j) {
    if constexpr (ABL == 2 || ABL == 6) return;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bf[ks] = lds_read16_raw(sbase + (2 + j) * REG + b_addr[ks]);
  };
  auto mma = [&](auto i2_tag, auto j_tag) {
    constexpr int I2 = decltype(i2_tag)::value, J = decltype(j_tag)::value;
    if constexpr (ABL == 3) { asm volatile("" ::"v"(af[0][0]), "v"(af[1][3]), "v"(bf[0]), "v"(bf[3])); return; }
    if constexpr (ABL != 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[I2 * 2 + i][J] = Cvt<bf16_tag>::mfma32(make_uint4(bf[ks].x, bf[ks].y, bf[ks].z, bf[ks].w),
                                                   make_uint4(af[i][ks].x, af[i][ks].y, af[i][ks].z, af[i][ks].w), acc[I2 * 2 + i][J]);
    if constexpr (ABL != 4) __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  // ---- prologue: slab 0 (slot 0) complete + A-lo, B-hi, A-hi of slab 1 (slot 1), as the steady state would have issued them
  // (slab 0's regions must all belong to the same producer slab: B-lo advances the cursor, so it goes last)
  stage(0, R0{}); stage(0, R3{}); stage(0, R1{}); stage(0, R2{});
  stage(1, R0{}); stage(1, R3{}); stage(1, R1{});
  wait_vm<6>();
  bar();

  int c_ks = 0, c_it = 0;
  const unsigned strip = lds_base + STRIP_OFF + wid * 4096;
  auto strip_off = [](int row, int quad) { return row * 128 + ((quad ^ (row & 7)) << 4); };

  auto epilogue = [&]() {
    int m0, n0;
    tile_of(c_it, m0, n0);
    ++c_it;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float4 b4[FN];
    {
      raw_u32x4_t t[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) t[j] = lds_read16_raw(strip + j * 128 + (lane & 7) * 16);
      lds_wait<0>();
#pragma unroll
      for (int j = 0; j < FN; ++j) b4[j] = make_float4(__uint_as_float(t[j].x), __uint_as_float(t[j].y), __uint_as_float(t[j].z), __uint_as_float(t[j].w));
    }
    const int qq = lane & 7, rr = lane >> 3;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int gn = n0 + wc * WTN + j * 32 + qq * 4;
        uint2 rq[4];
        if (HAS_RES) {
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int gm = m0 + wr * WTM + i * 32 + ps * 8 + rr;
            const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rres, (gm < p.m && gn < p.n) ? (int)(((long)gm * p.ldr + gn) * 2) : kInv, 0, 0);
            rq[ps] = make_uint2(v.x, v.y);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
          lds_write16_raw(strip + strip_off(l31, 2 * g + hi), acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
        raw_u32x4_t tq[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) tq[ps] = lds_read16_raw(strip + strip_off(ps * 8 + rr, qq));
        lds_wait<0>();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int gm = m0 + wr * WTM + i * 32 + ps * 8 + rr;
          float v[4] = {__uint_as_float(tq[ps].x) + b4[j].x, __uint_as_float(tq[ps].y) + b4[j].y,
                        __uint_as_float(tq[ps].z) + b4[j].z, __uint_as_float(tq[ps].w) + b4[j].w};
          if (HAS_RES) {
            float r4[4];
            unpack4<bf16_tag>(rq[ps], r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r4[e];
          }
          __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){pack2<bf16_tag>(v[0], v[1]), pack2<bf16_tag>(v[2], v[3])}, ro,
                                                (gm < p.m && gn < p.n) ? (int)(((long)gm * p.ldo + gn) * 2) : kInv, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  // M-segment: the 8 MFMAs of quadrant (I2, J); right after the two MFMAs of k-step ks have issued, the fragment registers of
  // that k-step are refilled IN PLACE with what the next phase needs (NA: A half `na`, NB: B fragment `nb`, from slot base nsb)
  auto mma_refill = [&](auto i2_tag, auto j_tag, auto na_tag, auto nb_tag, unsigned nsb) {
    constexpr int I2 = decltype(i2_tag)::value, J = decltype(j_tag)::value, NA = decltype(na_tag)::value, NB = decltype(nb_tag)::value;
    if constexpr (ABL != 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if constexpr (ABL != 3)
          acc[I2 * 2 + i][J] = Cvt<bf16_tag>::mfma32(make_uint4(bf[ks].x, bf[ks].y, bf[ks].z, bf[ks].w),
                                                     make_uint4(af[i][ks].x, af[i][ks].y, af[i][ks].z, af[i][ks].w), acc[I2 * 2 + i][J]);
      if constexpr (ABL == 3) asm volatile("" ::"v"(af[0][ks]), "v"(af[1][ks]), "v"(bf[ks]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABL != 2) {
        if constexpr (NA >= 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[i][ks] = lds_read16_raw(nsb + NA * REG + a_addr[i][ks]);
        }
        if constexpr (NB >= 0) bf[ks] = lds_read16_raw(nsb + (2 + NB) * REG + b_addr[ks]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ABL != 4) __builtin_amdgcn_s_setprio(0);
  };
  using N_ = std::integral_constant<int, -1>;
  read_a(lds_base, 0); read_b(lds_base, 0);                  // fragments of p0 of slab 0 (retired by the wait in R(p0))
  if (PPMODE == 1 && grp == 1) bar();
  for (int s = 0; s < S; ++s) {
    const int slot = s & 1;
    const unsigned sb = lds_base + slot * SLOT, sbn = lds_base + (slot ^ 1) * SLOT;
    // ---- p0 (0,0) ; DMA: [bias of this tile] + B-lo of slab s+1 ; under its MFMAs: B-hi -> bf
    if (c_ks == 0 && ABL != 1 && ABL != 6) {
      const int it = c_it < my_tiles ? c_it : 0;
      int m0, n0;
      tile_of(it, m0, n0);
      const int gn = n0 + wc * 64 + lane;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(smem + STRIP_OFF + wid * 4096), 4,
                                               gn < p.n ? gn * 4 : kInv, 0, 0, 0);
    }
    stage(slot ^ 1, R2{});
    lds_wait<0>();
    bar();
    mma_refill(I0{}, I0{}, N_{}, I1{}, sb);
    bar();
    // ---- p1 (0,1) ; DMA: A-lo of slab s+2 ; under its MFMAs: A-hi -> af
    stage(slot, R0{});
    lds_wait<0>();
    bar();
    mma_refill(I0{}, I1{}, I1{}, N_{}, sb);
    bar();
    // ---- p2 (1,1) ; DMA: B-hi of slab s+2 ; B-lo of slab s+1 (and everything older) has landed ; under its MFMAs: B-lo -> bf
    stage(slot, R3{});
    if constexpr (ABL != 1) wait_vm<4>();
    lds_wait<0>();
    bar();
    mma_refill(I1{}, I1{}, N_{}, I0{}, sb);
    bar();
    // ---- p3 (1,0) ; DMA: A-hi of slab s+2 ; under its MFMAs: A-lo and B-lo of slab s+1 -> af, bf
    stage(slot, R1{});
    lds_wait<0>();
    bar();
    mma_refill(I1{}, I0{}, I0{}, I0{}, sbn);
    bar();
    if (++c_ks == KS) {
      c_ks = 0;
      if (PPMODE == 1 && grp == 0) bar();
      epilogue();
      if (PPMODE == 1 && grp == 1) bar();
    }
  }
}

template <int PPMODE, int HAS_RES, int ABL>
void launch_q2(PP& p, int grid_cap, hipStream_t st) {
  constexpr int lds = 2 * 65536 + 8 * 4096;
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)gemm_q2<PPMODE, HAS_RES, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); done = true; }
  p.tiles_m = (p.m + 255) / 256; p.tiles_n = (p.n + 255) / 256; p.ntiles = p.tiles_m * p.tiles_n;
  int gm = 1;
  if (p.tiles_n > 8) while (gm * 2 * 256 * gm * 2 <= 32L * 256 && gm * 2 <= p.tiles_m) gm *= 2;
  p.group_m = gm;
  const int grid = p.ntiles < grid_cap ? p.ntiles : grid_cap;
  hipLaunchKernelGGL((gemm_q2<PPMODE, HAS_RES, ABL>), dim3(grid), dim3(512), lds, st, p);
}

template <int PPMODE, int HAS_RES, int ABL>
void launch_q(PP& p, int grid_cap, hipStream_t st) {
  constexpr int lds = 2 * 65536 + 8 * 4096;
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)gemm_q<PPMODE, HAS_RES, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); done = true; }
  p.tiles_m = (p.m + 255) / 256; p.tiles_n = (p.n + 255) / 256; p.ntiles = p.tiles_m * p.tiles_n;
  int gm = 1;
  if (p.tiles_n > 8) while (gm * 2 * 256 * gm * 2 <= 32L * 256 && gm * 2 <= p.tiles_m) gm *= 2;
  p.group_m = gm;
  const int grid = p.ntiles < grid_cap ? p.ntiles : grid_cap;
  hipLaunchKernelGGL((gemm_q<PPMODE, HAS_RES, ABL>), dim3(grid), dim3(512), lds, st, p);
}

// ---- naive check kernel: sampled outputs in fp32
__global__ void ref_samples(const PP p, const int* rows, const int* cols, float* outv, int ns, int has_res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const int r = rows[i], c = cols[i];
  const unsigned short* a = (const unsigned short*)p.a + (long)r * p.lda;
  const unsigned short* w = (const unsigned short*)p.w + (long)c * p.ldw;
  float s = 0.f;
  for (int k = 0; k < p.k; ++k) s = fmaf(Cvt<bf16_tag>::to_f32(a[k]), Cvt<bf16_tag>::to_f32(w[k]), s);
  s += p.bias[c];
  if (has_res) s += Cvt<bf16_tag>::to_f32(((const unsigned short*)p.res)[(long)r * p.ldr + c]);
  outv[i] = s;
}
__global__ void fill_rand(unsigned short* p, long n, unsigned seed, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    const float f = ((x & 0xffffff) / 16777216.0f * 2.f - 1.f) * scale;
    p[i] = Cvt<bf16_tag>::from_f32(f);
  }
}
__global__ void fill_randf(float* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (x & 0xffff) / 65536.0f - 0.5f;
  }
}


template <int BM, int BN, int WGM, int WGN, int PPMODE, int HAS_RES, int VAR, bool TIMING = false, int NSLOT = 4>
void launch(PP& p, int grid_cap, hipStream_t st) {
  constexpr int lds = NSLOT == 5 ? 5 * (BM * 64 + BN * 64) : NSLOT * (BM * 64 + BN * 64) + 8 * 4096;
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)gemm_pp<BM, BN, WGM, WGN, PPMODE, HAS_RES, VAR, TIMING, NSLOT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); done = true; }
  p.tiles_m = (p.m + BM - 1) / BM; p.tiles_n = (p.n + BN - 1) / BN; p.ntiles = p.tiles_m * p.tiles_n;
  int gm = 1;
  if (p.tiles_n > 8) while (gm * 2 * BM * gm * 2 <= 32L * BN && gm * 2 <= p.tiles_m) gm *= 2;
  p.group_m = gm;
  const int grid = p.ntiles < grid_cap ? p.ntiles : grid_cap;
  hipLaunchKernelGGL((gemm_pp<BM, BN, WGM, WGN, PPMODE, HAS_RES, VAR, TIMING, NSLOT>), dim3(grid), dim3(512), lds, st, p);
}

int main(int argc, char** argv) {
  struct Shape { int m, n, k; };
  std::vector<Shape> shapes = {{8192, 8192, 8192}, {4096, 4096, 4096}, {50176, 2560, 320}, {12544, 5120, 640}, {3136, 10240, 1280}, {12544, 640, 2560}, {12544, 640, 640}, {3136, 1280, 1280}};
  int grid_cap = 256;
  if (argc > 1) grid_cap = atoi(argv[1]);
  hipStream_t st; CK(hipStreamCreate(&st));
  for (const Shape& sh : shapes) {
    const int m = sh.m, n = sh.n, k = sh.k;
    // rotate through enough buffers that A / residual / out come from HBM, as in the model
    const long per = ((long)m * k + 2L * m * n) * 2;
    const int nb = (int)std::max(2L, std::min(12L, (600L << 20) / per + 1));
    std::vector<unsigned short*> A(nb), O(nb), R(nb);
    unsigned short* W; float* bias;
    for (int i = 0; i < nb; ++i) {
      CK(hipMalloc(&A[i], (long)m * k * 2)); CK(hipMalloc(&O[i], (long)m * n * 2)); CK(hipMalloc(&R[i], (long)m * n * 2));
      fill_rand<<<2048, 256, 0, st>>>(A[i], (long)m * k, 17 + i, 1.0f);
      fill_rand<<<2048, 256, 0, st>>>(R[i], (long)m * n, 91 + i, 1.0f);
    }
    CK(hipMalloc(&W, (long)n * k * 2)); CK(hipMalloc(&bias, n * 4));
    fill_rand<<<2048, 256, 0, st>>>(W, (long)n * k, 5, 1.0f / sqrtf((float)k));
    fill_randf<<<64, 256, 0, st>>>(bias, n, 3);
    PP p{};
    p.m = m; p.n = n; p.k = k; p.lda = k; p.ldw = k; p.ldo = n; p.ldr = n; p.bias = bias; p.w = (const char*)W;
    p.a_bytes = (unsigned)((long)m * k * 2); p.w_bytes = (unsigned)((long)n * k * 2); p.out_bytes = (unsigned)((long)m * n * 2);
    p.bias_bytes = n * 4; p.res_bytes = p.out_bytes;
    // samples for the check
    const int ns = 4096;
    std::vector<int> hr(ns), hc(ns);
    for (int i = 0; i < ns; ++i) { hr[i] = (int)((i * 2654435761u >> 7) % (unsigned)m); hc[i] = (int)((i * 40503u + 7) % (unsigned)n); }
    for (int i = 0; i < 64; ++i) { hr[i] = m - 1 - (i % 40); hc[i] = n - 1 - (i * 7 % n); }     // ragged edge
    int *dr, *dc; float* dv;
    CK(hipMalloc(&dr, ns * 4)); CK(hipMalloc(&dc, ns * 4)); CK(hipMalloc(&dv, ns * 4));
    CK(hipMemcpy(dr, hr.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), ns * 4, hipMemcpyHostToDevice));
    struct Var { const char* name; void (*fn)(PP&, int, hipStream_t); int res; };
    const Var vars[] = {
      {"Q 256x256x64 ping-pong           ", launch_q<1, 0, 0>, 0},
      {"Q 256x256x64 lockstep            ", launch_q<0, 0, 0>, 0},
      {"Q 256x256x64 ping-pong +res      ", launch_q<1, 1, 0>, 1},
      {"Q MFMA+barriers only, RANDOM frags", launch_q<1, 0, 6>, 0},
      {"Q2 256x256x64 ping-pong          ", launch_q2<1, 0, 0>, 0},
      {"Q2 256x256x64 lockstep           ", launch_q2<0, 0, 0>, 0},
      {"Q2 pp no setprio                 ", launch_q2<1, 0, 4>, 0},
      {"Q2 lockstep no setprio           ", launch_q2<0, 0, 4>, 0},
      {"Q2 pp +res                       ", launch_q2<1, 1, 0>, 1},
      {"Q2 pp NO DMA (abl)               ", launch_q2<1, 0, 1>, 0},
      {"Q2 pp NO LDS reads (abl)         ", launch_q2<1, 0, 2>, 0},
      {"Q2 pp NO MFMA (abl)              ", launch_q2<1, 0, 3>, 0},
      {"Q pp no setprio                  ", launch_q<1, 0, 4>, 0},
      {"Q lockstep no setprio            ", launch_q<0, 0, 4>, 0},
      {"Q pp NO DMA (abl)                ", launch_q<1, 0, 1>, 0},
      {"Q pp NO LDS reads (abl)          ", launch_q<1, 0, 2>, 0},
      {"Q pp NO MFMA (abl)               ", launch_q<1, 0, 3>, 0},
      {"256x256 v0 lockstep   ", launch<256, 256, 2, 4, 0, 0, 0>, 0},
      {"256x256 v0 ping-pong  ", launch<256, 256, 2, 4, 1, 0, 0>, 0},
      {"256x256 v0 pp NO DMA  (ablation)", launch<256, 256, 2, 4, 1, 0, 10>, 0},
      {"256x256 v0 pp NO LDS reads (abl)", launch<256, 256, 2, 4, 1, 0, 11>, 0},
      {"256x256 v0 pp NO MFMA (ablation)", launch<256, 256, 2, 4, 1, 0, 12>, 0},
      {"256x256 v0 lockstep NO DMA (abl)", launch<256, 256, 2, 4, 0, 0, 10>, 0},
      {"256x256 v0 pp no setprio        ", launch<256, 256, 2, 4, 1, 0, 20>, 0},
      {"256x256 v0 lockstep no setprio  ", launch<256, 256, 2, 4, 0, 0, 20>, 0},
      {"256x256 v1 pp no setprio        ", launch<256, 256, 2, 4, 1, 0, 21>, 0},
      {"256x256 v1 lockstep no setprio  ", launch<256, 256, 2, 4, 0, 0, 21>, 0},
      {"256x256 MFMA + barriers only, ping-pong ", launch<256, 256, 2, 4, 1, 0, 13>, 0},
      {"256x256 MFMA + barriers only, lockstep  ", launch<256, 256, 2, 4, 0, 0, 13>, 0},
      {"256x256 MFMA only, no barriers          ", launch<256, 256, 2, 4, 0, 0, 15>, 0},
      {"256x256 v1 ping-pong  ", launch<256, 256, 2, 4, 1, 0, 1>, 0},
      {"256x256 v1 pp +res    ", launch<256, 256, 2, 4, 1, 1, 1>, 1},
      {"128x256 v1 ping-pong  ", launch<128, 256, 2, 4, 1, 0, 1>, 0},
      {"256x128 v1 ping-pong  ", launch<256, 128, 4, 2, 1, 0, 1>, 0},
      {"256x128 v1 pp +res    ", launch<256, 128, 4, 2, 1, 1, 1>, 1},
      {"256x128 v0 pp +res    ", launch<256, 128, 4, 2, 1, 1, 0>, 1},
    };
    printf("M=%d N=%d K=%d (%d rotating buffers)\n", m, n, k, nb);
    for (const Var& v : vars) {
      auto run = [&](int i) {
        p.a = (const char*)A[i % nb]; p.out = (char*)O[i % nb]; p.res = (const char*)R[i % nb];
        v.fn(p, grid_cap, st);
      };
      for (int i = 0; i < nb; ++i) run(i);
      CK(hipStreamSynchronize(st));
      CK(hipGetLastError());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 2 * nb < 10 ? 10 : 2 * nb;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) run(i);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      // check buffer 0
      p.a = (const char*)A[0]; p.out = (char*)O[0]; p.res = (const char*)R[0];
      CK(hipMemsetAsync(O[0], 0xff, (long)m * n * 2, st));
      v.fn(p, grid_cap, st);
      ref_samples<<<(ns + 255) / 256, 256, 0, st>>>(p, dr, dc, dv, ns, v.res);
      CK(hipStreamSynchronize(st));
      std::vector<float> hv(ns);
      CK(hipMemcpy(hv.data(), dv, ns * 4, hipMemcpyDeviceToHost));
      std::vector<unsigned short> ho((long)m * n);
      CK(hipMemcpy(ho.data(), O[0], (long)m * n * 2, hipMemcpyDeviceToHost));
      double maxerr = 0; int bad = 0;
      for (int i = 0; i < ns; ++i) {
        const unsigned short b = ho[(long)hr[i] * n + hc[i]];
        unsigned u = ((unsigned)b) << 16; float got; memcpy(&got, &u, 4);
        const double err = fabs(got - hv[i]), tol = 0.02 + 0.01 * fabs(hv[i]);
        if (!(err <= tol)) ++bad;
        if (err > maxerr || err != err) maxerr = err;
      }
      printf("  %s %8.1f us  %7.1f TFLOP/s   check: max|err| %.4f, %d/%d outside tol\n", v.name, us, 2.0 * m * n * k / us * 1e-6, maxerr, bad, ns);
      fflush(stdout);
    }
    if (m == 4096 || m == 8192 || m == 12544) {
      unsigned long long* dtl; CK(hipMalloc(&dtl, 64 * 8)); CK(hipMemset(dtl, 0, 64 * 8));
      p.tl = dtl; p.a = (const char*)A[0]; p.out = (char*)O[0]; p.res = (const char*)R[0];
      launch_q<1, 0, 5>(p, grid_cap, st);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> tl(64);
      CK(hipMemcpy(tl.data(), dtl, 64 * 8, hipMemcpyDeviceToHost));
      const int wv[4] = {0, 4, 3, 7};
      for (int w = 0; w < 4; ++w) {
        const unsigned long long* t = tl.data() + w * 16;
        const double nn = (double)t[8];
        if (w == 0) printf("  kernel span on wave 0: %llu s_memtime ticks, %llu s_memrealtime ticks (100 MHz) -> %.3f GHz if s_memtime counts shader clocks\n", t[9], t[10], (double)t[9] / ((double)t[10] * 10.0));
        printf("  Q timeline wave %d (cycles): p0: R %.0f | bar %.0f | M %.0f | bar %.0f || p1: R %.0f | bar %.0f | M %.0f | bar %.0f   (2 of 4 phases: %.0f)\n", wv[w],
               t[0] / nn, t[1] / nn, t[2] / nn, t[3] / nn, t[4] / nn, t[5] / nn, t[6] / nn, t[7] / nn, (t[0]+t[1]+t[2]+t[3]+t[4]+t[5]+t[6]+t[7]) / nn);
      }
      CK(hipFree(dtl)); p.tl = nullptr;
    }
    for (int i = 0; i < nb; ++i) { CK(hipFree(A[i])); CK(hipFree(O[i])); CK(hipFree(R[i])); }
    CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(dr)); CK(hipFree(dc)); CK(hipFree(dv));
  }
  return 0;
}
