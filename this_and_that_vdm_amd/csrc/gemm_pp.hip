// gemm_pp: the persistent big-tile member of the tt_gemm family (round 3), for tall-and-wide Linear problems whose epilogue
// needs no per-row operand -- the GEGLU projections (nn.Linear of diffusers' GEGLU, reached from
// svd/diffusion_arch/transformer_temporal.py:323-376 through BasicTransformerBlock / TemporalBasicTransformerBlock.ff) and fused
// Q | K projections, with the LayerNorm in front of them folded in (TtGemmArgs.ln_fold = 1).
//
// Structure (tools/gemm_pp_probe.hip `gemm_q` measured it):
//   * 256 x 256 output tiles, 8 waves (2 x 4, wave tile 128 x 64 = 4 x 2 MFMA fragments of 32 x 32), ONE workgroup per CU,
//     persistent: workgroup b walks tiles b, b + grid, ... in the XCD-contiguous, grouped order of gemm_kernel.h;
//   * K slabs of 64 elements = 128-byte rows: every LDS-DMA piece (buffer_load ... lds, 1 KiB) is 8 full cache lines (pieces of
//     64-byte row segments run at half the texture-addresser rate: tools/dma_shape_test.hip);
//   * two 64 KiB slots of four 16 KiB regions: A rows {0-63, 128-191} ("A-lo": the first 64 rows of each wave row), A-hi,
//     W rows {32 of every 64}: "W-lo" / "W-hi".  A slab is consumed in four quadrant phases of 8 MFMAs each
//         p0 (A-lo x W-lo)   p1 (A-lo x W-hi)   p2 (A-hi x W-hi)   p3 (A-hi x W-lo)
//     so a region is released every phase and re-staged ONE phase later for slab s+2 (2 pieces per thread per phase): three to
//     four regions are always in flight, and the only counted wait is vmcnt(6) at the end of p3;
//   * the two groups of four waves (one wave of each per SIMD) run one barrier apart: while a group issues the 8 MFMAs of a
//     phase, the other reads its fragments and issues DMA, then they swap (s_setprio 1 around the MFMAs);
//   * the ring keeps streaming across tile boundaries; the tile's bias arrives by a dword LDS-DMA into the wave's strip,
//     so the kernel has NO register-destination loads and every wait is counted;
//   * the LayerNorm sums of the operand stream (gemm_kernel.h KMODE 3) run in the short read intervals p1 / p3 on the A half still
//     in registers; in the GEGLU instances each of the four waves of a wave row sums one K step of every slab and the epilogue
//     adds the four partials through the free half of the strips (tools/pp_scan.py: 2.03 -> 1.73 us per slab; no sums: 1.56);
//   * epilogue (both groups together): 1/sigma of the fused LayerNorm,
//     bias, scale, GEGLU (value / gate lane-local, exact-erf GELU), packed to the storage type in the MFMA layout, transposed
//     through a wave-private 4 KiB strip and stored as 16 bytes per lane (plain: full 128-byte lines).
#include <stdlib.h>
#include <type_traits>
#include "gemm_kernel.h"

#ifndef TT_PP_GEGLU_STORE_AUX
#define TT_PP_GEGLU_STORE_AUX 0      // cache policy of the GEGLU output stores (aux: 2 = nt); measured, see DESIGN.md 6.R4
#endif

namespace ttg {

template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_write8_raw(unsigned addr, unsigned a, unsigned b) {
  const raw_u32x2_t v = {a, b};
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <typename Tag, int LNROWS, int GEGLU>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  kernarg_touch<sizeof(GemmP)>();
#ifdef TT_PP_HALF_PROBE   // probe only (never shipped): 128 x 256 tiles = the A-lo half of every quadrant schedule; measures the K-loop rate at 48 KiB of DMA per slab
  constexpr int BM = 128, BN = 256, WTM = 64, WTN = 64, FM = 2, FN = 2, CPR = 8, ES = 2;
  constexpr bool HALF = true;
#else
  constexpr int BM = 256, BN = 256, WTM = 128, WTN = 64, FM = 4, FN = 2, CPR = 8, ES = 2;
  constexpr bool HALF = false;
#endif
  constexpr int REG = 16384, SLOT = 4 * REG, STRIP_OFF = 2 * SLOT;      // regions of a slot: 0 A-lo, 1 A-hi, 2 W-lo, 3 W-hi
  constexpr bool SPLIT = LNROWS && GEGLU;                               // LayerNorm sums split over the waves of a row (needs the free strip half)
  static_assert(Elem<Tag>::ES == 2, "16-bit storage types only");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wr = wid >> 2, wc = wid & 3;
  const int l31 = lane & 31, hi = lane >> 5;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.a0, p.a0_bytes), rw = make_rsrc(p.w, p.w_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.bias, p.bias_bytes), ro = make_rsrc(p.out, p.out_bytes);
  const int KS = p.k0 >> 6;
  const int nwg = gridDim.x;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / nwg + 1 : 0;
  const int S = my_tiles * KS;
  auto tile_of = [&](int it, int& m0, int& n0) {
    int bid = it * nwg + (int)blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gm = p.group_m, per_group = gm * p.tiles_n;
    const int g = bid / per_group, first = g * gm, rows = min(gm, p.tiles_m - first), rem = bid - g * per_group;
    const int tn = rem / rows, tm = first + (rem - tn * rows);
    m0 = tm * BM; n0 = tn * BN;
  };
  // ---- producer: per-lane source offsets of the 2 pieces of each region for the producer's tile.
  // region row rr (0..127) -> tile row: A-lo (rr/64)*128 + rr%64, A-hi +64 ; W-lo (rr/32)*64 + rr%32, W-hi +32
  int pv[4][2];
  int p_it = 0, p_ks = 0;
  // origins of the producer's last two tiles, by tile parity: the consumer (at most one tile behind) takes its tile's origin from
  // here instead of repeating the integer divisions of tile_of() in its first phase and in its epilogue
  int q_m0[2] = {0, 0}, q_n0[2] = {0, 0};
  auto producer_tile = [&](int it) {
    int m0 = 0, n0 = 0;
    const bool ok = it < my_tiles;
    if (ok) tile_of(it, m0, n0);
    if (it & 1) { q_m0[1] = m0; q_n0[1] = n0; } else { q_m0[0] = m0; q_n0[0] = n0; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = i * 512 + tid, rr = c >> 3, ch = (c & 7) ^ tile_swz<CPR>(rr);
      const int ar = HALF ? rr : (rr >> 6) * 128 + (rr & 63), br = (rr >> 5) * 64 + (rr & 31);
      pv[0][i] = (ok && m0 + ar < p.m) ? (int)(((long)(m0 + ar) * p.lda0 + ch * 8) * ES) : kInv;
      pv[1][i] = (!HALF && ok && m0 + ar + 64 < p.m) ? (int)(((long)(m0 + ar + 64) * p.lda0 + ch * 8) * ES) : kInv;
      pv[2][i] = (ok && n0 + br < p.n) ? (int)(((long)(n0 + br) * p.ldw + ch * 8) * ES) : kInv;
      pv[3][i] = (ok && n0 + br + 32 < p.n) ? (int)(((long)(n0 + br + 32) * p.ldw + ch * 8) * ES) : kInv;
    }
  };
  producer_tile(0);
  // Workgroups that walk one tile fewer than the busiest ones (1960 tiles on 256 CUs: 88 of them) start up to ~3 x 5 us later, in
  // four cohorts: all CUs run in lock step otherwise, so their epilogue store bursts (64 KiB each) and the restart of their DMA
  // streams fall into the same microsecond of every round.  The delay comes out of the idle tile time those workgroups have at
  // the end anyway (K >= 320: a tile takes >= 14 us).  Measured: K = 320 121.1 -> 117.4 us, K = 640 189.8 -> 186.6 us, the step
  // 31.96 -> 31.80 ms (interleaved A/B in one call); twice the delay loses.
  if (KS >= 5 && my_tiles > 0 && my_tiles * nwg < ntiles) {
    const int units = (int)blockIdx.x & 3;
    for (int u = 0; u < units; ++u) __builtin_amdgcn_s_sleep(127);
  }
  // stage region R (compile-time) of the producer's slab into `slot`; region 2 (W-lo) is the last of a slab: advance
  auto stage = [&](int slot, auto reg_tag) {
    constexpr int R = decltype(reg_tag)::value;
    char* base = smem + slot * SLOT + R * REG + wid * 1024;
    const int soff = __builtin_amdgcn_readfirstlane(p_ks) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = pv[R][i];
      if (R < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(base + i * 8192), 16, v, soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(base + i * 8192), 16, v, soff, 0, 0);
    }
    if constexpr (R == (HALF ? 3 : 2)) { if (++p_ks == KS) { p_ks = 0; ++p_it; producer_tile(p_it); } }
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
  float ln_s[FM], ln_q[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) ln_s[i] = ln_q[i] = 0.f;

  const unsigned lds_base = lds_addr(smem);
  // fragment addresses inside a slot: A half i2 -> region i2, rows wr*64 + i*32 + l31 ; W fragment j -> region 2+j, rows wc*32 + l31
  unsigned a_addr[2][4], b_addr[4];
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
  auto read_a = [&](unsigned sbase, int i2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[i][ks] = lds_read16_raw(sbase + i2 * REG + a_addr[i][ks]);
  };
  auto read_b = [&](unsigned sbase, int j) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bf[ks] = lds_read16_raw(sbase + (2 + j) * REG + b_addr[ks]);
  };
  // the 8 MFMAs of quadrant (I2, J)
  int c_ks = 0, c_it = 0;                                    // consumer cursor: slab inside the tile, tile
  // GEGLU instances, first slab of a tile: the first MFMA of every fragment takes the constant 0 as its C operand -- the epilogue
  // does not have to clear 128 accumulator registers per lane and tile (-2..3 % at K = 320 / 640; the plain instances lose 4 %
  // with it and the LayerNorm + plain one spills, so they keep the clearing loop)
  constexpr bool ZERO_BY_MFMA = GEGLU != 0;
  auto mma = [&](auto i2_tag, auto j_tag) {
    constexpr int I2 = decltype(i2_tag)::value, J = decltype(j_tag)::value;
    __builtin_amdgcn_s_setprio(1);
    if (ZERO_BY_MFMA && c_ks == 0) {
      f32x16_t z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[I2 * 2 + i][J] = Cvt<Tag>::mfma32(make_uint4(bf[ks].x, bf[ks].y, bf[ks].z, bf[ks].w),
                                                make_uint4(af[i][ks].x, af[i][ks].y, af[i][ks].z, af[i][ks].w), ks == 0 ? z : acc[I2 * 2 + i][J]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[I2 * 2 + i][J] = Cvt<Tag>::mfma32(make_uint4(bf[ks].x, bf[ks].y, bf[ks].z, bf[ks].w),
                                                make_uint4(af[i][ks].x, af[i][ks].y, af[i][ks].z, af[i][ks].w), acc[I2 * 2 + i][J]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // LayerNorm sums of the A half in the fragment registers (each element of A passes through exactly one p0 / p2 per slab).  They
  // run in the READ interval of the FOLLOWING phase (p1 / p3: only W fragments to read, the A half still in registers), under
  // the latency of those reads: that is when the other group's wave on this SIMD issues its 8 MFMAs, so the two pipes overlap.
  // Behind the wave's own MFMAs they lengthen its MFMA interval while the other group waits at the barrier (+0.45 us per slab);
  // in the read interval of p0 / p2 they sit on top of the longest read bursts (+0.2 us per slab).  The four waves of a wave row hold the SAME A fragments:
  // with SPLIT each sums only K step ks == wc and the epilogue adds the four partial sums.
  auto stats = [&](auto i2_tag) {
    constexpr int I2 = decltype(i2_tag)::value;
    if constexpr (LNROWS) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (!SPLIT || ks == wc) {
#pragma unroll
          for (int i = 0; i < 2; ++i) ln_stat<Tag>(af[i][ks], ln_s[I2 * 2 + i], ln_q[I2 * 2 + i]);
        }
      __builtin_amdgcn_sched_barrier(0);       // the sums read raw-asm fragment registers: keep them between the wait and the barrier
    }
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  // ---- prologue: slab 0 (slot 0) complete + A-lo, W-hi, A-hi of slab 1 (slot 1), as the steady state would have issued them
  // (W-lo closes a slab: it advances the producer cursor, so it goes last)
  if constexpr (HALF) {
    stage(0, R0{}); stage(0, R2{}); stage(0, R3{});
    stage(1, R0{}); stage(1, R2{});
    pp_wait_vm<4>();
  } else {
  stage(0, R0{}); stage(0, R3{}); stage(0, R1{}); stage(0, R2{});
  stage(1, R0{}); stage(1, R3{}); stage(1, R1{});
  pp_wait_vm<6>();
  }
  bar();
  if (grp == 1) bar();                                       // group 1 runs one barrier behind group 0

  const unsigned strip = lds_base + STRIP_OFF + wid * 4096;

  // ---- epilogue of one tile (both groups together; no barrier inside: every wave works on its own strip)
  int c_m0 = 0, c_n0 = 0;                                    // origin of the consumer's tile (set in p0 of its first slab)
  auto epilogue = [&]() {
    const int m0 = c_m0, n0 = c_n0;
    ++c_it;

    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA results -> VALU / raw ds_write: no hazard recogniser in asm
    // bias of this lane's columns: the DMA put the wave's 64 values (columns wc*64 ..) at the start of the strip.  In the MFMA
    // layout lane (l31, hi) holds columns j*32 + 8g + 4hi + {0..3} of row l31
    float4 b4[FN][4];
    {
      raw_u32x4_t t[FN][4];
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) t[j][g] = lds_read16_raw(strip + (j * 32 + 8 * g + 4 * hi) * 4);
      lds_wait<0>();
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          b4[j][g] = make_float4(__uint_as_float(t[j][g].x), __uint_as_float(t[j][g].y), __uint_as_float(t[j][g].z), __uint_as_float(t[j][g].w));
    }
    if constexpr (SPLIT) {
      // partial sums of this wave (its K steps, its half of each K step) -> upper half of its strip, 32 bytes per lane; after the
      // barrier every wave adds the four partials of its wave row (strips wr*4 + 0..3) for its own half; the halves meet below
      lds_write16_raw(strip + 2048 + lane * 32, ln_s[0], ln_q[0], ln_s[1], ln_q[1]);
      if constexpr (FM == 4) lds_write16_raw(strip + 2048 + lane * 32 + 16, ln_s[FM - 2], ln_q[FM - 2], ln_s[FM - 1], ln_q[FM - 1]);
      lds_wait<0>();
      bar();
      raw_u32x4_t t[4][2];
      const unsigned row_strips = lds_base + STRIP_OFF + wr * 4 * 4096 + 2048 + lane * 32;
#pragma unroll
      for (int c = 0; c < 4; ++c) { t[c][0] = lds_read16_raw(row_strips + c * 4096); t[c][1] = lds_read16_raw(row_strips + c * 4096 + 16); }
      lds_wait<0>();
#pragma unroll
      for (int h = 0; h < FM / 2; ++h) {
        ln_s[2 * h] = (__uint_as_float(t[0][h].x) + __uint_as_float(t[1][h].x)) + (__uint_as_float(t[2][h].x) + __uint_as_float(t[3][h].x));
        ln_q[2 * h] = (__uint_as_float(t[0][h].y) + __uint_as_float(t[1][h].y)) + (__uint_as_float(t[2][h].y) + __uint_as_float(t[3][h].y));
        ln_s[2 * h + 1] = (__uint_as_float(t[0][h].z) + __uint_as_float(t[1][h].z)) + (__uint_as_float(t[2][h].z) + __uint_as_float(t[3][h].z));
        ln_q[2 * h + 1] = (__uint_as_float(t[0][h].w) + __uint_as_float(t[1][h].w)) + (__uint_as_float(t[2][h].w) + __uint_as_float(t[3][h].w));
      }
    }
    float rs[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      rs[i] = 1.0f;
      if constexpr (LNROWS) {
        const float inv_k = 1.0f / (float)p.k0;
        const float sm = (ln_s[i] + __shfl_xor(ln_s[i], 32)) * inv_k, sq = (ln_q[i] + __shfl_xor(ln_q[i], 32)) * inv_k;
        rs[i] = rsqrtf(fmaxf(sq - sm * sm, 0.f) + p.ln_eps);
        ln_s[i] = ln_q[i] = 0.f;
      }
    }
    // (acc * rs + b) * scale as ONE fused multiply-add per value: scale folded into rs and into the bias registers
    const float scale = p.acc_scale;
#pragma unroll
    for (int i = 0; i < FM; ++i) rs[i] *= scale;
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) { b4[j][g].x *= scale; b4[j][g].y *= scale; b4[j][g].z *= scale; b4[j][g].w *= scale; }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int mb = m0 + wr * WTM + i * 32;
      if constexpr (!GEGLU) {
        // strip image: 32 rows x 128 bytes (64 columns of the 16-bit type), 16-byte chunks XOR-swizzled by the row
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 b = b4[j][g];
            const float v0 = fmaf(acc[i][j][g * 4], rs[i], b.x), v1 = fmaf(acc[i][j][g * 4 + 1], rs[i], b.y);
            const float v2 = fmaf(acc[i][j][g * 4 + 2], rs[i], b.z), v3 = fmaf(acc[i][j][g * 4 + 3], rs[i], b.w);
            const int col = j * 32 + 8 * g + 4 * hi;                       // first of 4 columns: byte col*2 inside chunk col/8
            lds_write8_raw(strip + l31 * 128 + (((col >> 3) ^ (l31 & 7)) << 4) + ((col & 7) << 1), pack2<Tag>(v0, v1), pack2<Tag>(v2, v3));
          }
        raw_u32x4_t tq[4];
        const int ch = lane & 7, rr = lane >> 3;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) tq[ps] = lds_read16_raw(strip + (ps * 8 + rr) * 128 + ((ch ^ ((ps * 8 + rr) & 7)) << 4));
        lds_wait<0>();
        const int gn = n0 + wc * WTN + ch * 8;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int gm = mb + ps * 8 + rr;
          __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){tq[ps].x, tq[ps].y, tq[ps].z, tq[ps].w}, ro,
                                                 (gm < p.m && gn < p.n) ? (int)(((long)gm * p.ldo + gn) * ES) : kInv, 0, 0);
        }
      } else {
        // GEGLU: quads g = 0 / 2 of a fragment are values, g = 1 / 3 their gates (packing.pack_geglu): 8 outputs per fragment and
        // lane, 32 output columns per wave row = 64 bytes; strip image 32 rows x 64 bytes, chunks swizzled by (row >> 1) & 3
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const float4 bv = b4[j][2 * tt], bg = b4[j][2 * tt + 1];
            float v[4];
#ifdef TT_GELU_SCALAR
            const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, bgv[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = fmaf(acc[i][j][(2 * tt) * 4 + e], rs[i], bvv[e]) * gelu_erf_f(fmaf(acc[i][j][(2 * tt + 1) * 4 + e], rs[i], bgv[e]));
#else
            // two outputs per instruction (packed fp32): the GEGLU epilogue is VALU-bound (64 outputs per lane and tile)
            {
              const f32pk_t r2 = {rs[i], rs[i]};
              const f32pk_t g01 = (f32pk_t){acc[i][j][(2 * tt + 1) * 4], acc[i][j][(2 * tt + 1) * 4 + 1]} * r2 + (f32pk_t){bg.x, bg.y};
              const f32pk_t g23 = (f32pk_t){acc[i][j][(2 * tt + 1) * 4 + 2], acc[i][j][(2 * tt + 1) * 4 + 3]} * r2 + (f32pk_t){bg.z, bg.w};
#ifdef TT_GELU_ERF_AS          // A/B build only (make variant): the Abramowitz-Stegun erf form of rounds 3-4
#define TT_PP_GELU gelu_erf_pk
#elif defined(TT_GELU_SIG)    // A/B build only: the sigmoid form of round 5 for every storage type
#define TT_PP_GELU gelu_sig_pk
#else                          // bf16 storage: the transcendental-free polynomial (its error sits below bf16's output rounding); fp16: the sigmoid form
#define TT_PP_GELU(x) (std::is_same<Tag, bf16_tag>::value ? gelu_poly_pk(x) : gelu_sig_pk(x))
#endif
              const f32pk_t v01 = ((f32pk_t){acc[i][j][(2 * tt) * 4], acc[i][j][(2 * tt) * 4 + 1]} * r2 + (f32pk_t){bv.x, bv.y}) * TT_PP_GELU(g01);
              const f32pk_t v23 = ((f32pk_t){acc[i][j][(2 * tt) * 4 + 2], acc[i][j][(2 * tt) * 4 + 3]} * r2 + (f32pk_t){bv.z, bv.w}) * TT_PP_GELU(g23);
              v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y;
            }
#endif
            const int col = j * 16 + tt * 8 + 4 * hi;                      // output column inside the wave's 32
            lds_write8_raw(strip + l31 * 64 + (((col >> 3) ^ ((l31 >> 1) & 3)) << 4) + ((col & 7) << 1), pack2<Tag>(v[0], v[1]), pack2<Tag>(v[2], v[3]));
          }
        raw_u32x4_t tq[2];
        const int ch = lane & 3, rr = lane >> 2;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) tq[ps] = lds_read16_raw(strip + (ps * 16 + rr) * 64 + ((ch ^ (((ps * 16 + rr) >> 1) & 3)) << 4));
        lds_wait<0>();
        const int oc = ((n0 + wc * WTN) >> 1) + ch * 8;                    // output column
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int gm = mb + ps * 16 + rr;
          __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){tq[ps].x, tq[ps].y, tq[ps].z, tq[ps].w}, ro,
                                                 (gm < p.m && oc * 2 < p.n) ? (int)(((long)gm * p.ldo + oc) * ES) : kInv, 0, TT_PP_GEGLU_STORE_AUX);
        }
      }
      if constexpr (!ZERO_BY_MFMA) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    }
  };

  for (int s = 0; s < S; ++s) {
    const int slot = s & 1;
    const unsigned sb = lds_base + slot * SLOT;
    if constexpr (HALF) {
      // p0: A-lo + W-lo ; DMA: W-hi of slab s+1 (other slot; read in p1 of slab s-1) closes that slab
      read_a(sb, 0); read_b(sb, 0);
      if (c_ks == 0) {
        const int m0 = (c_it & 1) ? q_m0[1] : q_m0[0], n0 = (c_it & 1) ? q_n0[1] : q_n0[0];
        c_m0 = m0; c_n0 = n0;
        const int gn = n0 + wc * 64 + lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(smem + STRIP_OFF + wid * 4096), 4,
                                                 gn < p.n ? gn * 4 : kInv, 0, 0, 0);
      }
      stage(slot ^ 1, R3{});
      lds_wait<0>();
      bar();
      mma(I0{}, I0{});
      bar();
      // p1: W-hi ; DMA: A-lo and W-lo of slab s+2 (this slot, read in p0)
      read_b(sb, 1);
      stage(slot, R0{}); stage(slot, R2{});
      stats(I0{});
      pp_wait_vm<4>();
      lds_wait<0>();
      bar();
      mma(I0{}, I1{});
      bar();
    } else {
    // ---- p0 (0,0): A-lo + W-lo ; DMA: [bias of this tile] + W-lo of slab s+1 (the other slot; released in p3 of slab s-1)
    read_a(sb, 0); read_b(sb, 0);
    if (c_ks == 0) {
      const int m0 = (c_it & 1) ? q_m0[1] : q_m0[0], n0 = (c_it & 1) ? q_n0[1] : q_n0[0];
      c_m0 = m0; c_n0 = n0;
      const int gn = n0 + wc * 64 + lane;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(smem + STRIP_OFF + wid * 4096), 4,
                                               gn < p.n ? gn * 4 : kInv, 0, 0, 0);
    }
    stage(slot ^ 1, R2{});
    lds_wait<0>();
    bar();
    mma(I0{}, I0{});
    bar();
    // ---- p1 (0,1): W-hi ; DMA: A-lo of slab s+2 (this slot; released in p0)
    read_b(sb, 1);
    stage(slot, R0{});
    stats(I0{});
    lds_wait<0>();
    bar();
    mma(I0{}, I1{});
    bar();
    // ---- p2 (1,1): A-hi ; DMA: W-hi of slab s+2 (released in p1)
    read_a(sb, 1);
    stage(slot, R3{});
    lds_wait<0>();
    bar();
    mma(I1{}, I1{});
    bar();
    // ---- p3 (1,0): W-lo again ; DMA: A-hi of slab s+2 (released in p2) ; everything up to W-lo of slab s+1 has landed
    // (vmcnt counts the epilogue's stores too, in issue order.  Issuing W-lo before the epilogue and letting the stores stay in
    // flight for one more slab -- vmcnt(7 + stores) in a tile's first slab -- measured no gain: the write burst of 256 CUs
    // finishing their tiles together costs ~2 us per tile wherever the wave meets it.)
    read_b(sb, 0);
    stage(slot, R1{});
    stats(I1{});
    pp_wait_vm<6>();
    lds_wait<0>();
    bar();
    mma(I1{}, I0{});
    bar();
    }
    if (++c_ks == KS) {
      c_ks = 0;
      if (grp == 0) bar();                                   // both groups run the epilogue together ...
      epilogue();
      if (grp == 1) bar();                                   // ... and group 1 falls one barrier behind again
    }
  }
}

template <typename Tag, int LNROWS, int GEGLU>
static void launch_pp_inst(GemmP& p, hipStream_t st) {
  constexpr int lds = 2 * 65536 + 8 * 4096;
  static unsigned long long attr_done = 0;
  tt_lds_opt_in((const void*)gemm_pp_kernel<Tag, LNROWS, GEGLU>, lds, &attr_done);
#ifdef TT_PP_HALF_PROBE
  p.tiles_m = ceil_div(p.m, 128);
#else
  p.tiles_m = ceil_div(p.m, 256);
#endif
  p.tiles_n = ceil_div(p.n, 256);
  int gm = 1;                                               // group height of the tile order: one XCD's 32 resident tiles ~ gm x 32/gm
  if (p.tiles_n > 8) while (gm * 2 * gm * 2 <= 32 && gm * 2 <= p.tiles_m) gm *= 2;
  p.group_m = p.group_m_override > 0 ? (p.group_m_override > p.tiles_m ? p.tiles_m : p.group_m_override) : gm;
  const int ntiles = p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((gemm_pp_kernel<Tag, LNROWS, GEGLU>), dim3(ntiles < 256 ? ntiles : 256), dim3(512), lds, st, p);
}
template <typename Tag>
static void launch_pp_tag(GemmP& p, hipStream_t st) {
  if (p.ln_fold) { if (p.geglu) launch_pp_inst<Tag, 1, 1>(p, st); else launch_pp_inst<Tag, 1, 0>(p, st); }
  else { if (p.geglu) launch_pp_inst<Tag, 0, 1>(p, st); else launch_pp_inst<Tag, 0, 0>(p, st); }
}
void launch_pp_bf16(GemmP& p, hipStream_t st) { launch_pp_tag<bf16_tag>(p, st); }
void launch_pp_f16(GemmP& p, hipStream_t st) { launch_pp_tag<f16_tag>(p, st); }

}  // namespace ttg
