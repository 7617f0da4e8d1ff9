// r3m_amd — persistent, warp-specialised GEMM on bf16 operands for gfx950 (MI355X): the forward / dgrad launches of the bf16 plans
// (BASELINE configs[2], [4]) whose OUTPUT rows are dense or parity-strided — every 1x1 convolution, strided 3x3 / 1x1 forward
// launches, the parity classes of stride-2 dgrads. out[M x Nc] = sum_t A[pix(m) + tap_t][K] * B[Nc][t][K]^T, bf16 in HBM / LDS,
// fp32 accumulation on v_mfma_f32_32x32x16_bf16, one rounding to bf16. Reference call site: the torchvision convolutions reached
// from /root/reference/r3m/models/models_r3m.py:99 (the reference is fp32-only; precision="bf16" is this build's counterpart of
// torch.autocast around that call).
//
// Why (round 5). The per-tile bf16 kernels of conv_bf16.hip sit on NEITHER roof (0.19-0.41 of HBM, 0.25-0.43 of the MFMA
// peak): with a matrix pipe 16x faster than fp32 a K step of 64 is 512 matrix cycles per wave while an L2 / HBM round trip is
// 1-2 thousand, the two-stage ring keeps ONE step in flight, and every tile pays a cold prologue and an epilogue nobody overlaps.
// Here:
//   * PERSISTENT blocks, one per CU, walk tiles taken from per-XCD queues (as conv_pw.hip); the LDS ring of NS stages (K = 64 per
//     stage, 32-40 KB) is carried ACROSS tiles: the DMA cursor runs NS - 1 K steps ahead of the MFMA cursor whatever tile those
//     steps belong to, so 64-80 KB per CU are always in flight and no tile starts cold;
//   * WARP SPECIALISATION: waves 0-3 ("compute") issue the LDS DMA and the MFMAs and never touch global memory otherwise; waves
//     4-7 ("store") own the epilogue's global traffic. gfx9 has ONE counter (vmcnt) for loads, LDS DMA and stores, and stores may
//     retire out of order with loads — a wave that stores cannot use a counted wait for its DMA (conv_pw.hip documents the
//     intermittent failure). With the roles split, a compute wave's vmcnt only ever counts its own DMA pieces, which retire in
//     order: `s_waitcnt vmcnt((NS - 2) pieces)` is exact, and the stores of tile i fly during tile i + 1 with nobody waiting for
//     them. The hand-off is an LDS out-buffer: at the end of a tile the compute waves take the BatchNorm statistics in-lane from
//     the accumulators (32x32 layout: a lane owns a column), round to bf16, and write row PAIRS packed in dwords
//     ([row / 2][column][row & 1]: one v_cvt_pk + one ds_write_b32 per two results, 128 bytes per wave instruction, conflict-free);
//     after the K loop's next barrier the store waves read it linearly (ds_read_b128), swap halves with the neighbouring lane
//     (DPP, no LDS) so that every lane holds 8 consecutive columns of one row, and issue 16-byte buffer stores whose row offset is
//     a scalar. One s_barrier per K step synchronises all eight waves; the store waves spend the rest of the step parked.
//   * the tile tickets are drawn by a store-wave lane two tiles ahead (its vmcnt waits cost nothing), handed over through LDS.
#include "common.h"
#include "conv_dev.h"

namespace r3m {

typedef unsigned p16_u32x4 __attribute__((ext_vector_type(4)));
typedef float p16_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 p16_bf16x2 __attribute__((ext_vector_type(2)));

constexpr int P16_RSRC_FLAGS = 0x00020000;   // raw buffer, 32-bit offsets, out-of-range lanes read 0 / store nothing

// all LDS traffic of this wave has completed, then the workgroup barrier. (NOT __syncthreads(): its fences drain vmcnt as well,
// which is exactly what the counted DMA waits must not do.) The "memory" clobber keeps the compiler from moving LDS accesses across.
__device__ __forceinline__ void p16_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte store through a descriptor; store + wait states in ONE asm statement (conv_pw.hip, HAZARD: a vector write to the data
// registers in the slot after a wide store with a scalar soffset corrupts lanes)
__device__ __forceinline__ void p16_st4(void* base, int bytes, unsigned voff, int soff, p16_u32x4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (base / bytes / soff are wave-uniform by construction; said explicitly, or the descriptor may be allocated to vector registers)
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  const p16_u32x4 rsrc = {lo, hi, (unsigned)__builtin_amdgcn_readfirstlane(bytes), (unsigned)P16_RSRC_FLAGS};
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
}
__device__ __forceinline__ p16_u32x4 p16_ld4(const void* base, int bytes, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, P16_RSRC_FLAGS), voff, soff, 0);
#else
  return p16_u32x4{0u, 0u, 0u, 0u};
#endif
}
__device__ __forceinline__ unsigned p16_pack(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(p16_f32x2{lo, hi}, p16_bf16x2));
}
__device__ __forceinline__ unsigned p16_swap1(unsigned v) {   // the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2])
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
#else
  return v;
#endif
}
__device__ __forceinline__ const char* p16_uniform_ptr(const char* q) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// BM x BN block tile; EIGHT compute waves WM x WN of 64 x 32 each (two per SIMD: a lone wave issues at most one instruction per four
// cycles, and the first build of this kernel — four compute waves of 64 x 64 — measured issue-bound: with stores, hand-over, DMA and
// MFMAs all switched off its loop skeleton alone took half of the launch, profiles/r05_pw16_probe_v1.txt) + four store waves.
// NS ring stages; NOB out-buffers (2 when a tile may be a single K step: the dump of tile j then runs beside the store waves' reads of
// tile j - 1).
// EPI: 0, EPI_STATS, EPI_ACCUM, EPI_MASKED_ADD (1-bit mask): the read-modify-write forms hand the accumulators over in fp32 (the sum
// is rounded once), their operands are requested by the store waves a whole tile ahead.
// GATHER / OSTR: as conv_pw.hip (A rows are pixels selected per tap / output rows are strided pixels).
// Tile order: a block owns whole ROW PANELS and walks their gridN column tiles back to back — the A rows are read from HBM by one CU
// (and re-read from its own XCD's L2 for the other column tiles), and the BM x Nc block of the result is written by one CU within a
// few tiles, so the L2 evicts whole rows. Panels: static id0 + q W, or (engine launches) ticket k of per-XCD queue xq = panel 8 k + xq.
template <int BM, int BN, int WM, int WN, int NS, int NOB, int EPI, bool GATHER, bool OSTR>
__global__ __launch_bounds__(768, 3) void pw16_gemm_kernel(const GatherGemmParams p, const int gridM, const int gridN) {
  static_assert(WM * WN == 8 && BM / WM == 64 && BN / WN == 32, "eight compute waves of 64 x 32");
  static_assert(NS >= 2 && NS <= 3 && (NOB == 1 || NOB == 2), "ring of 2 or 3 stages, one or two out-buffers");
  constexpr bool RMW = (EPI & (EPI_ACCUM | EPI_MASKED_ADD)) != 0;
  constexpr bool MADD = (EPI & EPI_MASKED_ADD) != 0, ACC = (EPI & EPI_ACCUM) != 0;
  static_assert(!(MADD && ACC), "one added tensor");
  static_assert(!OSTR || (GATHER && !RMW), "strided output rows: gather form, plain / statistics epilogues");
  constexpr int NCW = 8;                                // compute waves
  constexpr int STAGE = (BM + BN) * 128;                // bytes per ring stage: {A[BM][64], B[BN][64]} bf16, 128-byte rows
  constexpr int AJ = BM / 64, BJ = BN / 64, NP = AJ + BJ;   // DMA instructions (8 rows each) per compute wave and stage
  constexpr int OUT_B = BM * BN * (RMW ? 4 : 2);        // bytes of one out-buffer
  constexpr int RED_F = WM * 2 * BN;                    // floats of one statistics scratch
  static_assert(NP * (NS - 2) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const outb = smem + NS * STAGE;
  float* const red = reinterpret_cast<float*>(outb + NOB * OUT_B);              // [2][WM][2][BN]
  __shared__ int tk[4];                                 // tickets q, q + 1, q + 2 of this block (slot q & 3). Plain LDS accesses: every reader
                                                        // sits behind a p16_bar() (a compiler barrier); a volatile pointer would be read with a
                                                        // FLAT load, whose wait drains vmcnt — the DMA ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = gridDim.x;
  const int K = p.Ci, Nc = p.Nc, Kb = K * 2;
  const int kpt = K >> 6;                               // K steps per tap
  const int nsteps = (GATHER ? p.ntaps : 1) * kpt;      // K steps per tile
  const bool dyn = p.tile_ctr != nullptr;
  const int xq = blockIdx.x & 7;
  const int id0 = xcd_remap(blockIdx.x, W);
  // tile cursor: (ticket index q, row panel mt, column tile nt); advancing costs an add and a compare unless the panel changes
  struct Cur { int q, mt, nt; bool has; };
  auto first_tile = [&]() __attribute__((always_inline)) -> Cur {
    Cur c;
    c.q = 0;
    c.nt = 0;
    c.mt = dyn ? __builtin_amdgcn_readfirstlane(tk[0]) * 8 + xq : id0;
    c.has = c.mt < gridM;
    return c;
  };
  auto next_tile = [&](Cur& c) __attribute__((always_inline)) {
    if (++c.nt < gridN) return;
    c.nt = 0;
    ++c.q;
    c.mt = dyn ? __builtin_amdgcn_readfirstlane(tk[c.q & 3]) * 8 + xq : c.mt + W;
    c.has = c.mt < gridM;
  };

  unsigned* const ctr = dyn ? p.tile_ctr + xq : nullptr;
  unsigned pending = 0u;                                // (first store wave, lane 0) the ticket requested one panel ago
  if (dyn && tid == NCW * 64) {
    tk[0] = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk[1] = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk[2] = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pending = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ticket 3, collected at the start of panel 1
  }
  p16_bar();

  Cur cc = first_tile();                                // MFMA cursor
  int cs = 0, cj = 0;                                   // K step in the tile; tile sequence number (parity picks the out-buffer / scratch)

  if (wave_s < NCW) {
    // =============================================== compute waves ===============================================
    const int wm = wave_s / WN, wn = wave_s % WN;
    const int lrow = lane & 31, lh = lane >> 5;
    const int srow = lane >> 3, pslot = lane & 7;
    const char* const Ab = reinterpret_cast<const char*>(p.A);
    const char* const Bb = reinterpret_cast<const char*>(p.B);
    const int KbB = (GATHER ? p.T : 1) * Kb;            // bytes of one weight row: [tap][Ci]
    // ---- DMA: wave w stages rows [w BM/8, +BM/8) of A and [w BN/8, +BN/8) of B, 8 rows (1 KiB) per instruction; the 16-byte slot a
    // lane fetches is XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128 fragments)
    unsigned voffA[AJ], voffB[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int r = wave_s * (BM / NCW) + j * 8 + srow;
      voffA[j] = (unsigned)(r * Kb + ((pslot ^ ((r >> 1) & 7)) << 4));          // GATHER: recomputed per tile and tap
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int r = wave_s * (BN / NCW) + j * 8 + srow;
      voffB[j] = (unsigned)(r * KbB + ((pslot ^ ((r >> 1) & 7)) << 4));
    }
    const char* dA = Ab;
    const char* dB = Bb;
    int dAbytes = 0, dsA = 0, dsB = 0, tap_soffB = 0;
    unsigned poff[GATHER ? AJ : 1], iyx[GATHER ? AJ : 1];
    auto aim_tile = [&](int tmt_, int tnt_) __attribute__((always_inline)) {
      const int tmt = __builtin_amdgcn_readfirstlane(tmt_), tnt = __builtin_amdgcn_readfirstlane(tnt_);
      dB = Bb + (long long)tnt * BN * KbB;
      if constexpr (!GATHER) {
        dA = Ab + (long long)tmt * BM * Kb;
        dAbytes = min(BM, p.M - tmt * BM) * Kb;
      } else {
        const int hw = p.Hg * p.Wg;
        const int m0t = tmt * BM;
        const int nf = __builtin_amdgcn_readfirstlane(m0t / hw);   // first frame of the tile: 32-bit offsets are relative to it
        const long long frame = (long long)p.Hi * p.Wi * Kb;
        dA = p16_uniform_ptr(Ab + nf * frame);
        const long long rest = (long long)(p.N - nf) * frame;
        dAbytes = __builtin_amdgcn_readfirstlane(rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
        const int r0 = wave_s * (BM / NCW) + srow;
        int m = m0t + r0;
        int n = m / hw;
        int rem = m - n * hw;
        int gy = rem / p.Wg;
        int gx = rem - gy * p.Wg;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const int r = r0 + j * 8;
          const int iy0 = gy * p.is, ix0 = gx * p.is;
          poff[j] = (unsigned)((((n - nf) * p.Hi + iy0) * p.Wi + ix0) * Kb + ((pslot ^ ((r >> 1) & 7)) << 4));
          iyx[j] = m < p.M ? (unsigned)((iy0 << 16) | ix0) : 0x40004000u;
          m += 8;                                           // the lane's next row is 8 GEMM rows further: branch-free carries
          gx += 8;                                          // (Wg >= 4: at most two row wraps; Hg >= 2: at most two frame wraps — pw16_form)
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const bool c = gx >= p.Wg;
            gx -= c ? p.Wg : 0;
            gy += c ? 1 : 0;
          }
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const bool c = gy >= p.Hg;
            gy -= c ? p.Hg : 0;
            n += c ? 1 : 0;
          }
        }
      }
    };
    auto set_tap = [&](int t) __attribute__((always_inline)) {
      if constexpr (GATHER) {
        const int pack = __builtin_amdgcn_readfirstlane(p.tap[t]);
        const int dy = (pack << 24) >> 24, dx = (pack << 16) >> 24, wt = pack >> 16;
        const int delta = (dy * p.Wi + dx) * Kb;
        tap_soffB = wt * Kb;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const unsigned iy = (iyx[j] >> 16) + (unsigned)dy, ix = (iyx[j] & 0xFFFFu) + (unsigned)dx;
          voffA[j] = (iy < (unsigned)p.Hi && ix < (unsigned)p.Wi) ? poff[j] + (unsigned)delta : BUF_OOB;
        }
      }
    };
    auto aim_step = [&](int chunk) __attribute__((always_inline)) {
      dsA = __builtin_amdgcn_readfirstlane(chunk * 128);
      dsB = __builtin_amdgcn_readfirstlane(tap_soffB + chunk * 128);
    };
    auto issue_all = [&](int stage) __attribute__((always_inline)) {
      unsigned char* const st = smem + stage * STAGE;
      static_for<NP>([&](auto pc_c) __attribute__((always_inline)) {
        constexpr int pc = decltype(pc_c)::value;
        if (R3M_PROBE(p) & (pc < AJ ? 4 : 8)) return;     // timing probes (probe builds; wrong results): 4 no A DMA, 8 no B DMA
        if constexpr (pc < AJ)
          buf_dma16_uniform(dA, dAbytes, st + (wave_s * (BM / NCW) + pc * 8) * 128, voffA[pc], dsA);
        else
          buf_dma16_uniform(dB, BN * KbB, st + BM * 128 + (wave_s * (BN / NCW) + (pc - AJ) * 8) * 128, voffB[pc - AJ], dsB);
      });
    };

    // ---- DMA cursor: (tile, tap, chunk); NS - 1 K steps ahead of the MFMA cursor
    Cur ic = cc;
    int itap = 0, ichunk = 0;
    if (ic.has) {
      aim_tile(ic.mt, ic.nt);
      set_tap(0);
      aim_step(0);
    }
    auto advance_issue = [&]() __attribute__((always_inline)) {
      if (++ichunk == kpt) {
        ichunk = 0;
        if (!GATHER || ++itap == p.ntaps) {
          itap = 0;
          next_tile(ic);
          if (ic.has) aim_tile(ic.mt, ic.nt);
        }
        if (ic.has) set_tap(itap);
      }
      if (ic.has) aim_step(ichunk);
    };
    int ahead = 0;                                        // K steps issued and not yet multiplied; step b lives in ring slot b % NS
    int istage = 0, cstage = 0;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
      if (ic.has) {
        issue_all(istage);
        ++ahead;
        istage = istage + 1 == NS ? 0 : istage + 1;
        advance_issue();
      }
    }

    // ---- fragments: lane half h reads k = 16 g + 8 h .. + 7 of group g (one ds_read_b128 = the 32x32x16 operand of a row)
    unsigned fa[4], fb[4];                                // LDS byte offsets inside a stage
    {
      const int xr = (lrow >> 1) & 7;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int go = ((2 * g + lh) ^ xr) * 16;
        fa[g] = (unsigned)((wm * 64 + lrow) * 128 + go);
        fb[g] = (unsigned)(BM * 128 + (wn * 32 + lrow) * 128 + go);
      }
    }
    f32x16 acc[2];
    bf16x8 fra[4][2], frb[4];                             // the fragments of one K step: 12 x 16 bytes per lane
    auto frag_load_all = [&](const unsigned char* st) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fra[g][t] = *reinterpret_cast<const bf16x8*>(st + fa[g] + t * 32 * 128);
        frb[g] = *reinterpret_cast<const bf16x8*>(st + fb[g]);
      }
    };
    auto mfma_all = [&](auto first_c) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          if (FIRST && g == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[0][tm], frb[0], z, 0, 0, 0);
          } else {
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fra[g][tm], frb[g], acc[tm], 0, 0, 0);
          }
        }
    };

    // ---- end of a tile: BatchNorm statistics from the accumulators, then the accumulators into the out-buffer.
    // acc[tm][r] is tile row wm 64 + tm 32 + 8 (r >> 2) + 4 lh + (r & 3), tile column wn 32 + lrow.
    auto dump = [&](int j) __attribute__((always_inline)) {
      if constexpr ((EPI & EPI_STATS) != 0) {
        float* const rd = red + (j & 1) * RED_F;          // same summation order as gg_stats (conv_dev.h)
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[tm][r];
            s += v;
            ss = fmaf(v, v, ss);
          }
        s += __shfl_xor(s, 32);
        ss += __shfl_xor(ss, 32);
        if (lane < 32) {
          const int c = wn * 32 + lane;
          rd[(wm * 2 + 0) * BN + c] = s;
          rd[(wm * 2 + 1) * BN + c] = ss;
        }
      }
      unsigned char* const ob = outb + (NOB == 2 ? (j & 1) * OUT_B : 0);
      if constexpr (!RMW) {
        // [row][column] bf16 (ds_write_b16 / _d16_hi of the packed conversions): the store waves read whole rows and need no shuffles.
        // (The first build packed row PAIRS in dwords — half the LDS writes here, but four v_perm, eight v_cndmask and two DPP moves
        // per 16-byte store in the store waves, whose work sits between two barriers of ONE K step: profiles/r05_pw16_probe_v2.txt.)
        bf16_t* const w0 = reinterpret_cast<bf16_t*>(ob) + (wm * 64 + 4 * lh) * BN + wn * 32 + lrow;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) w0[(tm * 32 + 8 * (r >> 2) + (r & 3)) * BN] = (bf16_t)acc[tm][r];
      } else {
        // [row][column] fp32
        float* const w0 = reinterpret_cast<float*>(ob) + (wm * 64 + 4 * lh) * BN + wn * 32 + lrow;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) w0[(tm * 32 + 8 * (r >> 2) + (r & 3)) * BN] = acc[tm][r];
      }
    };

    while (cc.has) {
      // my pieces of the step about to be multiplied have landed; the pieces of the (up to NS - 2) steps issued after it may still fly
      if (NS == 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      p16_bar();                                          // everyone's have; and everyone is done with the ring slot of the previous step
      // the step's fragment reads go out FIRST: their LDS latency elapses under the DMA issue and the cursor arithmetic (all waves of
      // the block run these phases in lock-step, so nothing else would cover it)
      const bool mm = !(R3M_PROBE(p) & 16);               // probe 16: no fragment reads, no MFMAs
      if (mm) frag_load_all(smem + cstage * STAGE);
      if (ic.has) {
        issue_all(istage);
        ++ahead;
        istage = istage + 1 == NS ? 0 : istage + 1;
        advance_issue();
      }
      if (!mm) {
        if (cs == 0) acc[0] = acc[1] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      } else if (cs == 0) mfma_all(std::true_type{});
      else mfma_all(std::false_type{});
      --ahead;
      cstage = cstage + 1 == NS ? 0 : cstage + 1;
      if (++cs == nsteps) {
        if (!(R3M_PROBE(p) & 2)) dump(cj);                // probe 2: no hand-over
        cs = 0;
        ++cj;
        next_tile(cc);
      }
    }
    p16_bar();                                            // the last tile's dump is visible to the store waves
  } else {
    // =============================================== store waves ================================================
    const int v = wave_s - NCW;
    char* const outp = reinterpret_cast<char*>(p.out);

    // store layout: a lane owns 8 consecutive columns of one row (16 bytes of bf16); LPR_R lanes per row, RPI_R rows per instruction
    constexpr int LPR_R = BN / 8, RPI_R = 64 / LPR_R, NIT_R = BM / 4 / RPI_R;
    // read-modify-write operands of the tile being computed, requested at its first K step (same layout)
    const int rrow_s = v * (BM / 4) + lane / LPR_R, rcol_s = 8 * (lane % LPR_R);
    p16_u32x4 pg[RMW ? NIT_R : 1];
    unsigned pgm[MADD ? NIT_R : 1];
    auto prefetch = [&](int mt, int nt) __attribute__((always_inline)) {
      if constexpr (RMW) {
        const int m0 = mt * BM, n0 = nt * BN;
        const int rows_valid = min(BM, p.M - m0);
        const long long eo0 = (long long)m0 * Nc + n0;
        const int obytes = ((rows_valid - 1) * Nc + BN) * 2;
        const char* gb = reinterpret_cast<const char*>(MADD ? p.add0 : p.out) + eo0 * 2;
        const unsigned vo = (unsigned)((rrow_s * Nc + rcol_s) * 2);
#pragma unroll
        for (int it = 0; it < NIT_R; ++it) pg[it] = p16_ld4(gb, obytes, vo, it * RPI_R * Nc * 2);
        if constexpr (MADD) {
          // 8 mask bits of (row, 8 columns): byte (row Nc + col) / 8 of the bit tensor
          const unsigned char* bb = reinterpret_cast<const unsigned char*>(p.addbits) + (eo0 >> 3);
          const int bbytes = (((rows_valid - 1) * Nc + BN) >> 3);
          const unsigned vb = (unsigned)((rrow_s * Nc + rcol_s) >> 3);
#pragma unroll
          for (int it = 0; it < NIT_R; ++it) {
#if defined(__HIP_DEVICE_COMPILE__)
            pgm[it] = __builtin_amdgcn_raw_buffer_load_b8(__builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(bb), 0, bbytes, P16_RSRC_FLAGS), vb,
                                                          (it * RPI_R * Nc) >> 3, 0);
#endif
          }
        }
      }
    };

    auto store_tile = [&](int mt, int nt, int j) __attribute__((always_inline)) {
      const int m0 = mt * BM, n0 = nt * BN;
      if constexpr ((EPI & EPI_STATS) != 0) {
        const float* const rd = red + (j & 1) * RED_F;
        const int t = tid - NCW * 64;
        if (t < BN) {
          float s = 0.f, ss = 0.f;
#pragma unroll
          for (int w = 0; w < WM; ++w) {
            s += rd[(w * 2 + 0) * BN + t];
            ss += rd[(w * 2 + 1) * BN + t];
          }
          p.stats[((long long)mt * 2 + 0) * Nc + n0 + t] = s;
          p.stats[((long long)mt * 2 + 1) * Nc + n0 + t] = ss;
        }
      }
      const unsigned char* const ob = outb + (NOB == 2 ? (j & 1) * OUT_B : 0);
      // destination: descriptor from the tile origin (dense rows) / from the tile's first frame (strided rows)
      char* obase;
      int obytes;
      unsigned vo = 0u;
      int nf = 0, hw = 1;
      if constexpr (!OSTR) {
        const int rows_valid = min(BM, p.M - m0);
        obase = outp + ((long long)m0 * Nc + n0) * 2;
        obytes = ((rows_valid - 1) * Nc + BN) * 2;        // a lane's offset is inside iff its row is < rows_valid
      } else {
        hw = p.Hg * p.Wg;
        nf = __builtin_amdgcn_readfirstlane(m0 / hw);
        const long long oframe = (long long)p.Ho * p.Wo * Nc;
        obase = outp + (nf * oframe + n0) * 2;
        const long long rest = ((long long)(p.N - nf) * oframe - n0) * 2;
        obytes = __builtin_amdgcn_readfirstlane(rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
      }
      auto row_off = [&](int trow, int tcol) __attribute__((always_inline)) -> unsigned {   // OSTR: byte offset of (tile row, tile column)
        const int m = m0 + trow;
        const int n = m / hw;
        const int rem = m - n * hw;
        const int gy = rem / p.Wg, gx = rem - gy * p.Wg;
        return m < p.M ? (unsigned)(((((n - nf) * p.Ho + gy * p.os + p.ooy) * p.Wo + gx * p.os + p.oox) * Nc + tcol) * 2) : BUF_OOB;
      };
      if constexpr (!RMW) {
        if constexpr (!OSTR) vo = (unsigned)((rrow_s * Nc + rcol_s) * 2);
        const unsigned char* src = ob + (rrow_s * BN + rcol_s) * 2;
        p16_u32x4 d[NIT_R];
#pragma unroll
        for (int it = 0; it < NIT_R; ++it) d[it] = *reinterpret_cast<const p16_u32x4*>(src + it * RPI_R * BN * 2);   // all reads in flight at once
#pragma unroll
        for (int it = 0; it < NIT_R; ++it) {
          if (R3M_PROBE(p) & 1) continue;                 // probe 1: no result stores
          if constexpr (!OSTR) p16_st4(obase, obytes, vo, it * RPI_R * Nc * 2, d[it]);
          else p16_st4(obase, obytes, row_off(rrow_s + it * RPI_R, rcol_s), 0, d[it]);
        }
      } else {
        if constexpr (!OSTR) vo = (unsigned)((rrow_s * Nc + rcol_s) * 2);
        const float* src = reinterpret_cast<const float*>(ob) + rrow_s * BN + rcol_s;
#pragma unroll
        for (int it = 0; it < NIT_R; ++it) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(src + it * RPI_R * BN);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(src + it * RPI_R * BN + 4);
          float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          const bf16x8 g = __builtin_bit_cast(bf16x8, pg[it]);
          unsigned nb = 255u;
          if constexpr (MADD) nb = pgm[it];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += ((nb >> e) & 1u) ? (float)g[e] : 0.f;
          const p16_u32x4 o = {p16_pack(x[0], x[1]), p16_pack(x[2], x[3]), p16_pack(x[4], x[5]), p16_pack(x[6], x[7])};
          if constexpr (!OSTR) p16_st4(obase, obytes, vo, it * RPI_R * Nc * 2, o);
          else p16_st4(obase, obytes, row_off(rrow_s + it * RPI_R, rcol_s), 0, o);
        }
      }
    };

    bool pend = false;
    int pmt = 0, pnt = 0;
    while (cc.has) {
      if (cs == 0 && dyn && cc.nt == 0 && cc.q >= 1 && tid == NCW * 64) {
        tk[(cc.q + 2) & 3] = (int)pending;                // ticket q + 2, requested one panel ago
        pending = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      p16_bar();
      if (cs == 0) {
        if (pend) store_tile(pmt, pnt, cj - 1);
        if constexpr (RMW && !OSTR) prefetch(cc.mt, cc.nt);
      }
      if (++cs == nsteps) {
        pend = true;
        pmt = cc.mt;
        pnt = cc.nt;
        cs = 0;
        ++cj;
        next_tile(cc);
      }
    }
    p16_bar();
    if (pend) store_tile(pmt, pnt, cj - 1);
  }
}

// ---- launcher ----------------------------------------------------------------------------------------------------------
static int g_pw16_mode = 1;              // diagnostic switch (r3m_debug_set_pw16): 0 = the per-tile kernels of conv_bf16.hip everywhere
int pw16_set_mode(int on) { const int old = g_pw16_mode; g_pw16_mode = on; return old; }

static int p16_cu_count() {
  static int cus[32] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
  const bool cached = dev < 32;
  if (cached) {
    const int c = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (c > 0) return c;
  }
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (cached) __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
  return n;
}

// 0: not for this kernel; 1: pointwise form (1x1 / stride 1); 2: gather form (dense output rows); 3: gather form with strided
// output rows (a parity class of a stride-2 dgrad). A pure function of the launch parameters (tests/test_dispatch.py).
int pw16_form(const GatherGemmParams& p) {
  if (!g_pw16_mode || p.dtype != DT_BF16) return 0;
  if ((p.Ci & 63) || p.Ci > 4096 || (p.Nc & 63) || p.M < 1) return 0;
  const bool rmw = p.flags == EPI_ACCUM || (p.flags == EPI_MASKED_ADD && p.addbits != nullptr);
  if (p.flags != 0 && p.flags != EPI_STATS && !rmw) return 0;
  const bool dense_out = p.os == 1 && p.ooy == 0 && p.oox == 0 && p.Hg == p.Ho && p.Wg == p.Wo;
  const long long ksteps = (long long)(p.simple_rows ? 1 : p.ntaps) * (p.Ci >> 6);
  if (rmw && ksteps < 2) return 0;                       // one fp32 out-buffer: the dump of a tile must not meet the reads of the previous one
  if (dense_out && p.simple_rows && p.ntaps == 1 && p.T == 1 && p.dy[0] == 0 && p.dx[0] == 0 && p.wt[0] == 0) return 1;
  if (p.simple_rows || p.ntaps < 1 || p.ntaps > MAX_TAPS) return 0;
  // 32-bit offsets: a tile's rows span at most ceil(256 / (Hg Wg)) + 1 frames of the input; one weight tile [128][T][Ci]
  if (p.Hi >= 16384 || p.Wi >= 16384 || p.Hi < 1 || p.Wi < 1 || p.Wg < 4 || p.Hg < 2) return 0;
  const long long frame = (long long)p.Hi * p.Wi * p.Ci * 2;
  const long long span = (256 / ((long long)p.Hg * p.Wg) + 2) * frame;
  if (span >= (long long)BUF_OOB || 128LL * p.T * p.Ci * 2 >= (long long)BUF_OOB) return 0;
  if (!dense_out) {
    if (p.os < 1 || p.ooy < 0 || p.oox < 0 || (p.Hg - 1) * p.os + p.ooy >= p.Ho || (p.Wg - 1) * p.os + p.oox >= p.Wo) return 0;
    if ((256 / ((long long)p.Hg * p.Wg) + 2) * (long long)p.Ho * p.Wo * p.Nc * 2 >= (long long)BUF_OOB) return 0;
    if (p.flags != 0) return 0;
    return 3;
  }
  return 2;
}

template <int BM, int BN, int WM, int WN, int NS, int NOB, int EPI, bool GA, bool OS>
static int launch_pw16_one(const GatherGemmParams& p, int W, int gridM, int gridN, hipStream_t s) {
  constexpr bool RMW = (EPI & (EPI_ACCUM | EPI_MASKED_ADD)) != 0;
  constexpr int LDS = NS * (BM + BN) * 128 + NOB * BM * BN * (RMW ? 4 : 2) + 2 * WM * 2 * BN * 4 + 64;
  static_assert(LDS <= 160 * 1024, "LDS budget of one CU");
  auto kern = pw16_gemm_kernel<BM, BN, WM, WN, NS, NOB, EPI, GA, OS>;
  static DynLdsOptIn oi;
  if (int e = ensure_dyn_lds(oi, reinterpret_cast<const void*>(kern), LDS, "pw16_gemm")) return e;
  hipLaunchKernelGGL(kern, dim3(W), dim3(768), LDS, s, p, gridM, gridN);
  return 0;
}

template <int BM, int BN, int WM, int WN, bool GA, bool OS>
static int launch_pw16_shape(const GatherGemmParams& p_in, hipStream_t s) {
  const int gridM = ceil_div(p_in.M, BM), gridN = p_in.Nc / BN;
  const int slots = p16_cu_count();                                     // one twelve-wave block per CU
  const int W = gridM < slots ? gridM : slots;
  GatherGemmParams p = p_in;
  if (W < 64 || gridM < 64) p.tile_ctr = nullptr;                       // small launches: every queue needs blocks AND panels; static split
  const long long ksteps = (long long)(p.simple_rows ? 1 : p.ntaps) * (p.Ci >> 6);
  if constexpr (OS) {
    if (p.flags == 0) return ksteps >= 2 ? launch_pw16_one<BM, BN, WM, WN, 3, 1, 0, GA, OS>(p, W, gridM, gridN, s)
                                         : launch_pw16_one<BM, BN, WM, WN, 2, 2, 0, GA, OS>(p, W, gridM, gridN, s);
    set_last_error("pw16_gemm: form not built");
    return 1;
  } else {
    switch (p.flags) {
      case 0:
        return ksteps >= 2 ? launch_pw16_one<BM, BN, WM, WN, 3, 1, 0, GA, OS>(p, W, gridM, gridN, s)
                           : launch_pw16_one<BM, BN, WM, WN, 2, 2, 0, GA, OS>(p, W, gridM, gridN, s);
      case EPI_STATS:
        return ksteps >= 2 ? launch_pw16_one<BM, BN, WM, WN, 3, 1, EPI_STATS, GA, OS>(p, W, gridM, gridN, s)
                           : launch_pw16_one<BM, BN, WM, WN, 2, 2, EPI_STATS, GA, OS>(p, W, gridM, gridN, s);
      case EPI_ACCUM: return launch_pw16_one<BM, BN, WM, WN, 2, 1, EPI_ACCUM, GA, OS>(p, W, gridM, gridN, s);
      case EPI_MASKED_ADD: return launch_pw16_one<BM, BN, WM, WN, 2, 1, EPI_MASKED_ADD, GA, OS>(p, W, gridM, gridN, s);
    }
    set_last_error("pw16_gemm: unsupported epilogue flag combination %d", p.flags);
    return 1;
  }
}

int launch_pw16(const GatherGemmParams& p, hipStream_t s) {
  const bool wide = (p.Nc & 127) == 0;
  const int form = pw16_form(p);
  R3M_REQUIRE(form != 0, "pw16_gemm: launch not eligible");
  if (form == 3) return wide ? launch_pw16_shape<128, 128, 2, 4, true, true>(p, s) : launch_pw16_shape<256, 64, 4, 2, true, true>(p, s);
  if (form == 2) return wide ? launch_pw16_shape<128, 128, 2, 4, true, false>(p, s) : launch_pw16_shape<256, 64, 4, 2, true, false>(p, s);
  return wide ? launch_pw16_shape<128, 128, 2, 4, false, false>(p, s) : launch_pw16_shape<256, 64, 4, 2, false, false>(p, s);
}

}  // namespace r3m
