// r3m_amd — persistent pointwise GEMM for gfx950 (MI355X), fp32: the 1x1 / stride-1 convolutions of the bottleneck blocks
// (forward and dgrad), i.e. out[M x Nc] = A[M x K] * B[Nc x K]^T with contiguous rows on all three sides.
//
// Replaces, for these launches, gather_gemm_glds2_kernel / gather_gemm_k16_kernel of conv.hip (same MFMA order, same LDS image:
// the plain-store and BatchNorm-statistics results are bit-identical to theirs). Reference call site: torchvision Bottleneck
// conv1 / conv3 / downsample reached from /root/reference/r3m/models/models_r3m.py:99.
//
// Why a second kernel. Round 3's launch report: the 1x1 launches are 90 ms of the 149 ms the 128-wide class takes per ResNet-50
// step and run at 69-128 TFLOP/s while the 1x1 WEIGHT gradients (tiny outputs, no epilogue) reach 135-142. What the forward /
// dgrad launches pay on top is per-tile: a cold prologue (descriptor set-up, first DMA latency), an epilogue that transposes
// through LDS with ~4 vector instructions per MFMA for K = 64 (on gfx950 the f32 MFMA shares the SIMD's fp32 lanes with the VALU:
// vector work is matrix time), and a block boundary whose stores nobody overlaps. Here:
//   * PERSISTENT blocks (two per CU) walk tiles w, w + W, ...; the two-stage LDS ring is carried ACROSS tiles: the last K step
//     of a tile carries the DMA of the next tile's first step, so no tile starts cold;
//   * operands arrive by `buffer_load_dwordx4 ... lds` through wave-uniform descriptors: per-lane offsets are kernel constants,
//     the K position rides in the instruction's scalar offset, rows past M fall off the descriptor and land zeros — the K loop
//     holds NO vector ALU instruction;
//   * the epilogue takes BatchNorm forward statistics straight from the accumulators (32x32 MFMA result layout: a lane owns one
//     COLUMN, its 16 registers are rows: in-lane sums) and stores through a 2 KB per-wave LDS slab, 8 rows at a time: 16
//     `buffer_store_dwordx4` per wave tile through a descriptor, row offsets scalar, per-lane offsets constant — no address
//     arithmetic, no exec masks. (Round 4 first stored the accumulators directly, 64 `buffer_store_dword` of two 128-byte row
//     segments each and no LDS: tools/micro/storepat.hip shows a CU retires one store INSTRUCTION per 14-19 cycles whatever its
//     width — 3.0 us of a wave's issue time per tile against 1.0 us for the 16 wide stores.)
//   * read-modify-write epilogues (residual-gradient join, EPI_BNRED's y, EPI_ACCUM's old result) PREFETCH their operands into
//     registers during the last two K steps of the tile, a few loads between every four MFMAs, so that the epilogue itself
//     waits for nothing (measured with loads inside the epilogue: every sub-tile paid a full HBM latency and these launches ran
//     5 % SLOWER than in the per-tile kernels whose four blocks per CU hide it); the operands arrive in the STORE layout (one
//     dwordx4 per tensor and store). The 1-bit masks of a wave's 64 x 64 tile are ONE dwordx2 load (lane = row) expanded with
//     ds_bpermute;
//   * the epilogue of tile i is DEFERRED into the first K step of tile i + 1, after that step's data has landed and the
//     following step's DMA is issued: gfx9 has ONE counter for loads and stores, so the `vmcnt(0)` that publishes the next LDS
//     stage also waits for the stores — one K step (64 MFMAs per wave) after their issue instead of immediately.
#include "common.h"
#include "conv_dev.h"

namespace r3m {

constexpr int PW_RSRC_FLAGS = 0x00020000;   // raw buffer, 32-bit offsets, out-of-range lanes read 0 / store nothing

__device__ __forceinline__ float pw_ld(const void* base, int bytes, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                       __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, PW_RSRC_FLAGS), voff, soff, 0));
#else
  return 0.f;
#endif
}
__device__ __forceinline__ unsigned pw_ldu(const void* base, int bytes, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, PW_RSRC_FLAGS), voff, soff, 0);
#else
  return 0u;
#endif
}
typedef unsigned pw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 pw_ld4(const void* base, int bytes, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                       __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, PW_RSRC_FLAGS), voff, soff, 0));
#else
  return f32x4{0.f, 0.f, 0.f, 0.f};
#endif
}
// HAZARD (found the hard way, round 4): a VMEM store of more than 64 bits reads its data registers over several cycles; a vector
// instruction that overwrites them in the very next slot corrupts what some lanes store. The compiler's hazard recognizer inserts
// the wait states only when the store's soffset operand is NOT a register ("this hazard only exists if the instruction is not
// using a register in the soffset field") — on gfx950 it exists with a scalar soffset too: `buffer_store_dwordx4 v[42:45] ... s20
// offen` followed immediately by `v_cndmask_b32 v45, 0, v45` (EPI_BNRED masks the stored value in place) stored a zero in element
// 3 of lanes 12-15 of each 16-lane group — sporadically, only in blocks that share their CU with an earlier block, and only after
// other kernels had run in the process (tests/test_gpu_ops.py, the `bits` cases with more tiles than workers). A separate
// `s_nop` statement after the builtin does not help: the scheduler slides vector instructions between the two. So the store and
// its wait states are ONE inline-asm statement.
#ifndef PW_ST_MOD
#define PW_ST_MOD ""   // cache policy of the result stores (experiment switch: " nt", " sc1", " sc0 sc1")
#endif
__device__ __forceinline__ void pw_st4(void* base, int bytes, unsigned voff, int soff, f32x4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const pw_u32x4 rsrc = {(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, (unsigned)bytes, (unsigned)PW_RSRC_FLAGS};   // stride 0: raw buffer
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" PW_ST_MOD "\n\ts_nop 3" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#endif
}
typedef unsigned pw_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pw_u32x2 pw_ld2(const void* base, int bytes, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_raw_buffer_load_b64(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, PW_RSRC_FLAGS), voff, 0, 0);
#else
  return pw_u32x2{0u, 0u};
#endif
}
__device__ __forceinline__ void pw_st(void* base, int bytes, unsigned voff, int soff, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, PW_RSRC_FLAGS), voff,
                                        soff, 0);
#endif
}

__device__ __forceinline__ const float* pw_uniform_ptr(const float* q) {   // a wave-uniform pointer, in scalar registers
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

// EPI: 0, EPI_STATS, EPI_ACCUM, EPI_MASKED_ADD (1-bit mask only), EPI_BNRED, EPI_BNRED | EPI_MASKED_ADD; inference forward (round 6):
//      EPI_AFFINE, EPI_AFFINE | EPI_RELU, EPI_AFFINE | EPI_ACCUM | EPI_RELU (eval-mode BatchNorm, residual already in `out`, ReLU at the store)
// YBITS (EPI_BNRED): the consumer BatchNorm's ReLU mask comes as bits (block outputs) / is recomputed from y (inner BatchNorms)
// Two shapes: <128, 128, 2, 2> — four waves of 64 x 64, two blocks per CU — for outputs whose width is a multiple of 128, and
// <256, 64, 4, 2> — EIGHT waves of 64 x 32 (TN = 1), one block per CU — for the 64-channel outputs (conv1 / the dgrad of conv3 in
// layer1): the 256-row tile keeps the statistics partial-row geometry of the other 64-wide kernels (gather_gemm_grid_m).
// GATHER: the A rows are pixels of an NHWC tensor selected per tap ((gy is + dy, gx is + dx), zero outside the image) instead of
// rows of a matrix: every convolution whose OUTPUT rows are dense (forward of any k / stride / pad, dgrad of stride-1 layers). The
// row -> pixel decode runs once per tile, a tap switch is ~7 vector instructions per staged row (offset + bounds test -> out-of-range
// offset), everything else is the pointwise kernel.
// OSTR (with GATHER): the OUTPUT rows are strided too — pixel (gy os + ooy, gx os + oox) of an [N, Ho, Wo, Nc] tensor: the parity-class
// launches of a stride-2 dgrad. The row -> pixel decode of the wave's 64 result rows runs once per tile (lane l = row l), the 8 TN
// stores / operand loads take their row offsets from it with one ds_bpermute each, all at the start of the tile.
// BURST: the DMA pieces of a K step leave in one burst before its MFMAs instead of one per eight MFMAs. Spreading hides their issue
// cost under the MFMAs; the burst gives the loads a whole K step to land. Measured per shape (profiles/r04_cluster_dma_ab.txt): the
// burst wins 2-3 % on contracting launches (K >= 2 N: their A stream is the larger side) and loses 2-6 % on expanding ones; see pw_burst.
template <int BM, int BN, int WM, int WN, int EPI, bool YBITS = false, bool GATHER = false, bool OSTR = false, bool BURST = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 4 ? 2 : 1) void pw_gemm_kernel(const GatherGemmParams p, const int tiles, const int gridN) {
  constexpr int NW = WM * WN, TN = BN / WN / 32;        // waves; 32-column MFMA tiles per wave (2 or 1)
  static_assert(BM / WM == 64 && (TN == 1 || TN == 2) && (NW == 4 || NW == 8), "wave tile is 64 rows x 32 TN columns");
  constexpr int STAGE = (BM + BN) * 32;                 // floats per stage: {A[BM][32], B[BN][32]}, 128-byte rows
  constexpr int AJ = BM / (8 * NW), BJ = BN / (8 * NW), NP = AJ + BJ;   // DMA pieces (8 rows each) per wave and stage
  static_assert(AJ >= 1 && BJ >= 1 && (NP <= 8 || BURST), "a wave stages whole DMA instructions, at most one per odd MFMA slot (burst: any number)");
  constexpr int NST = 8 * TN;                           // dwordx4 stores per wave tile (a store covers 64 / (8 TN) rows)
  // BatchNorm statistics rows (EPI_STATS) are per SR result rows whatever the tile (gather_gemm_grid_m): R of them per tile
  constexpr int SR = BN >= 128 ? 128 : 256, R = BM / SR;
  static_assert(R >= 1 && WM % R == 0 && (WM / R) * 64 == SR, "wave rows nest in statistics rows");
  // LDS: [2 stages][one slab of 8 rows x CW floats per wave]. Nothing else: the statistics of a wave (2 x CW floats) and the next
  // tile ticket wait in the slabs between the end of a tile's stores and the next barrier (see epilogue1 / epilogue2) — the
  // 512 x 64 tile fills the CU's 160 KB to the byte.
  extern __shared__ __attribute__((aligned(128))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_s / WN, wn = wave_s % WN;
  const int W = gridDim.x;
  const int K = p.Ci, Nc = p.Nc, Kb = K * 4;
  const int lrow = lane & 31, lh = lane >> 5;

  // ---- DMA: wave w stages rows [w BM/NW, +BM/NW) of A and [w BN/NW, +BN/NW) of B, 8 rows (1 KiB) per instruction; the 16-byte
  // slot a lane fetches is XOR-swizzled by (row >> 1) & 7 (conv.hip, glds2 kernel). Offsets are constants of the kernel.
  unsigned voffA[AJ], voffB[BJ];
  const int KbB = (GATHER ? p.T : 1) * Kb;              // bytes of one weight row: [tap][Ci]
  {
    const int srow = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int r = wave_s * (BM / NW) + j * 8 + srow;
      voffA[j] = (unsigned)(r * Kb + ((pslot ^ ((r >> 1) & 7)) << 4));          // GATHER: recomputed per tile and tap (set_tap)
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int r = wave_s * (BN / NW) + j * 8 + srow;
      voffB[j] = (unsigned)(r * KbB + ((pslot ^ ((r >> 1) & 7)) << 4));
    }
  }
  // the source the NEXT DMA pieces read from (scalars): A / B descriptor of the tile being staged, K position of the step
  const float* dA = p.A;
  const float* dB = p.B;
  int dAbytes = 0, dsA = 0, dsB = 0;
  auto dma_piece = [&](auto stg_c, auto pc_c) __attribute__((always_inline)) {
    constexpr int STG = decltype(stg_c)::value, pc = decltype(pc_c)::value;
    if (R3M_PROBE(p) & 8) return;                         // timing probe: no DMA (stale LDS)
    // (the descriptors are loop-carried scalars; pinned to scalar registers or the compiler may park them in vector registers and
    // wrap each DMA instruction in a waterfall loop — conv_dev.h, buf_dma16_uniform)
    if constexpr (pc < AJ)
      buf_dma16_uniform(dA, dAbytes, smem + STG * STAGE + wave_s * (BM / NW) * 32 + pc * 8 * 32, voffA[pc], dsA);
    else
      buf_dma16_uniform(dB, BN * KbB, smem + STG * STAGE + BM * 32 + wave_s * (BN / NW) * 32 + (pc - AJ) * 8 * 32, voffB[pc - AJ], dsB);
  };
  auto dma_all = [&](auto stg_c) __attribute__((always_inline)) {
    static_for<NP>([&](auto pc_c) __attribute__((always_inline)) { dma_piece(stg_c, pc_c); });
  };
  // GATHER row state of the tile being staged: byte offset of the row's top-left input pixel from the tile's first frame (+ the
  // lane's swizzled slot), and (iy0 << 16 | ix0) for the bounds test; rows past M sit far outside every image
  unsigned poff[GATHER ? AJ : 1], iyx[GATHER ? AJ : 1];
  int tap_soffB = 0;
  auto aim_tile = [&](int tmt_, int tnt_) __attribute__((always_inline)) {     // descriptors (+ GATHER: row decode) of tile (tmt, tnt)
    // tile coordinates are wave-uniform; say so, or the descriptors end up in vector registers and every DMA instruction in a
    // waterfall loop (the uniform integer division below is expanded on the vector unit)
    const int tmt = __builtin_amdgcn_readfirstlane(tmt_), tnt = __builtin_amdgcn_readfirstlane(tnt_);
    dB = p.B + (long long)tnt * BN * (KbB / 4);
    if constexpr (!GATHER) {
      dA = p.A + (long long)tmt * BM * K;
      dAbytes = min(BM, p.M - tmt * BM) * Kb;
    } else {
      const int hw = p.Hg * p.Wg;
      const int m0t = tmt * BM;
      const int nf = __builtin_amdgcn_readfirstlane(m0t / hw);   // first frame of the tile (uniform): 32-bit offsets are relative to it
      const long long frame = (long long)p.Hi * p.Wi * K;
      dA = pw_uniform_ptr(p.A + nf * frame);
      const long long rest = (long long)(p.N - nf) * frame * 4;
      dAbytes = __builtin_amdgcn_readfirstlane(rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
      const int srow = lane >> 3, pslot = lane & 7;
      const int r0 = wave_s * (BM / NW) + srow;
      int m = m0t + r0;
      int n = m / hw;
      int rem = m - n * hw;
      int gy = rem / p.Wg;
      int gx = rem - gy * p.Wg;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int r = r0 + j * 8;
        const int iy0 = gy * p.is, ix0 = gx * p.is;
        poff[j] = (unsigned)(((((n - nf) * p.Hi + iy0) * p.Wi + ix0) * K) * 4 + ((pslot ^ ((r >> 1) & 7)) << 4));
        iyx[j] = m < p.M ? (unsigned)((iy0 << 16) | ix0) : 0x40004000u;
        m += 8;                                           // the lane's next row is 8 GEMM rows further: branch-free carries
        gx += 8;                                          // (Wg >= 4: at most two row wraps; Hg >= 2: at most two frame wraps — pw_gemm_form)
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
  auto set_tap = [&](int t) __attribute__((always_inline)) {                   // GATHER: per-lane offsets of tap t of the aimed tile
    if constexpr (GATHER) {
      const int pack = __builtin_amdgcn_readfirstlane(p.tap[t]);
      const int dy = (pack << 24) >> 24, dx = (pack << 16) >> 24, wt = pack >> 16;
      const int delta = (dy * p.Wi + dx) * Kb;            // scalar
      tap_soffB = wt * Kb;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const unsigned iy = (iyx[j] >> 16) + (unsigned)dy, ix = (iyx[j] & 0xFFFFu) + (unsigned)dx;
        voffA[j] = (iy < (unsigned)p.Hi && ix < (unsigned)p.Wi) ? poff[j] + (unsigned)delta : BUF_OOB;
      }
    }
  };
  auto aim_step = [&](int chunk) __attribute__((always_inline)) {              // K position inside the current tap: 32-channel chunk
    dsA = __builtin_amdgcn_readfirstlane(chunk * 128);
    dsB = __builtin_amdgcn_readfirstlane(tap_soffB + chunk * 128);
  };

  // ---- fragments (as in the glds2 kernel: lane half h reads k = 8g + 4h .. +3 of group g; MFMA j contracts k = {8g + j, 8g + 4 + j})
  const float* fa[4];
  const float* fb[4];
  {
    const int xr = (lrow >> 1) & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int go = ((2 * g + lh) ^ xr) * 4;
      fa[g] = smem + (wm * 64 + lrow) * 32 + go;
      fb[g] = smem + BM * 32 + (wn * TN * 32 + lrow) * 32 + go;
    }
  }
  f32x16 acc[2][TN];

  // ---- epilogue operands of the tile being computed, prefetched into registers (read-modify-write epilogues only)
  // acc[tm][tn][r] is row m0 + wm 64 + tm 32 + 8 (r >> 2) + 4 lh + (r & 3), column n0 + wn 64 + tn 32 + lrow.
  constexpr bool MADD = (EPI & EPI_MASKED_ADD) != 0, BNR = (EPI & EPI_BNRED) != 0, ACC = (EPI & EPI_ACCUM) != 0;
  constexpr bool AFF = (EPI & EPI_AFFINE) != 0, RELU = (EPI & EPI_RELU) != 0;   // round 6, inference: eval-mode BatchNorm (+ residual in `out`) (+ ReLU) at the store
  static_assert(!(AFF && (BNR || MADD)) && (!RELU || AFF), "EPI_AFFINE / EPI_RELU: forward epilogues");
  constexpr bool PRE = MADD || BNR || ACC;
  static_assert(!(MADD && ACC), "one added tensor");
  // Store layout (after the per-wave LDS transposition of the epilogue): a lane owns FOUR consecutive columns of one row; a
  // dwordx4 instruction covers 4 rows x 64 columns. Store s = (tm, q, i) of 16 holds wave-tile rows tm 32 + 8 q + 4 i + (lane >> 4),
  // columns 4 (lane & 15) .. +3. (Measured, tools/micro/storepat.hip: a CU retires ~1 store instruction per 14-19 cycles whatever
  // its width, so 16 dwordx4 per wave tile cost a third of 64 dwords; the same goes for the operand loads.)
  constexpr int LPR = 8 * TN, RPS = 64 / LPR;             // lanes per row of a store, rows per store (TN = 2: 16, 4; TN = 1: 8, 8)
  const int srow = lane / LPR, scol = (lane % LPR) * 4;
  const unsigned vo4 = (unsigned)(((wm * 64 + srow) * Nc + wn * TN * 32 + scol) * 4);      // per-lane byte offset in the tile (constant)
  const int Nw = Nc >> 5;                                                                   // mask words per row
  const unsigned vow = (unsigned)(((wm * 64 + lane) * Nw + wn * TN) * 4);                   // mask word(s) of row `lane`, this wave's columns
  f32x4 pg[PRE ? NST : 1], py[PRE ? NST : 1];             // add0 (or the old result) and y of the tile, store layout
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 pgm = {0u, 0u}, pym = {0u, 0u};                   // lane l: the two mask words of tile row wm 64 + l
  const float* t_gb = nullptr;                            // set per tile (scalars): operand bases at the tile origin + byte counts
  const float* t_yb = nullptr;
  const unsigned* t_gbits = nullptr;
  const unsigned* t_ybits = nullptr;
  int t_obytes = 0, t_bbytes = 0;
  static_assert(!OSTR || (GATHER && !MADD && !YBITS), "strided output rows: gather form, no mask words");
  unsigned orows[OSTR ? NST : 1];                         // OSTR: byte offset of this lane's row of store st (+ its columns); rows >= M: out of range
  long long t_eo = 0;                                     // OSTR: element offset of (first frame of the tile, column n0)
  auto pre_tile = [&](int tmt, int tnt) __attribute__((always_inline)) {
    if constexpr (OSTR) {
      const int m0 = tmt * BM, n0 = tnt * BN;
      const int hw = p.Hg * p.Wg;
      const int nf = __builtin_amdgcn_readfirstlane(m0 / hw);       // first frame of the tile: 32-bit offsets are relative to it
      const long long oframe = (long long)p.Ho * p.Wo * Nc;
      t_eo = nf * oframe + n0;
      const long long rest = ((long long)(p.N - nf) * oframe - n0) * 4;
      t_obytes = __builtin_amdgcn_readfirstlane(rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
      const int m = m0 + wm * 64 + lane;                  // lane l decodes row l of the wave's 64
      const int n = m / hw;
      const int rem = m - n * hw;
      const int gy = rem / p.Wg, gx = rem - gy * p.Wg;
      const unsigned mine = m < p.M ? (unsigned)(((((n - nf) * p.Ho + gy * p.os + p.ooy) * p.Wo + gx * p.os + p.oox) * Nc) * 4) : BUF_OOB;
#pragma unroll
      for (int st = 0; st < NST; ++st)
        orows[st] = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (st * RPS + srow), (int)mine) + (unsigned)((wn * TN * 32 + scol) * 4);
      if constexpr (ACC) t_gb = p.out + t_eo;
      if constexpr (BNR) t_yb = p.bn_y + t_eo;
    } else if constexpr (PRE) {
      const int m0 = tmt * BM, n0 = tnt * BN;
      const int rows_valid = min(BM, p.M - m0);
      const long long eo0 = (long long)m0 * Nc + n0;
      t_obytes = ((rows_valid - 1) * Nc + BN) * 4;        // a lane's offset is inside iff its row is < rows_valid
      t_bbytes = ((rows_valid - 1) * Nw + BN / 32) * 4;
      if constexpr (MADD) { t_gb = p.add0 + eo0; t_gbits = p.addbits + (eo0 >> 5); }
      if constexpr (ACC) t_gb = p.out + eo0;
      if constexpr (BNR) { t_yb = p.bn_y + eo0; t_ybits = YBITS ? p.bn_bits + (eo0 >> 5) : nullptr; }
    }
  };
  // piece pc of NST = the operands of store pc (one dwordx4 load per tensor); piece 0 also fetches the mask words
  auto pre_piece = [&](auto pc_c) __attribute__((always_inline)) {
    if constexpr (PRE) {
      constexpr int pc = decltype(pc_c)::value;
      if (R3M_PROBE(p) & 16) return;                      // timing probe: no epilogue operand loads (stale registers; wrong results)
      // the row stride is made opaque here so that the scalar row offsets of a tile are s_mul'ed where they are used instead of
      // being hoisted out of the tile loop (they do not fit the SGPR file and would be reloaded with v_readlane)
      int ncb = Nc * 4;
      asm volatile("" : "+s"(ncb));
      const int so = OSTR ? 0 : (pc * RPS) * ncb;         // first row of store pc
      const unsigned vo = OSTR ? orows[OSTR ? pc : 0] : vo4;
      if constexpr (MADD || ACC) pg[pc] = pw_ld4(t_gb, t_obytes, vo, so);
      if constexpr (BNR) py[pc] = pw_ld4(t_yb, t_obytes, vo, so);
      if constexpr (pc == 0) {
        if constexpr (MADD) {
          if constexpr (TN == 2) pgm = pw_ld2(t_gbits, t_bbytes, vow);
          else pgm[0] = pw_ldu(t_gbits, t_bbytes, vow, 0);
        }
        if constexpr (BNR && YBITS) {
          if constexpr (TN == 2) pym = pw_ld2(t_ybits, t_bbytes, vow);
          else pym[0] = pw_ldu(t_ybits, t_bbytes, vow, 0);
        }
      }
    }
  };

  // one K step: 64 MFMAs on stage STG_M; FIRST: the accumulators start from zero (C operand = 0, no clears);
  // DMA: the pieces of another step go out between the MFMAs into stage STG_M ^ 1 (one per 8 MFMAs)
  auto kstep = [&](auto stgm_c, auto first_c, auto dma_c, auto pf_c, bool do_dma, bool do_pf) __attribute__((always_inline)) {
    constexpr int STG_M = decltype(stgm_c)::value, PF = decltype(pf_c)::value;   // PF: 0 none, 1 / 2: first / second half of the tile's epilogue operands
    constexpr bool FIRST = decltype(first_c)::value, DMA = decltype(dma_c)::value;
    using SD = std::integral_constant<int, STG_M ^ 1>;
    // fragments of group g + 1 are requested before the MFMAs of group g (two register sets; one where the prefetched epilogue
    // operands of EPI_BNRED | EPI_MASKED_ADD leave no room: 64 accumulators + 128 operands)
    // (likewise the 512-row gather tile with EPI_BNRED: 64 accumulators + 64 operands + the row state of 8 staged rows per lane)
    constexpr int FB = ((MADD && BNR) || (GATHER && BNR && AJ > 4)) ? 1 : 2;
    f32x4 af[FB][2], bf[FB][TN];
    auto frag_load = [&](auto g_c) __attribute__((always_inline)) {
      constexpr int g = decltype(g_c)::value;
#pragma unroll
      for (int t = 0; t < 2; ++t) af[g % FB][t] = *reinterpret_cast<const f32x4*>(fa[g] + STG_M * STAGE + t * 32 * 32);
#pragma unroll
      for (int t = 0; t < TN; ++t) bf[g % FB][t] = *reinterpret_cast<const f32x4*>(fb[g] + STG_M * STAGE + t * 32 * 32);
    };
    if constexpr (FB == 2) frag_load(std::integral_constant<int, 0>{});
    static_for<4>([&](auto g_c) __attribute__((always_inline)) {
      constexpr int g = decltype(g_c)::value;
      if constexpr (FB == 1) frag_load(g_c);
      if constexpr (FB == 2 && g < 3) frag_load(std::integral_constant<int, g + 1>{});
      const f32x4(&a)[2] = af[g % FB];
      const f32x4(&b)[TN] = bf[g % FB];
      static_for<4>([&](auto j_c) __attribute__((always_inline)) {
        constexpr int j = decltype(j_c)::value;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            if constexpr (FIRST && g == 0 && j == 0) {
              const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], z, 0, 0, 0);
            } else {
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
            }
          }
        if constexpr (PRE && PF != 0) {
          constexpr int slot = g * 4 + j;
          // slot s of this step carries operand piece (PF - 1) * NST / 2 + s / 2 (even slots; the DMA pieces ride on the odd ones).
          // (Round 5: all pieces on the step's FIRST slots, two per slot — so that the last one has most of a K step instead of two
          // slots to come back from HBM before the step's vmcnt(0) — measured no different, launch by launch and in the step:
          // the CU's other block covers the wait. profiles/r05_rmw_dgrad.txt)
          if constexpr ((slot & 1) == 0 && (slot >> 1) < NST / 2) {
            if (do_pf) {
              __builtin_amdgcn_sched_barrier(0);
              pre_piece(std::integral_constant<int, (PF - 1) * (NST / 2) + (slot >> 1)>{});
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if constexpr (DMA) {
          // 16 slots of 2 TN MFMAs; piece k rides on odd slot 2 k + 1
          constexpr int slot = g * 4 + j;
          constexpr bool fire = (slot & 1) == 1 && (slot >> 1) < NP;
          constexpr int piece = slot >> 1;
          if constexpr (fire) {
            if (do_dma) {
              __builtin_amdgcn_sched_barrier(0);
              dma_piece(SD{}, std::integral_constant<int, piece>{});
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      });
    });
  };

  // ---- epilogue, part 1 (stores + partial sums) of tile (emt, ent). BatchNorm forward statistics come straight from the
  // accumulators (a lane owns a column there: in-lane sums). The result leaves through a per-wave LDS slab of 8 rows x 64 columns
  // (2 KB): the four registers r = 4 q .. 4 q + 3 of both column tiles are written (8 ds_write_b32, immediates only), read back as
  // two float4 per lane and stored with two buffer_store_dwordx4 whose row offset is a scalar. A wave's LDS accesses execute in
  // order, so the slab needs no barrier; read-modify-write operands are in pg / py / pgm / pym by now (their loads were waited
  // for with the K step's DMA).
  constexpr int CW = TN * 32;                             // columns of a wave tile = floats per slab row
  float* const slabs = smem + 2 * STAGE;
  float* slab = slabs + wave_s * (8 * CW);
  float* slab_w = slab + 4 * lh * CW + lrow;              // + (e CW + tn 32): element (row 4 lh + e, column tn 32 + lrow)
  const float* slab_r = slab + srow * CW + scol;          // + i RPS CW: row i RPS + srow, columns scol .. +3
  const int bp0 = 4 * srow;                               // ds_bpermute byte address of lane `srow` (+ 4 x the store's first row)
  const bool hiw = TN == 2 && (lane & 8) != 0;            // the lane's columns lie in the second mask word of the wave's 64
  const int nsh = (lane & 7) * 4;                         // ... at this bit
  auto mask_nibble = [&](const u32x2& words, int st) __attribute__((always_inline)) -> unsigned {
    const unsigned w0 = (unsigned)__builtin_amdgcn_ds_bpermute(bp0 + st * RPS * 4, (int)words[0]);   // mask word(s) of the store's row
    unsigned w = w0;
    if constexpr (TN == 2) {
      const unsigned w1 = (unsigned)__builtin_amdgcn_ds_bpermute(bp0 + st * RPS * 4, (int)words[1]);
      w = hiw ? w1 : w0;
    }
    return (w >> nsh) & 15u;
  };
  auto epilogue1 = [&](int emt, int ent) __attribute__((always_inline)) {
    const int m0 = emt * BM, n0 = ent * BN;
    if (R3M_PROBE(p) & 4) return;                         // timing probes (probe builds only; wrong results)
    float st_s[TN], st_ss[TN];
    if ((EPI & EPI_STATS) != 0 && !(R3M_PROBE(p) & 2)) {
      // same summation order as gg_stats (conv_dev.h): rows >= M were staged as zeros and add nothing
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[tm][tn][r];
            s += v;
            ss = fmaf(v, v, ss);
          }
        st_s[tn] = s + __shfl_xor(s, 32);
        st_ss[tn] = ss + __shfl_xor(ss, 32);
      }
    }
    const int rows_valid = min(BM, p.M - m0);
    const long long eo0 = (long long)m0 * Nc + n0;                  // element offset of the tile origin
    const int obytes = OSTR ? t_obytes : ((rows_valid - 1) * Nc + BN) * 4;   // a lane's offset is inside iff its row is < rows_valid
    float* ob = p.out + (OSTR ? t_eo : eo0);              // (OSTR: the tile's own values are still in place — pre_tile of the next tile comes after)
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, mu = s1, sc = s1, sh = s1;
    if constexpr (BNR) {
      const int col = n0 + wn * CW + scol;
      mu = *reinterpret_cast<const f32x4*>(p.bn_mean + col);
      if constexpr (!YBITS) {
        sc = *reinterpret_cast<const f32x4*>(p.bn_scale + col);
        sh = *reinterpret_cast<const f32x4*>(p.bn_shift + col);
      }
    }
    if constexpr (AFF) {
      const int col = n0 + wn * CW + scol;
      sc = *reinterpret_cast<const f32x4*>(p.bn_scale + col);
      sh = *reinterpret_cast<const f32x4*>(p.bn_shift + col);
    }
    static_for<8>([&](auto c_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value;             // chunk (tm, q): wave-tile rows 8 c .. 8 c + 7
      constexpr int tm = c >> 2, q = c & 3;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int e = 0; e < 4; ++e) slab_w[e * CW + tn * 32] = acc[tm][tn][4 * q + e];
      __builtin_amdgcn_wave_barrier();
      int ncb = Nc * 4;
      asm volatile("" : "+s"(ncb));                       // see pre_piece: row offsets computed at the point of use
#pragma unroll
      for (int i = 0; i < TN; ++i) {                      // TN stores of RPS rows each cover the chunk's 8 rows
        const int st = c * TN + i;                        // store index 0 .. NST - 1: rows st RPS + srow
        f32x4 v = *reinterpret_cast<const f32x4*>(slab_r + i * RPS * CW);
        if constexpr (AFF) {                               // the same fmaf(y, scale, shift) bn_act_fwd computes (csrc/bn.hip)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
        }
        if constexpr (ACC) v += pg[st];
        if constexpr (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (MADD) {
          const unsigned nb = mask_nibble(pgm, st);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += ((nb >> e) & 1u) ? pg[st][e] : 0.f;
        }
        if (!(R3M_PROBE(p) & 1)) pw_st4(ob, obytes, OSTR ? orows[OSTR ? st : 0] : vo4, OSTR ? 0 : (st * RPS) * ncb, v);
        if constexpr (BNR) {
          const f32x4 y = py[st];
          unsigned nb = 0u;
          if constexpr (YBITS) nb = mask_nibble(pym, st);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool on = YBITS ? (((nb >> e) & 1u) != 0u) : (fmaf(y[e], sc[e], sh[e]) > 0.f);
            const float gg = on ? v[e] : 0.f;
            s1[e] += gg;
            s2[e] = fmaf(gg, y[e] - mu[e], s2[e]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();                    // the chunk's slab reads are issued before the next chunk's writes
    });
    if ((EPI & EPI_STATS) != 0 && !(R3M_PROBE(p) & 2)) {
      // the wave's column sums wait in ITS OWN slab, [2][CW], for epilogue2 (after the next barrier); a wave's LDS accesses execute in
      // order, so they land behind the slab traffic above, and the slab is next written by the next tile's epilogue1 — barriers later
      if (lane < 32) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          slab[tn * 32 + lane] = st_s[tn];
          slab[CW + tn * 32 + lane] = st_ss[tn];
        }
      }
    }
    if constexpr (BNR) {
      // the wave's 64 rows are one partial row of the consumer BatchNorm's backward sums (geometry of bnred_partial_rows):
      // lanes l, l + LPR, l + 2 LPR, ... hold the same four columns
      const int g0 = m0 + wm * 64;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
          s1[e] += __shfl_xor(s1[e], o);
          s2[e] += __shfl_xor(s2[e], o);
        }
      if (lane < LPR && g0 < p.M) {
        const long long prow = g0 >> 6;
        const int col = n0 + wn * CW + scol;
        *reinterpret_cast<f32x4*>(p.stats + (prow * 2 + 0) * Nc + col) = s1;
        *reinterpret_cast<f32x4*>(p.stats + (prow * 2 + 1) * Nc + col) = s2;
      }
    }
  };
  // part 2 (EPI_STATS, one barrier after part 1): combine the wave rows — stats[mt][2][Nc] as the other kernels write it
  auto epilogue2 = [&](int emt, int ent) __attribute__((always_inline)) {
    if constexpr ((EPI & EPI_STATS) != 0) {
      if (tid < R * BN) {
        const int r = tid / BN, c = tid - r * BN;           // statistics row r of the tile, column c
        const float* src = slabs + (c / CW) * (8 * CW) + (c % CW);    // wave column c / CW, its slab's [0][c % CW]
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int w = 0; w < WM / R; ++w) {                  // the wave rows of that statistics row, top to bottom
          const float* q = src + ((r * (WM / R) + w) * WN) * (8 * CW);
          s += q[0];
          ss += q[CW];
        }
        const long long srow = (long long)emt * R + r;
        if (srow * SR < p.M) {                              // (a tile's second statistics row may lie past M)
          const int col = ent * BN + c;
          p.stats[(srow * 2 + 0) * Nc + col] = s;
          p.stats[(srow * 2 + 1) * Nc + col] = ss;
        }
      }
    }
  };

  // ---- persistent tile walk.
  // Static (p.tile_ctr == null): worker w takes tiles w, w + W, ...; workers of one XCD hold neighbouring ids (the column tiles of
  // one row panel share that XCD's L2). Dynamic (the engine's launches): eight queues, one per XCD — block b draws from queue
  // b % 8 (blocks are dispatched round-robin over the XCDs; if that ever changes only locality is lost), whose k-th ticket is column
  // tile k % gridN of row panel 8 (k / gridN) + queue. A block takes its next ticket at the START of a tile (one atomic by one
  // lane, its latency rides under the tile) and hands it to the other waves through LDS at the tile's second barrier. With a
  // static split a block that cannot become resident — its CU is shared with another stream's kernel, e.g. RCCL during an
  // overlapped all-reduce — would start its whole share only after another block has finished: twice the launch time; with the
  // queues it finds them (nearly) empty.
  const int hp = K >> 6;                                 // pairs of 32-wide K steps per tap (Ci is a multiple of 64)
  const int npairs = (GATHER ? p.ntaps : 1) * hp;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using T_ = std::true_type;
  using F_ = std::false_type;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  const bool dyn = p.tile_ctr != nullptr;
  const int xq = blockIdx.x & 7;
  unsigned* ctr = dyn ? p.tile_ctr + xq : nullptr;
  int* nxt = reinterpret_cast<int*>(slabs + 2 * CW);     // the next ticket, one int: in wave 0's slab behind its statistics (written after
                                                         // epilogue1 has used the slab, read after the next barrier)
  const int gridM = tiles / gridN;
  auto ticket_tile = [&](int k, int& tmt, int& tnt) -> bool {
    const int kp = k / gridN;
    tnt = k - kp * gridN;
    tmt = kp * 8 + xq;
    return tmt < gridM;
  };
  int id = 0, mt = 0, nt = 0;
  bool has;
  int dm = 0, dn = 0;
  if (dyn) {
    if (tid == 0) *nxt = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    has = ticket_tile(__builtin_amdgcn_readfirstlane(*nxt), mt, nt);
  } else {
    id = xcd_remap(blockIdx.x, W);
    mt = id / gridN;
    nt = id - mt * gridN;
    dm = W / gridN;
    dn = W - dm * gridN;
    has = id < tiles;
  }

  // K steps of a tile: (tap, 32-channel chunk), chunk fastest; `hp` pairs of steps per tap, `npairs` per tile. The DMA of step
  // s + 1 goes out during step s: the even step of a pair carries its odd step's (same tap), the odd step carries the next pair's
  // even step — which may belong to the next tap (set_tap first) or to the next tile (aim_tile + set_tap(0) first).
  bool pending = false;
  int pmt = 0, pnt = 0;
  if (has) {
    aim_tile(mt, nt);
    set_tap(0);
    aim_step(0);
    dma_all(I0{});
  }
  while (true) {
    unsigned ticket = 0u;
    if (has) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // step 0 of this tile has landed (and the previous tile's stores are out,
      __syncthreads();                                   //  its prefetched epilogue operands are in their registers)
      aim_step(1);
      dma_all(I1{});
      if (dyn && tid == 0) ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (pending) epilogue1(pmt, pnt);
    if (!has) break;
    pre_tile(mt, nt);
    kstep(I0{}, T_{}, F_{}, P1{}, false, npairs == 1);
    int nid = id + W;
    int nmt = mt + dm, nnt = nt + dn;
    if (nnt >= gridN) { nnt -= gridN; ++nmt; }
    bool nhas = nid < tiles;
    int ti = 0, cp = 0;                                  // tap and pair-within-tap of pair `pr`
    for (int pr = 0; pr < npairs; ++pr) {
      const bool last = pr + 1 == npairs;
      if (pr > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        aim_step(2 * cp + 1);
        if constexpr (BURST) {
          dma_all(I1{});
          kstep(I0{}, F_{}, F_{}, P1{}, false, last);
        } else if (R3M_PROBE(p) & 32) {                   // (probe 32: burst in every variant)
          dma_all(I1{});
          kstep(I0{}, F_{}, F_{}, P1{}, false, last);
        } else {
          kstep(I0{}, F_{}, T_{}, P1{}, true, last);
        }
      }
      if (pr == 0 && dyn && tid == 0) *nxt = (int)ticket;
      // (Tried and dropped: `s_waitcnt vmcnt(NST)` + a bare s_barrier here after a deferred epilogue, on the assumption that the
      // counter retires loads, LDS-DMA and stores strictly in issue order so that "all but the NST youngest" covers the DMA and
      // leaves the stores flying. It bought nothing measurable — the K = 64 launch ran 1.57 -> 1.62 ms — and an intermittent wrong
      // result showed up once in 231 GPU tests while it was in; the plain full wait costs the same and assumes nothing.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (pr == 0 && dyn) nhas = ticket_tile(__builtin_amdgcn_readfirstlane(*nxt), nmt, nnt);
      if (pr == 0 && pending) epilogue2(pmt, pnt);
      // aim the DMA of this odd step at the even step of the next pair
      bool do_dma = true;
      if (last) {
        do_dma = nhas;
        if (nhas) {
          aim_tile(nmt, nnt);
          set_tap(0);
          aim_step(0);
        }
      } else if (cp + 1 == hp) {
        set_tap(ti + 1);
        aim_step(0);
      } else {
        aim_step(2 * cp + 2);
      }
      if constexpr (BURST) {
        if (do_dma) dma_all(I0{});
        kstep(I1{}, F_{}, F_{}, P2{}, false, last);
      } else if (R3M_PROBE(p) & 32) {
        if (do_dma) dma_all(I0{});
        kstep(I1{}, F_{}, F_{}, P2{}, false, last);
      } else {
        kstep(I1{}, F_{}, T_{}, P2{}, do_dma, last);
      }
      if (++cp == hp) { cp = 0; ++ti; }
    }
    pending = true;
    pmt = mt;
    pnt = nt;
    id = nid;
    mt = nmt;
    nt = nnt;
    has = nhas;
  }
  if constexpr ((EPI & EPI_STATS) != 0) {
    if (pending) {
      __syncthreads();
      epilogue2(pmt, pnt);
    }
  }
}

// ---- launcher ----------------------------------------------------------------------------------------------------------
static int pw_cu_count() {
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

static bool pw_flags_ok(const GatherGemmParams& p) {
  switch (p.flags) {
    case 0: case EPI_STATS: case EPI_ACCUM: case EPI_BNRED: return true;
    case EPI_AFFINE: case EPI_AFFINE | EPI_RELU: case EPI_AFFINE | EPI_ACCUM | EPI_RELU:      // inference forward (dense output rows only)
      return p.os == 1 && p.ooy == 0 && p.oox == 0 && p.Hg == p.Ho && p.Wg == p.Wo && p.bn_scale && p.bn_shift;
    case EPI_MASKED_ADD: case EPI_BNRED | EPI_MASKED_ADD: return p.addbits != nullptr;
    default: return false;
  }
}

// 0: not for this kernel; 1: pointwise form (1x1 / stride 1: A rows are matrix rows); 2: gather form (dense OUTPUT rows: forward
// of any geometry, dgrad of stride-1 layers); 3: gather form with strided output rows (a parity class of a stride-2 dgrad)
int pw_gemm_form(const GatherGemmParams& p) {
  if (p.dtype != DT_F32) return 0;
  const bool dense_out = p.os == 1 && p.ooy == 0 && p.oox == 0 && p.Hg == p.Ho && p.Wg == p.Wo;
  if ((p.Ci & 63) || p.Ci > 2048 || (p.Nc & 63) || !pw_flags_ok(p)) return 0;   // widths: multiples of 128 (four-wave tile) or 64 (eight-wave)
  if (dense_out && p.simple_rows && p.ntaps == 1 && p.T == 1 && p.dy[0] == 0 && p.dx[0] == 0 && p.wt[0] == 0) return 1;
  if (p.simple_rows || p.ntaps < 1 || p.ntaps > MAX_TAPS) return 0;
  // 32-bit offsets: a tile's rows span at most ceil(256 / (Hg Wg)) + 1 frames of the input; one weight tile [128][T][Ci]
  if (p.Hi >= 16384 || p.Wi >= 16384 || p.Hi < 1 || p.Wi < 1 || p.Wg < 4 || p.Hg < 2) return 0;
  const long long frame = (long long)p.Hi * p.Wi * p.Ci * 4;
  const long long span = (512 / ((long long)p.Hg * p.Wg) + 2) * frame;     // (the largest tile has 512 rows)
  if (span >= (long long)BUF_OOB || 128LL * p.T * p.Ci * 4 >= (long long)BUF_OOB) return 0;
  if (!dense_out) {
    // strided output rows: every output pixel of the class inside the tensor, 32-bit offsets over the frames a tile spans, and the
    // epilogues a stride-2 dgrad uses (plain, accumulate, BatchNorm-backward partials with the mask recomputed from y)
    if (p.os < 1 || p.ooy < 0 || p.oox < 0 || (p.Hg - 1) * p.os + p.ooy >= p.Ho || (p.Wg - 1) * p.os + p.oox >= p.Wo) return 0;
    if ((256 / ((long long)p.Hg * p.Wg) + 2) * (long long)p.Ho * p.Wo * p.Nc * 4 >= (long long)BUF_OOB) return 0;
    if (p.flags != 0 && p.flags != EPI_ACCUM && !(p.flags == EPI_BNRED && !p.bn_bits)) return 0;
    return 3;
  }
  // four-wave gather form with the two register-hungriest epilogues (64 accumulators + 128 prefetched operands + the row state) would
  // spill; no ResNet layer needs them (128-wide 3x3 / stride-1 dgrads with W <= 28 run the window kernel): left to the gather kernel
  if ((p.Nc & 127) == 0 && (p.flags & EPI_BNRED) && ((p.flags & EPI_MASKED_ADD) || p.bn_bits)) return 0;
  return 2;
}
bool pw_gemm_eligible(const GatherGemmParams& p) { return pw_gemm_form(p) != 0; }

// Launches that run the burst variant: contracting ones (K >= 2 N) on the eight-wave tile — ONE block per CU, so no co-resident block
// covers a late DMA piece: -2.4 % on the 256 -> 64 launches, -0.7 % on the 64-channel 3x3 in the step; on the four-wave tile (two
// blocks per CU) the two issue orders measured the same within noise. Built for the three epilogues those launches use.
static bool pw_burst(const GatherGemmParams& p) {
  const long long ktot = (long long)(p.simple_rows ? 1 : p.ntaps) * p.Ci;
  return (p.Nc & 127) != 0 && ktot >= 2LL * p.Nc &&
         (p.flags == EPI_STATS || p.flags == 0 || p.flags == (EPI_AFFINE | EPI_RELU) || (p.flags == EPI_BNRED && !p.bn_bits));
}

static bool pw_tile512() { return R3M_ENV_INT("R3M_PW_512", 1) != 0; }   // probe builds: 0 = the 256 x 64 tile everywhere (A/B)

template <int BM, int BN, int WM, int WN, bool GA, bool OS = false>
static int launch_pw_shape(const GatherGemmParams& p_in, hipStream_t s) {
  constexpr int NW = WM * WN, TN = BN / WN / 32;
  const int gridM = ceil_div(p_in.M, BM), gridN = p_in.Nc / BN;
  const long long tiles_ll = (long long)gridM * gridN;
  R3M_REQUIRE(tiles_ll < 0x7FFFFFFFLL, "pw_gemm: too many tiles");
  const int tiles = (int)tiles_ll;
  const int slots = (NW == 4 ? (BN == 64 ? 3 : 2) : 1) * pw_cu_count();   // resident blocks: two four-wave blocks or one eight-wave block per CU (experiment: three 128 x 64 blocks)
  const int W = tiles < slots ? tiles : slots;
  GatherGemmParams p = p_in;
  if (W < 64 || gridM < 64) p.tile_ctr = nullptr;                     // small launches: every queue needs blocks AND panels; static split
  constexpr int LDS = (2 * (BM + BN) * 32 + NW * 8 * TN * 32) * 4;   // ring + one 8-row store slab per wave (statistics and ticket wait inside the slabs)
  static_assert(LDS <= 160 * 1024, "one CU's LDS");
  constexpr bool BURST_ONLY = BM / (8 * NW) + BN / (8 * NW) > 8;      // more DMA pieces per wave than odd MFMA slots: built in the burst form only
#define LAUNCH_PW(E, YB)                                                                                                            \
  do {                                                                                                                             \
    static DynLdsOptIn oi;                                                                                                         \
    if (int e = ensure_dyn_lds(oi, reinterpret_cast<const void*>(pw_gemm_kernel<BM, BN, WM, WN, E, YB, GA, OS>), LDS, "pw_gemm")) return e; \
    hipLaunchKernelGGL((pw_gemm_kernel<BM, BN, WM, WN, E, YB, GA, OS>), dim3(W), dim3(NW * 64), LDS, s, p, tiles, gridN);               \
  } while (0)
#define LAUNCH_PW_BURST(E)                                                                                                          \
  do {                                                                                                                             \
    static DynLdsOptIn oi;                                                                                                         \
    if (int e = ensure_dyn_lds(oi, reinterpret_cast<const void*>(pw_gemm_kernel<BM, BN, WM, WN, E, false, GA, false, true>), LDS, "pw_gemm")) return e; \
    hipLaunchKernelGGL((pw_gemm_kernel<BM, BN, WM, WN, E, false, GA, false, true>), dim3(W), dim3(NW * 64), LDS, s, p, tiles, gridN);   \
  } while (0)
  const bool yb = p.bn_bits != nullptr;
  if constexpr (OS) {                                                 // strided output rows: the three epilogues pw_gemm_form admits
    switch (p.flags) {
      case 0: LAUNCH_PW(0, false); return 0;
      case EPI_ACCUM: LAUNCH_PW(EPI_ACCUM, false); return 0;
      case EPI_BNRED: if (!yb) { LAUNCH_PW(EPI_BNRED, false); return 0; }
    }
    set_last_error("pw_gemm: form not built");
    return 1;
  } else {
  if constexpr (NW == 8) if (pw_burst(p)) {
    switch (p.flags) {
      case 0: LAUNCH_PW_BURST(0); return 0;
      case EPI_STATS: LAUNCH_PW_BURST(EPI_STATS); return 0;
      case EPI_AFFINE | EPI_RELU: LAUNCH_PW_BURST(EPI_AFFINE | EPI_RELU); return 0;
      case EPI_BNRED: LAUNCH_PW_BURST(EPI_BNRED); return 0;
    }
  }
  if constexpr (BURST_ONLY) {
    set_last_error("pw_gemm: form not built");
    return 1;
  } else
  switch (p.flags) {
    case 0: LAUNCH_PW(0, false); break;
    case EPI_STATS: LAUNCH_PW(EPI_STATS, false); break;
    case EPI_ACCUM: LAUNCH_PW(EPI_ACCUM, false); break;
    case EPI_AFFINE: LAUNCH_PW(EPI_AFFINE, false); break;
    case EPI_AFFINE | EPI_RELU: LAUNCH_PW(EPI_AFFINE | EPI_RELU, false); break;
    case EPI_AFFINE | EPI_ACCUM | EPI_RELU: LAUNCH_PW(EPI_AFFINE | EPI_ACCUM | EPI_RELU, false); break;
    case EPI_MASKED_ADD: LAUNCH_PW(EPI_MASKED_ADD, false); break;
    case EPI_BNRED:
      if (!yb) LAUNCH_PW(EPI_BNRED, false);
      else if constexpr (!(GA && NW == 4)) LAUNCH_PW(EPI_BNRED, true);
      else { set_last_error("pw_gemm: form not built"); return 1; }
      break;
    case EPI_BNRED | EPI_MASKED_ADD:
      if constexpr (!(GA && NW == 4)) {
        if (yb) LAUNCH_PW(EPI_BNRED | EPI_MASKED_ADD, true); else LAUNCH_PW(EPI_BNRED | EPI_MASKED_ADD, false);
      } else { set_last_error("pw_gemm: form not built"); return 1; }
      break;
    default: set_last_error("pw_gemm: unsupported epilogue flag combination %d", p.flags); return 1;
  }
  return 0;
  }
#undef LAUNCH_PW
#undef LAUNCH_PW_BURST
}

int launch_pw_gemm(const GatherGemmParams& p, hipStream_t s) {
  const bool wide = (p.Nc & 127) == 0;
  const int form = pw_gemm_form(p);
  // (round 4's probe-build experiment of a 128 x 64 tile at three blocks per CU — profiles/r04_narrow_tile128_ab.txt: no gain — left the
  // file in round 6: its statistics rows no longer nest in the 256-row geometry the kernel asserts since the 512 x 64 tile)
  if (form == 3) return wide ? launch_pw_shape<128, 128, 2, 2, true, true>(p, s) : launch_pw_shape<256, 64, 4, 2, true, true>(p, s);
  // 64-channel outputs, contracting launches (the burst form: conv1 / conv2 forward, conv2 / conv3 dgrad of layer1): 512 x 64 tile, eight
  // waves of 64 x 64 like the 128-wide kernel's — half the fragment reads and barriers per MFMA of the 256 x 64 tile's 64 x 32 waves
  if (!wide && pw_burst(p) && pw_tile512()) return form == 2 ? launch_pw_shape<512, 64, 8, 1, true>(p, s) : launch_pw_shape<512, 64, 8, 1, false>(p, s);
  if (form == 2) return wide ? launch_pw_shape<128, 128, 2, 2, true>(p, s) : launch_pw_shape<256, 64, 4, 2, true>(p, s);
  return wide ? launch_pw_shape<128, 128, 2, 2, false>(p, s) : launch_pw_shape<256, 64, 4, 2, false>(p, s);
}

}  // namespace r3m
