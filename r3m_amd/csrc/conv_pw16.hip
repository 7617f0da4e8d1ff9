// r3m_amd — persistent big-tile GEMM on bf16 operands for gfx950 (MI355X): forward / dgrad launches of the bf16 plans (BASELINE
// configs[2], [4]) whose OUTPUT rows are dense — the 1x1 convolutions, strided forward launches, stride-1 dgrads.
// out[M x Nc] = sum_t A[pix(m) + tap_t][K] * B[Nc][t][K]^T, bf16 in HBM / LDS, fp32 accumulation on v_mfma_f32_32x32x16_bf16, one
// rounding to bf16. Reference call site: the torchvision convolutions reached from /root/reference/r3m/models/models_r3m.py:99 (the
// reference is fp32-only; precision="bf16" is this build's counterpart of torch.autocast around that call).
//
// How it got its shape (round 5; evidence under profiles/r05_pw16_*). The per-tile kernels of conv_bf16.hip sit on NEITHER roof
// (0.19-0.41 of HBM, 0.25-0.43 of the MFMA peak). The first three builds of this file kept their 128 x 128 tile and
// made the block persistent with a three-stage ring and WARP SPECIALISATION (compute waves that only DMA and multiply, store waves
// that drain an LDS out-buffer, so that a compute wave's vmcnt counts nothing but its own in-order DMA): bit-identical results, and
// NOT faster. Timing probes (stores / hand-over / DMA / MFMAs switched off one by one) showed the parts ADDING UP instead of
// overlapping — all waves of the one resident block run the same phase at the same time — and the empty loop skeleton costing
// ~1000 cycles per K step (branches, the scalar cursor chain, the barrier), against the 256-512 matrix cycles a 64 x 32 / 64 x 64
// wave tile has per step on the bf16 MFMA; PMC: ~30 instructions per MFMA, waves waiting half of their cycles at the barrier.
// A bf16 K step must carry far more matrix work per instruction and per barrier. So here:
//   * BIG tiles: 256 x 256 (N a multiple of 256), 256 x 128, 256 x 64; eight waves of 128 x 64 (64 x 64, 64 x 32): 32 MFMAs per
//     wave and K step behind 24 fragment reads, half the L2 -> LDS bytes per flop of a 128 x 128 tile (the path that bounds these
//     launches: 2 GB per 1x1 launch of ResNet-50 at 128 x 128, against 0.3-0.6 GB of HBM traffic);
//   * the structure conv_pw.hip proved on the fp32 MFMA: PERSISTENT blocks (one per CU) walk the column tiles of whole row panels;
//     the two-stage LDS ring is carried ACROSS tiles (the last K step of a tile carries the DMA of the next tile's first), operands
//     arrive by `buffer_load ... lds` through wave-uniform descriptors with constant per-lane offsets, the epilogue of tile i is
//     DEFERRED into the first K step of tile i + 1 (its stores fly under a whole step of MFMAs before the next vmcnt(0));
//   * row-panel order: the A rows of a panel are read from HBM once, by one CU, and re-read from its XCD's L2 for the other column
//     tiles; the BM x Nc block of the result is written by one CU within a few tiles.
#include "common.h"
#include "conv_dev.h"

namespace r3m {

typedef unsigned p16_u32x4 __attribute__((ext_vector_type(4)));
typedef float p16_f32x2 __attribute__((ext_vector_type(2)));

constexpr int P16_RSRC_FLAGS = 0x00020000;   // raw buffer, 32-bit offsets, out-of-range lanes read 0 / store nothing

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
__device__ __forceinline__ const char* p16_uniform_ptr(const char* q) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// BM x BN block tile, eight waves WM x WN; wave tile (BM / WM) x (BN / WN) = TM x TN MFMA tiles of 32 x 32.
// EPI: 0 or EPI_STATS (BatchNorm forward statistics from the accumulators).
// FORM 0 (pointwise): A rows are matrix rows (1x1 / stride 1).
// FORM 1 (gather): the A rows are pixels of an NHWC tensor selected per tap ((gy is + dy, gx is + dx), zero outside the image) —
//   forward launches of any k / stride / pad, dgrads of stride-1 layers. K steps: tap-major.
// FORM 3 (window): 3x3 / stride 1 / pad 1 (forward and dgrad). The nine taps of a tile of BM consecutive output pixels read input
//   pixels m0 - (W + 1) .. m0 + BM - 1 + (W + 1): ONE window of BM + 2 W + 2 rows per 64-channel chunk instead of nine BM-row stages
//   (conv_bf16.hip's halo kernel, made persistent: TWO window buffers — the window of the next chunk, or of the next tile's first
//   chunk, arrives under the nine tap steps of the current one — and the weight tiles in a two-stage ring carried across chunks and
//   tiles). K steps: chunk-major. The A fragment of tap (dy, dx) is the window shifted by dy W + dx rows (the swizzle key follows the
//   shifted row); a lane whose pixel has no (y + dy, x + dx) inside the image reads a row of zeros instead. hri = window rows / 8.
template <int BM, int BN, int WM, int WN, int EPI, int FORM>
__global__ __launch_bounds__(512, 2) void pw16_gemm_kernel(const GatherGemmParams p, const int gridM, const int gridN, const int hri) {
  constexpr int NW = 8;
  constexpr bool GATHER = FORM == 1, WIN = FORM == 3;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN == NW && TM >= 1 && TN >= 1 && TM * 32 * WM == BM && TN * 32 * WN == BN, "eight waves of TM x TN MFMA tiles");
  constexpr int STAGE = (BM + BN) * 128;                // bytes per ring stage: {A[BM][64], B[BN][64]} bf16, 128-byte rows
  constexpr int BSTG = BN * 128;                        // bytes of the B part
  constexpr int AJ = BM / 64, BJ = BN / 64;             // DMA instructions (8 rows each) per wave and stage
  static_assert(AJ >= 1 && BJ >= 1, "every wave stages whole DMA instructions of both operands");
  constexpr int SR = BN >= 128 ? 128 : 256;             // result rows per BatchNorm statistics row (gather_gemm_grid_m: tile-independent)
  constexpr int R = BM / SR;                            // statistics rows per tile; each sums WM / R wave rows
  static_assert(R >= 1 && WM % R == 0 && (WM / R) * TM * 32 == SR, "wave rows nest in statistics rows");
  constexpr int CW = TN * 32;                           // columns of a wave tile
  constexpr int CWP = CW + 8;                           // slab row pitch in bf16 (16-byte skew)
  constexpr int SLAB = 8 * CWP * 2;                     // bytes of one wave's epilogue slab: 8 rows
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // LDS: pointwise / gather: [stage 0 {A, B}][stage 1][slabs][statistics scratch]
  //      window: [window 0][window 1][B stage 0][B stage 1][1 KiB holding a zero row][slabs][statistics scratch]
  const int winb = WIN ? hri * 1024 : 0;                // bytes of one window buffer
  unsigned char* const bring = WIN ? smem + 2 * winb : smem + BM * 128;          // B part of ring slot 0
  constexpr int BSTRIDE = WIN ? BSTG : STAGE;           // ... to ring slot 1
  const int zoff = 2 * winb + 2 * BSTG;                 // window form: LDS offset of the zero row
  unsigned char* const slabs = WIN ? smem + zoff + 1024 : smem + 2 * STAGE;
  float* const red = reinterpret_cast<float*>(slabs + NW * SLAB);               // [WM][2][BN]: statistics of the wave rows
  __shared__ int tk[4];                                 // panel tickets q, q + 1, q + 2 of this block (slot q & 3)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_s / WN, wn = wave_s % WN;
  const int lrow = lane & 31, lh = lane >> 5;
  const int srow = lane >> 3, pslot = lane & 7;
  const int W = gridDim.x;
  const int K = p.Ci, Nc = p.Nc, Kb = K * 2;
  const int kpt = K >> 6;                               // 64-channel chunks
  const int nsteps = (FORM == 0 ? 1 : p.ntaps) * kpt;   // K steps per tile
  const bool dyn = p.tile_ctr != nullptr;
  const int xq = blockIdx.x & 7;
  const int id0 = xcd_remap(blockIdx.x, W);
  const char* const Ab = reinterpret_cast<const char*>(p.A);
  const char* const Bb = reinterpret_cast<const char*>(p.B);
  const int KbB = (FORM == 0 ? 1 : p.T) * Kb;           // bytes of one weight row: [tap][Ci]
  if (WIN && tid < 8) *reinterpret_cast<uint4*>(smem + zoff + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // ---- DMA: wave w stages rows [w BM/8, +BM/8) of A and [w BN/8, +BN/8) of B, 8 rows (1 KiB) per instruction; the 16-byte slot a
  // lane fetches is XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128 fragments)
  unsigned voffA[WIN ? 1 : AJ], voffB[BJ];
  if constexpr (!WIN) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int r = wave_s * (BM / NW) + j * 8 + srow;
      voffA[j] = (unsigned)(r * Kb + ((pslot ^ ((r >> 1) & 7)) << 4));          // GATHER: recomputed per tile and tap
    }
  }
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int r = wave_s * (BN / NW) + j * 8 + srow;
    voffB[j] = (unsigned)(r * KbB + ((pslot ^ ((r >> 1) & 7)) << 4));
  }
  const char* dA = Ab;
  const char* dB = Bb;
  int dAbytes = 0, dsA = 0, dsB = 0, tap_soffB = 0;
  unsigned poff[GATHER ? AJ : 1], iyx[GATHER ? AJ : 1];
  auto aim_tile = [&](int tmt_, int tnt_) __attribute__((always_inline)) {
    const int tmt = __builtin_amdgcn_readfirstlane(tmt_), tnt = __builtin_amdgcn_readfirstlane(tnt_);
    dB = Bb + (long long)tnt * BN * KbB;
    if constexpr (FORM == 0) {
      dA = Ab + (long long)tmt * BM * Kb;
      dAbytes = min(BM, p.M - tmt * BM) * Kb;
    } else if constexpr (GATHER) {
      const int hw = p.Hg * p.Wg;
      const int m0t = tmt * BM;
      const int nf = __builtin_amdgcn_readfirstlane(m0t / hw);     // first frame of the tile: 32-bit offsets are relative to it
      const long long frame = (long long)p.Hi * p.Wi * Kb;
      dA = p16_uniform_ptr(Ab + nf * frame);
      const long long rest = (long long)(p.N - nf) * frame;
      dAbytes = __builtin_amdgcn_readfirstlane(rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
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
        poff[j] = (unsigned)((((n - nf) * p.Hi + iy0) * p.Wi + ix0) * Kb + ((pslot ^ ((r >> 1) & 7)) << 4));
        iyx[j] = m < p.M ? (unsigned)((iy0 << 16) | ix0) : 0x40004000u;
        m += 8;                                             // the lane's next row is 8 GEMM rows further: branch-free carries
        gx += 8;                                            // (Wg >= 4: at most two row wraps; Hg >= 2: at most two frame wraps — pw16_form)
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
  auto set_tap = [&](int t) __attribute__((always_inline)) {   // weight tap of the step being staged (+ GATHER: the A row offsets of that tap)
    if constexpr (FORM != 0) {
      const int pack = __builtin_amdgcn_readfirstlane(p.tap[t]);
      tap_soffB = (pack >> 16) * Kb;
      if constexpr (GATHER) {
        const int dy = (pack << 24) >> 24, dx = (pack << 16) >> 24;
        const int delta = (dy * p.Wi + dx) * Kb;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const unsigned iy = (iyx[j] >> 16) + (unsigned)dy, ix = (iyx[j] & 0xFFFFu) + (unsigned)dx;
          voffA[j] = (iy < (unsigned)p.Hi && ix < (unsigned)p.Wi) ? poff[j] + (unsigned)delta : BUF_OOB;
        }
      }
    }
  };
  auto aim_step = [&](int chunk) __attribute__((always_inline)) {
    dsA = __builtin_amdgcn_readfirstlane(chunk * 128);
    dsB = __builtin_amdgcn_readfirstlane(tap_soffB + chunk * 128);
  };
  auto dma_step = [&](int slot) __attribute__((always_inline)) {   // the ring pieces of one K step: A (pointwise / gather) and B
    if constexpr (!WIN) {
      unsigned char* const la = smem + slot * STAGE + wave_s * (BM / NW) * 128;
      static_for<AJ>([&](auto j_c) __attribute__((always_inline)) {
        constexpr int j = decltype(j_c)::value;
        if (R3M_PROBE(p) & 4) return;                     // timing probes (probe builds; wrong results): 4 no A DMA, 8 no B DMA
        buf_dma16_uniform(dA, dAbytes, la + j * 1024, voffA[j], dsA);
      });
    }
    unsigned char* const lb = bring + slot * BSTRIDE + wave_s * (BN / NW) * 128;
    static_for<BJ>([&](auto j_c) __attribute__((always_inline)) {
      constexpr int j = decltype(j_c)::value;
      if (R3M_PROBE(p) & 8) return;
      buf_dma16_uniform(dB, BN * KbB, lb + j * 1024, voffB[j], dsB);
    });
  };
  // window form: the window of (row panel tmt, 64-channel chunk) into window buffer `buf`: DMA instruction i covers window rows
  // 8 i .. 8 i + 7 = pixels hb + 8 i ..; rows before the tensor get an out-of-range offset, rows past it fall off the descriptor
  auto dma_win = [&](int buf, int tmt_, int chunk) __attribute__((always_inline)) {
    if constexpr (WIN) {
      const int tmt = __builtin_amdgcn_readfirstlane(tmt_);
      const long long hb = (long long)tmt * BM - (p.Wi + 1);          // pixel of window row 0 (negative in the first tile)
      const char* const wb = p16_uniform_ptr(Ab + hb * Kb);           // (may point in front of the tensor: those lanes are masked)
      const long long rest = ((long long)p.M - hb) * Kb;
      const int wbytes = __builtin_amdgcn_readfirstlane(rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
      unsigned char* const lw = smem + buf * winb;
      if (R3M_PROBE(p) & 4) return;
      for (int i = wave_s; i < hri; i += NW) {
        const int hr = 8 * i + srow;
        const unsigned vo = (hb + hr >= 0) ? (unsigned)(hr * Kb + ((pslot ^ ((hr >> 1) & 7)) << 4)) : BUF_OOB;
        buf_dma16_uniform(wb, wbytes, lw + i * 1024, vo, chunk * 128);
      }
    }
  };

  // ---- fragments: lane half h reads k = 16 g + 8 h .. + 7 of group g (one ds_read_b128 = the 32x32x16 operand of a row)
  unsigned fa[4], fb[4];                                  // LDS byte offsets inside the A / B part of a ring slot
  {
    const int xr = (lrow >> 1) & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int go = ((2 * g + lh) ^ xr) * 16;
      fa[g] = (unsigned)((wm * TM * 32 + lrow) * 128 + go);
      fb[g] = (unsigned)((wn * TN * 32 + lrow) * 128 + go);
    }
  }
  // window form, per lane and A row tile: window row of the centre tap (constant) and the validity of the nine taps for the current
  // tile's pixel (bit k = tap p.tap[k]); scalar: which taps look up / down / left / right
  int crow[WIN ? TM : 1];
  unsigned vmask[WIN ? TM : 1];
  unsigned m_up = 0u, m_dn = 0u, m_lf = 0u, m_rt = 0u;
  if constexpr (WIN) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      crow[t] = wm * TM * 32 + t * 32 + lrow + p.Wi + 1;
      vmask[t] = 0u;
    }
    for (int k = 0; k < 9; ++k) {
      const int pk = p.tap[k];
      const int dy = (pk << 24) >> 24, dx = (pk << 16) >> 24;
      m_up |= (dy < 0 ? 1u : 0u) << k;
      m_dn |= (dy > 0 ? 1u : 0u) << k;
      m_lf |= (dx < 0 ? 1u : 0u) << k;
      m_rt |= (dx > 0 ? 1u : 0u) << k;
    }
  }
  auto win_rows = [&](int tmt) __attribute__((always_inline)) {
    if constexpr (WIN) {
      const int hw = p.Hi * p.Wi;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int m = tmt * BM + wm * TM * 32 + t * 32 + lrow;
        const int rem = m % hw;
        const int y = rem / p.Wi, x = rem - y * p.Wi;
        const unsigned bad = (y == 0 ? m_up : 0u) | (y == p.Hi - 1 ? m_dn : 0u) | (x == 0 ? m_lf : 0u) | (x == p.Wi - 1 ? m_rt : 0u);
        vmask[t] = m < p.M ? (0x1FFu & ~bad) : 0u;        // rows >= M read zeros for every tap (the statistics rely on that)
      }
    }
  };

  f32x16 acc[TM][TN];
  // one K step: fragments of group g + 1 are requested before the MFMAs of group g (two register sets).
  // sa: A part of the ring slot (pointwise / gather) or the window buffer (window form); sb: B part of the ring slot; tap: window form
  auto kstep = [&](const unsigned char* sa, const unsigned char* sb, int tap, auto first_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value;
    bf16x8 a[2][TM], b[2][TN];
    unsigned abase[WIN ? TM : 1], akey[WIN ? TM : 1];
    if constexpr (WIN) {
      const int pk = __builtin_amdgcn_readfirstlane(p.tap[tap]);
      const int shift = ((pk << 24) >> 24) * p.Wi + ((pk << 16) >> 24);
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const bool ok = ((vmask[t] >> tap) & 1u) != 0u;
        const int rv = crow[t] + shift;
        abase[t] = ok ? (unsigned)(sa - smem) + (unsigned)(rv * 128) : (unsigned)zoff;
        akey[t] = ok ? (unsigned)((rv >> 1) & 7) : 0u;
      }
    }
    auto frag_load = [&](auto g_c) __attribute__((always_inline)) {
      constexpr int g = decltype(g_c)::value;
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if constexpr (WIN) a[g & 1][t] = *reinterpret_cast<const bf16x8*>(smem + abase[t] + (((unsigned)(2 * g + lh) ^ akey[t]) << 4));
        else a[g & 1][t] = *reinterpret_cast<const bf16x8*>(sa + fa[g] + t * 32 * 128);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) b[g & 1][t] = *reinterpret_cast<const bf16x8*>(sb + fb[g] + t * 32 * 128);
    };
    frag_load(std::integral_constant<int, 0>{});
    static_for<4>([&](auto g_c) __attribute__((always_inline)) {
      constexpr int g = decltype(g_c)::value;
      if constexpr (g < 3) frag_load(std::integral_constant<int, g + 1>{});
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if constexpr (FIRST && g == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tm], b[0][tn], z, 0, 0, 0);
          } else {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g & 1][tm], b[g & 1][tn], acc[tm][tn], 0, 0, 0);
          }
        }
    });
  };

  // ---- epilogue, part 1 of tile (emt, ent): BatchNorm statistics in-lane from the accumulators (a lane owns a column; packed fp32
  // adds / fmas over row pairs), then the results: per 8-row chunk the wave rounds to bf16 into its private LDS slab ([8][CW] bf16),
  // reads it back one 16-byte row segment per lane and stores with ONE buffer_store_dwordx4 whose row offset is a scalar. A wave's
  // LDS accesses execute in order: the slab needs no barrier.
  // acc[tm][tn][r] is tile row wm TM 32 + tm 32 + 8 (r >> 2) + 4 lh + (r & 3), tile column wn CW + tn 32 + lrow.
  unsigned char* const slab = slabs + wave_s * SLAB;
  bf16_t* const slab_w = reinterpret_cast<bf16_t*>(slab) + (4 * lh) * CWP + lrow;          // + e CWP + tn 32: row 4 lh + e
  constexpr int LPR = CW / 8, RPS = 64 / LPR;             // lanes per row of a store, rows per store
  static_assert(RPS == 8, "64-column wave tiles: one store = one 8-row chunk");
  const int e_row = lane / LPR, e_col = (lane % LPR) * 8;
  const unsigned char* const slab_r = slab + ((e_row & 7) * CWP + e_col) * 2;
  const unsigned vo_st = (unsigned)(((wm * TM * 32 + e_row) * Nc + wn * CW + e_col) * 2);  // per-lane byte offset inside the tile (constant)
  auto epilogue1 = [&](int emt, int ent) __attribute__((always_inline)) {
    const int m0 = emt * BM, n0 = ent * BN;
    if constexpr ((EPI & EPI_STATS) != 0) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        p16_f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const p16_f32x2 v = {acc[tm][tn][r], acc[tm][tn][r + 1]};
            s2 += v;
            q2 = __builtin_elementwise_fma(v, v, q2);
          }
        float s = s2[0] + s2[1], ss = q2[0] + q2[1];
        s += __shfl_xor(s, 32);
        ss += __shfl_xor(ss, 32);
        if (lane < 32) {
          const int c = wn * CW + tn * 32 + lane;
          red[(wm * 2 + 0) * BN + c] = s;
          red[(wm * 2 + 1) * BN + c] = ss;
        }
      }
    }
    if (R3M_PROBE(p) & 2) return;                         // probe 2: statistics only
    const int rows_valid = min(BM, p.M - m0);
    char* const ob = reinterpret_cast<char*>(p.out) + ((long long)m0 * Nc + n0) * 2;
    const int obytes = ((rows_valid - 1) * Nc + BN) * 2;  // a lane's offset is inside iff its row is < rows_valid
    int ncb = Nc * 2;
    asm volatile("" : "+s"(ncb));                         // row offsets computed at the point of use, not hoisted into scarce SGPRs
    static_for<TM * 4>([&](auto c_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value;             // chunk (tm, q): wave-tile rows 8 c .. 8 c + 7
      constexpr int tm = c >> 2, q = c & 3;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int e = 0; e < 4; ++e) slab_w[e * CWP + tn * 32] = (bf16_t)acc[tm][tn][4 * q + e];
      __builtin_amdgcn_wave_barrier();
      const p16_u32x4 v = *reinterpret_cast<const p16_u32x4*>(slab_r);
      if (!(R3M_PROBE(p) & 1)) p16_st4(ob, obytes, vo_st, (c * 8) * ncb, v);
      __builtin_amdgcn_wave_barrier();                    // the chunk's slab read is issued before the next chunk's writes
    });
  };
  // part 2 (EPI_STATS, one barrier after part 1): combine the wave rows of each statistics row — stats[mt R + h][2][Nc]
  auto epilogue2 = [&](int emt, int ent) __attribute__((always_inline)) {
    if constexpr ((EPI & EPI_STATS) != 0) {
      for (int t = tid; t < BN; t += NW * 64) {
        const int col = ent * BN + t;
#pragma unroll
        for (int h = 0; h < R; ++h) {
          float s = 0.f, ss = 0.f;
#pragma unroll
          for (int w = h * (WM / R); w < (h + 1) * (WM / R); ++w) {
            s += red[(w * 2 + 0) * BN + t];
            ss += red[(w * 2 + 1) * BN + t];
          }
          const long long prow = (long long)emt * R + h;
          if (prow * SR < p.M) {
            p.stats[(prow * 2 + 0) * Nc + col] = s;
            p.stats[(prow * 2 + 1) * Nc + col] = ss;
          }
        }
      }
    }
  };

  // ---- persistent walk: a block owns whole row panels (static: id0 + q W; engine launches: the k-th ticket of per-XCD queue xq is
  // panel 8 k + xq) and walks their gridN column tiles back to back. Tickets are requested two panels ahead by one lane at the first
  // step of a panel (the K loop waits for vmcnt(0) at every step anyway) and handed over through LDS before the next barrier.
  unsigned* const ctr = dyn ? p.tile_ctr + xq : nullptr;
  if (dyn && tid == 0) {
    tk[0] = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk[1] = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  int q = 0, mt = dyn ? __builtin_amdgcn_readfirstlane(tk[0]) * 8 + xq : id0, nt = 0;
  bool has = mt < gridM;
  unsigned tkt = 0u;
  bool tkt_pending = false;
  int tkt_slot = 0;
  if (has) {
    aim_tile(mt, nt);
    set_tap(0);
    aim_step(0);
    dma_win(0, mt, 0);
    dma_step(0);
  }
  int slot = 0, wbuf = 0;                                 // ring slot / window buffer the current step reads
  bool pend = false, red_pend = false;
  int pmt = 0, pnt = 0, rmt = 0, rnt = 0;
  while (has) {
    int nq = q, nmt = mt, nnt = nt + 1;
    bool nhas = true;
    // (tap, chunk) of the current step and of the step whose operands go out next. Gather: tap-major; window: chunk-major.
    int ct = 0, cc = 0, ti = 0, ch = 0;
    for (int s = 0; s < nsteps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this step's operands have landed (and the previous tile's stores are out)
      if (tkt_pending) {
        if (tid == 0) tk[tkt_slot] = (int)tkt;
        tkt_pending = false;
      }
      __syncthreads();
      if (s == 0) {
        if (nnt == gridN) {                               // next tile: first column tile of the next panel
          nnt = 0;
          nq = q + 1;
          nmt = dyn ? __builtin_amdgcn_readfirstlane(tk[nq & 3]) * 8 + xq : mt + W;
          nhas = nmt < gridM;
        }
        win_rows(mt);
      }
      // the DMA of the following step goes out first: the rest of this step is its flight time
      if (s + 1 < nsteps) {
        if constexpr (WIN) {
          if (++ti == 9) {
            ti = 0;
            ++ch;
          }
          set_tap(ti);
        } else if (++ch == kpt) {
          ch = 0;
          ++ti;
          set_tap(ti);
        }
        aim_step(ch);
        dma_step(slot ^ 1);
      } else if (nhas) {
        aim_tile(nmt, nnt);
        set_tap(0);
        aim_step(0);
        dma_step(slot ^ 1);
      }
      if (WIN && ct == 0) {                               // first tap of a chunk: the next window goes out (nine steps of flight time)
        if (cc + 1 < kpt) dma_win(wbuf ^ 1, mt, cc + 1);
        else if (nhas) dma_win(wbuf ^ 1, nmt, 0);
      }
      if (s == 0 && dyn && nt == 0) {                     // first step of a panel: request ticket q + 2
        if (tid == 0) tkt = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tkt_pending = true;
        tkt_slot = (q + 2) & 3;
      }
      if (red_pend) {
        epilogue2(rmt, rnt);
        red_pend = false;
        // single-step tiles: part 1 of the NEXT tile follows in this very step and rewrites the scratch part 2 has just read
        if ((EPI & EPI_STATS) != 0 && nsteps == 1) __syncthreads();
      }
      if (s == 0 && pend) {
        epilogue1(pmt, pnt);
        pend = false;
        if constexpr ((EPI & EPI_STATS) != 0) {
          red_pend = true;
          rmt = pmt;
          rnt = pnt;
        }
      }
      const unsigned char* const sa = WIN ? smem + wbuf * winb : smem + slot * STAGE;
      const unsigned char* const sb = bring + slot * BSTRIDE;
      if (!(R3M_PROBE(p) & 16)) {                         // probe 16: no fragment reads, no MFMAs
        if (s == 0) kstep(sa, sb, ct, std::true_type{});
        else kstep(sa, sb, ct, std::false_type{});
      }
      slot ^= 1;
      if constexpr (WIN) {
        if (++ct == 9) {
          ct = 0;
          ++cc;
          wbuf ^= 1;
        }
      }
    }
    pend = true;
    pmt = mt;
    pnt = nt;
    q = nq;
    mt = nmt;
    nt = nnt;
    has = nhas;
  }
  // drain: the last tile's results, and the statistics rows still in LDS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (red_pend) {
    __syncthreads();
    epilogue2(rmt, rnt);
  }
  if (pend) {
    __syncthreads();                                      // (the statistics scratch is free again)
    epilogue1(pmt, pnt);
    if constexpr ((EPI & EPI_STATS) != 0) {
      __syncthreads();
      epilogue2(pmt, pnt);
    }
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

// window form: LDS bytes of tile (BM, BN) at image width W (two windows, two weight stages, zero row, slabs, statistics scratch)
static inline int p16_win_lds(int BM, int BN, int WMv, int W) {
  const int hri = ceil_div(BM + 2 * W + 2, 8);
  return 2 * hri * 1024 + 2 * BN * 128 + 1024 + 8 * (8 * (64 + 8) * 2) + WMv * 2 * BN * 4 + 64;
}

// 0: not for this kernel; 1: pointwise form (1x1 / stride 1: A rows are matrix rows); 2: gather form (dense OUTPUT rows: forward of
// any geometry, dgrad of stride-1 layers); 3: window form (3x3 / stride 1 / pad 1). A pure function of the launch parameters
// (tests/test_dispatch.py).
int pw16_form(const GatherGemmParams& p) {
  if (!g_pw16_mode || p.dtype != DT_BF16) return 0;
  if ((p.Ci & 63) || p.Ci > 4096 || (p.Nc & 63) || p.M < 1) return 0;
  if (p.flags != 0 && p.flags != EPI_STATS) return 0;
  const bool dense_out = p.os == 1 && p.ooy == 0 && p.oox == 0 && p.Hg == p.Ho && p.Wg == p.Wo;
  if (!dense_out) return 0;
  if (p.simple_rows && p.ntaps == 1 && p.T == 1 && p.dy[0] == 0 && p.dx[0] == 0 && p.wt[0] == 0) return 1;
  if (p.simple_rows || p.ntaps < 1 || p.ntaps > MAX_TAPS) return 0;
  // 32-bit offsets: a tile's rows span at most ceil(512 / (Hg Wg)) + 1 frames of the input; one weight tile [256][T][Ci]
  if (p.Hi >= 16384 || p.Wi >= 16384 || p.Hi < 1 || p.Wi < 1 || p.Wg < 4 || p.Hg < 2) return 0;
  const long long frame = (long long)p.Hi * p.Wi * p.Ci * 2;
  const long long span = (512 / ((long long)p.Hg * p.Wg) + 2) * frame;
  if (span >= (long long)BUF_OOB || 256LL * p.T * p.Ci * 2 >= (long long)BUF_OOB) return 0;
  // 3x3 / stride 1 / pad 1 with every tap inside [-1, 1]^2: the window form, if the whole tensor is addressable through one
  // 32-bit descriptor and the windows fit the LDS
  if ((g_pw16_mode & 2) && p.ntaps == 9 && p.is == 1 && p.Hg == p.Hi && p.Wg == p.Wi && p.Ho == p.Hi && p.Wo == p.Wi) {
    bool ok = true;
    for (int k = 0; k < 9; ++k) ok = ok && p.dy[k] >= -1 && p.dy[k] <= 1 && p.dx[k] >= -1 && p.dx[k] <= 1;
    const bool n256 = (p.Nc & 255) == 0, n128 = (p.Nc & 127) == 0;
    const int lds = n256 ? p16_win_lds(256, 256, 2, p.Wi) : n128 ? p16_win_lds(256, 128, 4, p.Wi) : p16_win_lds(256, 64, 8, p.Wi);
    if (ok && lds <= 160 * 1024 && ((long long)p.M + 2 * p.Wi + 2) * p.Ci * 2 < (long long)BUF_OOB) return 3;
  }
  return 2;
}

template <int BM, int BN, int WM, int WN, int EPI, int FORM>
static int launch_pw16_one(const GatherGemmParams& p, int W, int gridM, int gridN, hipStream_t s) {
  constexpr int TN = BN / WN / 32;
  static_assert(TN == 2, "64-column wave tiles");
  const int hri = FORM == 3 ? ceil_div(BM + 2 * p.Wi + 2, 8) : 0;
  const int lds = FORM == 3 ? p16_win_lds(BM, BN, WM, p.Wi)
                            : 2 * (BM + BN) * 128 + 8 * (8 * (TN * 32 + 8) * 2) + WM * 2 * BN * 4 + 64;   // ring + slabs + statistics scratch
  R3M_REQUIRE(lds <= 160 * 1024, "pw16_gemm: %d bytes of LDS", lds);
  auto kern = pw16_gemm_kernel<BM, BN, WM, WN, EPI, FORM>;
  static DynLdsOptIn oi;
  if (int e = ensure_dyn_lds(oi, reinterpret_cast<const void*>(kern), lds, "pw16_gemm")) return e;
  hipLaunchKernelGGL(kern, dim3(W), dim3(512), lds, s, p, gridM, gridN, hri);
  return 0;
}

template <int BM, int BN, int WM, int WN, int FORM>
static int launch_pw16_shape(const GatherGemmParams& p_in, hipStream_t s) {
  const int gridM = ceil_div(p_in.M, BM), gridN = p_in.Nc / BN;
  const int slots = p16_cu_count();                                     // one eight-wave block per CU
  const int W = gridM < slots ? gridM : slots;
  GatherGemmParams p = p_in;
  if (W < 64 || gridM < 64) p.tile_ctr = nullptr;                       // small launches: every queue needs blocks AND panels; static split
  switch (p.flags) {
    case 0: return launch_pw16_one<BM, BN, WM, WN, 0, FORM>(p, W, gridM, gridN, s);
    case EPI_STATS: return launch_pw16_one<BM, BN, WM, WN, EPI_STATS, FORM>(p, W, gridM, gridN, s);
  }
  set_last_error("pw16_gemm: unsupported epilogue flag combination %d", p.flags);
  return 1;
}

int launch_pw16(const GatherGemmParams& p, hipStream_t s) {
  const int form = pw16_form(p);
  R3M_REQUIRE(form != 0, "pw16_gemm: launch not eligible");
  // 256 x 256 (waves 128 x 64), 256 x 128 (64 x 64), 512 x 64 (64 x 64; window form: 256 x 64, waves 32 x 64): every wave tile is 64
  // columns wide
  const bool n256 = (p.Nc & 255) == 0, n128 = (p.Nc & 127) == 0;
  if (form == 3) return n256 ? launch_pw16_shape<256, 256, 2, 4, 3>(p, s) : n128 ? launch_pw16_shape<256, 128, 4, 2, 3>(p, s) : launch_pw16_shape<256, 64, 8, 1, 3>(p, s);
  if (form == 2) return n256 ? launch_pw16_shape<256, 256, 2, 4, 1>(p, s) : n128 ? launch_pw16_shape<256, 128, 4, 2, 1>(p, s) : launch_pw16_shape<512, 64, 8, 1, 1>(p, s);
  return n256 ? launch_pw16_shape<256, 256, 2, 4, 0>(p, s) : n128 ? launch_pw16_shape<256, 128, 4, 2, 0>(p, s) : launch_pw16_shape<512, 64, 8, 1, 0>(p, s);
}

}  // namespace r3m
