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

// Round 6: an experiment record, not product code — compiled into probe builds (-DR3M_PROBES) only; the shipped
// library gets the inline stubs of common.h and never routes here.
#ifdef R3M_PROBES
namespace r3m {

typedef unsigned p16_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned p16_u32x2 __attribute__((ext_vector_type(2)));
typedef float p16_f32x2 __attribute__((ext_vector_type(2)));
typedef short p16_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 p16_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned p16_pack(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(p16_f32x2{lo, hi}, p16_bf16x2));
}

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

// all LDS traffic of this wave has completed, then the workgroup barrier. NOT __syncthreads(): its fences drain vmcnt as well, which is
// exactly what a loader wave's counted DMA wait must not do. The "memory" clobber keeps the compiler from moving LDS accesses across.
__device__ __forceinline__ void p16_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// BM x BN block tile. EIGHT compute waves WM x WN (wave tile TM x TN MFMA tiles of 32 x 32, 64 columns wide) + FOUR loader waves.
//   loader waves: own the tile / tap / chunk cursor and issue ALL LDS DMA, NS - 1 K steps ahead of the multiplication, whatever tile
//     those steps belong to. They issue nothing else, so their vmcnt counts only DMA pieces, which retire in order: the wait for the
//     oldest ring slot is an exact `s_waitcnt vmcnt((NS - 2) pieces)` and (NS - 1) slots stay in flight per CU all the time.
//   compute waves: fragment reads, MFMAs, and at the end of a tile the BatchNorm statistics and the result stores through a private
//     LDS slab. Their vmcnt only ever counts their own stores and they never wait for it: the stores of tile i fly under tile i + 1.
//   one s_barrier per K step: "slot s has landed" (loaders) meets "slot s - 1 is free again" (compute).
// NS ring slots. EPI: 0 or EPI_STATS.
// FORM 0 (pointwise): A rows are matrix rows (1x1 / stride 1). FORM 1 (gather): A rows are pixels of an NHWC tensor selected per tap
//   ((gy is + dy, gx is + dx), zero outside the image); K steps tap-major. FORM 3 (window): 3x3 / stride 1 / pad 1: the nine taps of a
//   tile of BM consecutive output pixels read ONE window of BM + 2 W + 2 input rows per 64-channel chunk (two window buffers: the
//   next chunk's / next tile's window arrives under the nine tap steps of the current one), the ring holds weight tiles only; K steps
//   chunk-major; the A fragment of tap (dy, dx) is the window shifted by dy W + dx rows (the swizzle key follows the shifted row), a
//   lane whose pixel has no (y + dy, x + dx) inside the image reads a row of zeros. hri = window rows / 8.
template <int BM, int BN, int WM, int WN, int NS, int EPI, int FORM>
__global__ __launch_bounds__(768, 3) void pw16_gemm_kernel(const GatherGemmParams p, const int gridM, const int gridN, const int hri) {
  constexpr int NCW = 8, NLW = 4;
  constexpr bool GATHER = FORM == 1, WIN = FORM == 3;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN == NCW && TM >= 1 && TN == 2 && TM * 32 * WM == BM && TN * 32 * WN == BN, "eight compute waves, 64-column wave tiles");
  static_assert(NS == 2 || NS == 3, "ring of two or three slots");
  constexpr int STAGE = (BM + BN) * 128;                // bytes per ring slot: {A[BM][64], B[BN][64]} bf16, 128-byte rows
  constexpr int BSTG = BN * 128;                        // bytes of the B part
  constexpr int AJ = BM / (8 * NLW), BJ = BN / (8 * NLW);   // DMA instructions (8 rows each) per loader wave and K step
  static_assert(AJ >= 1 && BJ >= 1, "every loader wave stages whole DMA instructions of both operands");
  constexpr int PL = (WIN ? 0 : AJ) + BJ;               // ring pieces per loader wave and K step
  static_assert(PL * (NS - 2) <= 63, "vmcnt is a 6-bit counter");
  constexpr int SR = BN >= 128 ? 128 : 256;             // result rows per BatchNorm statistics row (gather_gemm_grid_m: tile-independent)
  constexpr int R = BM / SR;                            // statistics rows per tile; each sums WM / R wave rows
  static_assert(R >= 1 && WM % R == 0 && (WM / R) * TM * 32 == SR, "wave rows nest in statistics rows");
  constexpr int CW = TN * 32;                           // columns of a wave tile
  constexpr int SLAB = 1280;                            // bytes of one compute wave's epilogue slab: [64 columns][8 rows] bf16 + column skew
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // LDS: pointwise / gather: [NS slots {A, B}][slabs][statistics scratch]
  //      window: [window 0][window 1][NS weight slots][1 KiB holding a zero row][slabs][statistics scratch]
  const int winb = WIN ? hri * 1024 : 0;                // bytes of one window buffer
  unsigned char* const bring = WIN ? smem + 2 * winb : smem + BM * 128;          // B part of ring slot 0
  constexpr int BSTRIDE = WIN ? BSTG : STAGE;           // ... to the next slot
  const int zoff = 2 * winb + NS * BSTG;                // window form: LDS offset of the zero row
  unsigned char* const slabs = WIN ? smem + zoff + 1024 : smem + NS * STAGE;
  float* const red = reinterpret_cast<float*>(slabs + NCW * SLAB);              // [WM][2][BN]: statistics of the wave rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = gridDim.x;
  const int K = p.Ci, Nc = p.Nc, Kb = K * 2;
  const int kpt = K >> 6;                               // 64-channel chunks
  const int nsteps = (FORM == 0 ? 1 : p.ntaps) * kpt;   // K steps per tile
  if (WIN && tid < 16) *reinterpret_cast<uint4*>(smem + zoff + tid * 16) = make_uint4(0u, 0u, 0u, 0u);   // 256 zero bytes
  // tile walk (both roles): a block owns row panels id0, id0 + W, ... (workers of one XCD hold neighbouring panels) and walks their
  // gridN column tiles back to back — the A rows of a panel come from HBM once and from the XCD's L2 for the other column tiles, and
  // the BM x Nc block of the result is written by one CU within a few tiles
  int mt = xcd_remap(blockIdx.x, W), nt = 0;
  bool has = mt < gridM;
  auto next_tile = [&]() __attribute__((always_inline)) {
    if (++nt == gridN) {
      nt = 0;
      mt += W;
      has = mt < gridM;
    }
  };

  if (wave_s >= NCW) {
    // ================================================ loader waves ================================================
    const int lw = wave_s - NCW;
    const int srow = lane >> 3, pslot = lane & 7;
    const char* const Ab = reinterpret_cast<const char*>(p.A);
    const char* const Bb = reinterpret_cast<const char*>(p.B);
    const int KbB = (FORM == 0 ? 1 : p.T) * Kb;         // bytes of one weight row: [tap][Ci]
    // wave l stages rows [l BM/4, +BM/4) of A and [l BN/4, +BN/4) of B, 8 rows (1 KiB) per instruction; the 16-byte slot a lane
    // fetches is XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128 fragments)
    unsigned voffA[WIN ? 1 : AJ], voffB[BJ];
    if constexpr (!WIN) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int r = lw * (BM / NLW) + j * 8 + srow;
        voffA[j] = (unsigned)(r * Kb + ((pslot ^ ((r >> 1) & 7)) << 4));        // GATHER: recomputed per tile and tap
      }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int r = lw * (BN / NLW) + j * 8 + srow;
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
        const int nf = __builtin_amdgcn_readfirstlane(m0t / hw);   // first frame of the tile: 32-bit offsets are relative to it
        const long long frame = (long long)p.Hi * p.Wi * Kb;
        dA = p16_uniform_ptr(Ab + nf * frame);
        const long long rest = (long long)(p.N - nf) * frame;
        dAbytes = __builtin_amdgcn_readfirstlane(rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
        const int r0 = lw * (BM / NLW) + srow;
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
    auto dma_step = [&](int slot) __attribute__((always_inline)) {   // the ring pieces of one K step
      if constexpr (!WIN) {
        unsigned char* const la = smem + slot * STAGE + lw * (BM / NLW) * 128;
        static_for<AJ>([&](auto j_c) __attribute__((always_inline)) {
          constexpr int j = decltype(j_c)::value;
          if (R3M_PROBE(p) & 4) return;                   // timing probes (probe builds; wrong results): 4 no A DMA, 8 no B DMA
          buf_dma16_uniform(dA, dAbytes, la + j * 1024, voffA[j], dsA);
        });
      }
      unsigned char* const lb = bring + slot * BSTRIDE + lw * (BN / NLW) * 128;
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
        const long long hb = (long long)tmt * BM - (p.Wi + 1);        // pixel of window row 0 (negative in the first tile)
        const char* const wb = p16_uniform_ptr(Ab + hb * Kb);         // (may point in front of the tensor: those lanes are masked)
        const long long rest = ((long long)p.M - hb) * Kb;
        const int wbytes = __builtin_amdgcn_readfirstlane(rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
        unsigned char* const lwin = smem + buf * winb;
        if (R3M_PROBE(p) & 4) return;
        for (int i = lw; i < hri; i += NLW) {
          const int hr = 8 * i + srow;
          const unsigned vo = (hb + hr >= 0) ? (unsigned)(hr * Kb + ((pslot ^ ((hr >> 1) & 7)) << 4)) : BUF_OOB;
          buf_dma16_uniform(wb, wbytes, lwin + i * 1024, vo, chunk * 128);
        }
      }
    };

    // the DMA cursor: (tile, tap, chunk) of the step whose pieces go out next. Gather: tap-major; window: chunk-major.
    int imt = mt, int_ = nt;
    bool ihas = has;
    int iti = 0, ich = 0;
    auto next_issue_tile = [&]() __attribute__((always_inline)) {
      if (++int_ == gridN) {
        int_ = 0;
        imt += W;
        ihas = imt < gridM;
      }
    };
    // issue the pieces of the cursor's step into ring slot `slot` (+ window form: at the first tap of a chunk the chunk's window was
    // issued one chunk earlier; here the NEXT chunk's / tile's window goes out), then advance the cursor
    int iwbuf = 0;                                        // window buffer the cursor's chunk lives in
    auto issue_next = [&](int slot) __attribute__((always_inline)) {
      dma_step(slot);
      if constexpr (WIN) {
        // The window after this chunk's goes into the buffer the PREVIOUS chunk's window lives in, and the multiplication runs NS - 1
        // steps behind this cursor: at tap NS - 1 the compute waves have passed the barrier into the current chunk, the buffer is free.
        if (iti == NS - 1) {
          if (ich + 1 < kpt) dma_win(iwbuf ^ 1, imt, ich + 1);
          else {
            int nm = imt, nn = int_ + 1;
            bool nh = true;
            if (nn == gridN) {
              nn = 0;
              nm += W;
              nh = nm < gridM;
            }
            if (nh) dma_win(iwbuf ^ 1, nm, 0);
          }
        }
        if (++iti == 9) {
          iti = 0;
          iwbuf ^= 1;
          if (++ich == kpt) {
            ich = 0;
            next_issue_tile();
            if (ihas) aim_tile(imt, int_);
          }
        }
        if (ihas) {
          set_tap(iti);
          aim_step(ich);
        }
      } else {
        if (++ich == kpt) {
          ich = 0;
          if (FORM == 0 || ++iti == p.ntaps) {
            iti = 0;
            next_issue_tile();
            if (ihas) aim_tile(imt, int_);
          }
          if (ihas) set_tap(iti);
        }
        if (ihas) aim_step(ich);
      }
    };
    if (ihas) {
      aim_tile(imt, int_);
      set_tap(0);
      aim_step(0);
      dma_win(0, imt, 0);                                 // the first window (its successors go out with the first tap of each chunk)
    }
    int ahead = 0, islot = 0;                             // K steps issued and not yet multiplied; ring slot of the next issue
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
      if (ihas) {
        issue_next(islot);
        ++ahead;
        islot = islot + 1 == NS ? 0 : islot + 1;
      }
    }
    int cs = 0, done_tiles = 0;
    while (has) {
      // my pieces of the step about to be multiplied have landed; the (NS - 2) steps issued after it may still fly. (Window pieces are
      // older than the ring pieces that follow them: in-order retirement covers them.)
      if (NS == 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      p16_bar();                                          // ... everyone's have; and the compute waves are done with the slot of the previous step
      if ((EPI & EPI_STATS) != 0 && nsteps == 1 && done_tiles > 0) p16_bar();   // (the compute waves' extra barrier of single-step tiles)
      if (ihas) {
        issue_next(islot);
        ++ahead;
        islot = islot + 1 == NS ? 0 : islot + 1;
      }
      --ahead;
      if (++cs == nsteps) {
        cs = 0;
        ++done_tiles;
        next_tile();
      }
    }
    p16_bar();
    return;
  }

  // ================================================== compute waves ==================================================
  const int wm = wave_s / WN, wn = wave_s % WN;
  const int lrow = lane & 31, lh = lane >> 5;
  // fragments: lane half h reads k = 16 g + 8 h .. + 7 of group g (one ds_read_b128 = the 32x32x16 operand of a row)
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
        // a masked lane reads its zeros at the position of the 256-byte bank window its real row would have used (conv_bf16.hip)
        abase[t] = ok ? (unsigned)(sa - smem) + (unsigned)(rv * 128) : (unsigned)(zoff + (rv & 1) * 128);
        akey[t] = (unsigned)((rv >> 1) & 7);
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

  // ---- end of a tile, part 1: BatchNorm statistics in-lane from the accumulators (a lane owns a column; packed fp32 adds / fmas over
  // row pairs), then the results, 8 rows of the wave tile at a time, through the wave's private LDS slab — TRANSPOSED BY THE LDS:
  //   write: a lane's four accumulators of a chunk are four consecutive ROWS of its column -> two v_cvt_pk + ONE ds_write_b64 per
  //          column tile into a column-major slab [64 columns][8 rows] bf16;
  //   read:  ds_read_b64_tr_b16 (gfx950's transpose read: in each 16-lane group lane c receives element c & 3 of the 8 bytes that
  //          lanes 4 j + (c >> 2), j = 0..3, addressed) — with lane 4 j + i addressing rows R0..R0+3 of column cb + 8 i + j, lane c
  //          receives row R0 + (c & 3), columns cb + 8 (c >> 2) + 0..3; a second read (+ 4 columns) completes 8 consecutive columns
  //          = the 16 bytes of ONE buffer_store_dwordx4 whose row offset is a scalar.
  // Per chunk 4 conversions, 2 LDS writes, 2 LDS reads, 1 store. (The first form of this epilogue wrote a row-major slab with 8
  // ds_write_b16 per chunk: eight waves x 64 sub-dword LDS writes per tile made the epilogue LDS-bound — ~4000 cycles per tile against
  // 1024 matrix cycles per K step, profiles/r05_pw16_probe_v5.txt.) A wave's LDS accesses execute in order: the slab needs no barrier.
  // acc[tm][tn][r] is tile row wm TM 32 + tm 32 + 8 (r >> 2) + 4 lh + (r & 3), tile column wn CW + tn 32 + lrow.
  unsigned char* const slab = slabs + wave_s * SLAB;
  auto col_off = [](int c) __attribute__((always_inline)) { return c * 16 + (c >> 4) * 64; };   // column pitch 16 B, 64-byte skew per 16 columns
  unsigned char* const slab_w = slab + col_off(lrow) + 8 * lh;              // + col_off(32) per column tile: rows 4 lh .. 4 lh + 3 of the lane's column
  const int g16 = lane >> 4, c16 = lane & 15;             // 16-lane group of the transpose read and lane in it
  const int tr_r0 = 4 * (g16 & 1), tr_cb = 32 * (g16 >> 1);
  const unsigned char* const slab_r = slab + col_off(tr_cb + 8 * (c16 & 3) + (c16 >> 2)) + tr_r0 * 2;   // (the second read: 4 columns = 64 bytes further)
  const int e_row = tr_r0 + (c16 & 3), e_col = tr_cb + 8 * (c16 >> 2);      // what the lane holds after the two reads: (row of the chunk, 8 columns)
  const unsigned vo_st = (unsigned)(((wm * TM * 32 + e_row) * Nc + wn * CW + e_col) * 2);  // per-lane byte offset inside the tile (constant)
  auto epilogue1 = [&](int emt, int ent, int j) __attribute__((always_inline)) {
    const int m0 = emt * BM, n0 = ent * BN;
    if constexpr ((EPI & EPI_STATS) != 0) {
      float* const rd = red;
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
          rd[(wm * 2 + 0) * BN + c] = s;
          rd[(wm * 2 + 1) * BN + c] = ss;
        }
      }
    }
    if (R3M_PROBE(p) & 2) return;                         // probe 2: statistics only
    const int rows_valid = min(BM, p.M - m0);
    char* const ob = reinterpret_cast<char*>(p.out) + ((long long)m0 * Nc + n0) * 2;
    const int obytes = ((rows_valid - 1) * Nc + BN) * 2;  // a lane's offset is inside iff its row is < rows_valid
    int ncb = Nc * 2;
    asm volatile("" : "+s"(ncb));                         // row offsets computed at the point of use, not hoisted into scarce SGPRs
    // All chunks' LDS traffic is issued back to back — write c, read c, write c + 1, read c + 1, ... on ONE slab: the LDS executes a
    // wave's accesses in order, so read c sees chunk c and is done before write c + 1 lands — and only then the stores follow. (Chunk by
    // chunk — write, read, wait, store — every chunk paid a full LDS round trip: the epilogue was latency-bound, ~3400 cycles a tile.)
    p16_u32x4 v[TM * 4];
    static_for<TM * 4>([&](auto c_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value;             // chunk (tm, q): wave-tile rows 8 c .. 8 c + 7
      constexpr int tm = c >> 2, q = c & 3;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const p16_u32x2 w = {p16_pack(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1]), p16_pack(acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3])};
        *reinterpret_cast<p16_u32x2*>(slab_w + tn * (32 * 16 + 2 * 64)) = w;
      }
      __builtin_amdgcn_wave_barrier();
      const p16_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) p16_s16x4*)(slab_r));
      const p16_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) p16_s16x4*)(slab_r + 64));
      v[c] = __builtin_bit_cast(p16_u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      __builtin_amdgcn_wave_barrier();                    // (compiler fence: the chunk's reads stay in front of the next chunk's writes)
    });
    static_for<TM * 4>([&](auto c_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value;
      if (!(R3M_PROBE(p) & 1)) p16_st4(ob, obytes, vo_st, (c * 8) * ncb, v[c]);
    });
  };
  // part 2 (EPI_STATS, one barrier after part 1): combine the wave rows of each statistics row — stats[mt R + h][2][Nc]
  auto epilogue2 = [&](int emt, int ent, int j) __attribute__((always_inline)) {
    if constexpr ((EPI & EPI_STATS) != 0) {
      const float* const rd = red;
      for (int t = tid; t < BN; t += NCW * 64) {
        const int col = ent * BN + t;
#pragma unroll
        for (int h = 0; h < R; ++h) {
          float s = 0.f, ss = 0.f;
#pragma unroll
          for (int w = h * (WM / R); w < (h + 1) * (WM / R); ++w) {
            s += rd[(w * 2 + 0) * BN + t];
            ss += rd[(w * 2 + 1) * BN + t];
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

  int cs = 0, cj = 0, slot = 0;                           // K step in the tile, tile sequence number, ring slot of the step
  int ct = 0, cc = 0, wbuf = 0;                           // window form: tap and chunk of the step, its window buffer
  bool red_pend = false;
  int rmt = 0, rnt = 0;
  while (has) {
    p16_bar();                                            // the step's operands are in LDS (the loader waves waited for their DMA)
    if (red_pend) {
      epilogue2(rmt, rnt, cj - 1);
      red_pend = false;
      // single-step tiles: part 1 of THIS tile follows in this very step and rewrites the scratch part 2 has just read (the loader
      // waves run the same extra barrier)
      if (nsteps == 1) p16_bar();
    }
    if (WIN && cs == 0) win_rows(mt);
    const unsigned char* const sa = WIN ? smem + wbuf * winb : smem + slot * STAGE;
    const unsigned char* const sb = bring + slot * BSTRIDE;
    if (!(R3M_PROBE(p) & 16)) {                           // probe 16: no fragment reads, no MFMAs
      if (cs == 0) kstep(sa, sb, ct, std::true_type{});
      else kstep(sa, sb, ct, std::false_type{});
    }
    slot = slot + 1 == NS ? 0 : slot + 1;
    if constexpr (WIN) {
      if (++ct == 9) {
        ct = 0;
        ++cc;
        wbuf ^= 1;
      }
    }
    if (++cs == nsteps) {
      epilogue1(mt, nt, cj);
      if constexpr ((EPI & EPI_STATS) != 0) {
        red_pend = true;
        rmt = mt;
        rnt = nt;
      }
      cs = 0;
      cc = 0;
      ++cj;
      next_tile();
    }
  }
  p16_bar();
  if (red_pend) epilogue2(rmt, rnt, cj - 1);
}

// ---- launcher ----------------------------------------------------------------------------------------------------------
// OFF by default (round 5): bit-identical to the per-tile kernels and 10-25 % faster on some 1x1 launches in isolation, but every
// build of this file measured the STEP slower or neutral (configs[2] 90.1 -> 92.0 ms, configs[4] neutral; per-shape launches inside
// the step: profiles/r05_pw16_instep_v5.txt, r05_side_stream_ab.txt). r3m_debug_set_pw16 / R3M_PW16 (probe builds) switch it in:
// 1 = pointwise + gather forms, 3 = + the 3x3 window form; tests/test_gpu_pw16.py keeps it bit-identical while it is off.
static int g_pw16_mode = R3M_ENV_INT("R3M_PW16", 0);
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

// LDS bytes of a launch (slabs: 8 compute waves x 1280 bytes; statistics scratch [WM][2][BN])
static inline int p16_lds(int BM, int BN, int WMv, int NS, int win_W) {
  const int tail = 8 * 1280 + WMv * 2 * BN * 4;
  if (win_W < 0) return NS * (BM + BN) * 128 + tail;
  const int hri = ceil_div(BM + 2 * win_W + 2, 8);
  return 2 * hri * 1024 + NS * BN * 128 + 1024 + tail;
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
    const int lds = (p.Nc & 127) == 0 ? p16_lds(256, 128, 4, 3, p.Wi) : p16_lds(256, 64, 8, 3, p.Wi);
    if (ok && lds <= 160 * 1024 && ((long long)p.M + 2 * p.Wi + 2) * p.Ci * 2 < (long long)BUF_OOB) return 3;
  }
  return 2;
}

template <int BM, int BN, int WM, int WN, int NS, int EPI, int FORM>
static int launch_pw16_one(const GatherGemmParams& p, int W, int gridM, int gridN, hipStream_t s) {
  const int hri = FORM == 3 ? ceil_div(BM + 2 * p.Wi + 2, 8) : 0;
  const int lds = p16_lds(BM, BN, WM, NS, FORM == 3 ? p.Wi : -1);
  R3M_REQUIRE(lds <= 160 * 1024, "pw16_gemm: %d bytes of LDS", lds);
  auto kern = pw16_gemm_kernel<BM, BN, WM, WN, NS, EPI, FORM>;
  static DynLdsOptIn oi;
  if (int e = ensure_dyn_lds(oi, reinterpret_cast<const void*>(kern), lds, "pw16_gemm")) return e;
  hipLaunchKernelGGL(kern, dim3(W), dim3(768), lds, s, p, gridM, gridN, hri);
  return 0;
}

template <int BM, int BN, int WM, int WN, int NS, int FORM>
static int launch_pw16_shape(const GatherGemmParams& p, hipStream_t s) {
  const int gridM = ceil_div(p.M, BM), gridN = p.Nc / BN;
  const int slots = p16_cu_count();                                     // one twelve-wave block per CU
  const int W = gridM < slots ? gridM : slots;
  // (static row-panel split: the per-XCD tile queues the engine offers — p.tile_ctr — are not used by this kernel yet)
  switch (p.flags) {
    case 0: return launch_pw16_one<BM, BN, WM, WN, NS, 0, FORM>(p, W, gridM, gridN, s);
    case EPI_STATS: return launch_pw16_one<BM, BN, WM, WN, NS, EPI_STATS, FORM>(p, W, gridM, gridN, s);
  }
  set_last_error("pw16_gemm: unsupported epilogue flag combination %d", p.flags);
  return 1;
}

int launch_pw16(const GatherGemmParams& p, hipStream_t s) {
  const int form = pw16_form(p);
  R3M_REQUIRE(form != 0, "pw16_gemm: launch not eligible");
  // widths that are multiples of 128: 256 x 128 tile, compute waves 4 x 2 of 64 x 64, three ring slots (144 KB);
  // 64-channel outputs: 512 x 64, compute waves 8 x 1 of 64 x 64, two slots (window form: 256 x 64, waves of 32 x 64, three weight slots)
  const bool n128 = (p.Nc & 127) == 0;
  if (form == 3) return n128 ? launch_pw16_shape<256, 128, 4, 2, 3, 3>(p, s) : launch_pw16_shape<256, 64, 8, 1, 3, 3>(p, s);
  if (form == 2) return n128 ? launch_pw16_shape<256, 128, 4, 2, 3, 1>(p, s) : launch_pw16_shape<512, 64, 8, 1, 2, 1>(p, s);
  return n128 ? launch_pw16_shape<256, 128, 4, 2, 3, 0>(p, s) : launch_pw16_shape<512, 64, 8, 1, 2, 0>(p, s);
}

}  // namespace r3m
#endif  // R3M_PROBES
