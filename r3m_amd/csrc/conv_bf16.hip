// r3m_amd — bf16-activation convolution path for gfx950 (BASELINE configs[2], [4]: "bf16"): the same implicit GEMMs as
// conv.hip with bf16 operands in HBM/LDS, fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 BatchNorm partials taken
// from the accumulators, fp32 weight gradients. Master weights, optimizer state and every per-channel statistic stay
// fp32 (r3m_amd/encoder.py precision="bf16" is the mixed-precision counterpart of torch.autocast around the reference's
// encoder call, /root/reference/r3m/models/models_r3m.py:99; the reference itself trains in fp32).
//
//   gather_gemm_bf16_kernel : forward / dgrad. LDS rows are 64 bf16 = 128 B, the very byte image of the fp32 kernel
//       (8 rows per global_load_lds instruction, 16-byte slots XOR-swizzled by (row>>1)&7), so the fragment of MFMA step g
//       is ONE ds_read_b128 = 8 consecutive k of one row — exactly the 32x32x16 operand layout (k = 8*(lane>>5)..+7).
//   wgrad_bf16_kernel : the contraction runs over rows m while both operands are channel-contiguous, i.e. K-strided in
//       LDS. gfx950's transpose read (ds_read_b64_tr_b16) turns a [4 k][16 channel] LDS block into 4 consecutive k per
//       lane, so the operands still arrive by plain row DMA and no packing VALU is spent. 64-byte channel groups are
//       XOR-swizzled by the k row so the 4 rows one read touches sit in 4 different bank quarters.
#include "common.h"
#include "conv_dev.h"
#include <cstdlib>
#include <cstring>

namespace r3m {

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(256))) unsigned char g_zero_bytes[512];

__device__ __forceinline__ const char* sel_ptr(const char* s, const char* z, bool ok) {
  // bitwise select keeps the loader straight-line (a ?: is turned back into an exec-masked branch)
  const unsigned long long msk = ok ? ~0ull : 0ull;
  return reinterpret_cast<const char*>((reinterpret_cast<unsigned long long>(s) & msk) |
                                       (reinterpret_cast<unsigned long long>(z) & ~msk));
}

__device__ __forceinline__ void dma16(const char* src, unsigned char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// =====================================================================================================
// gather-GEMM on bf16 operands. Block BM x BN, K tiles of BK elements (one LDS row per GEMM row), WM x WN waves.
// LDS: a ring of NST stages of {A[BM][BK], B[BN][BK]} bf16 (dynamic LDS). Tile t + NST - 1 is issued right after the barrier that
// publishes tile t (all DMA pieces at once: with the 16x faster MFMA there is no issue cost worth hiding), so up to NST - 1 tiles
// are in flight per block; a wave waits only for ITS pieces of the oldest tile (s_waitcnt vmcnt(pieces of the newer tile), loads
// retire in order). BK = 64: 128-byte rows, 8 rows per DMA instruction, 16-byte slots swizzled by (row>>1)&7; BK = 32: 64-byte
// rows, 16 rows per instruction, slots swizzled by (row>>2)&3 — also conflict-free for ds_read_b128. Configurations in use
// (launch_gather_gemm_bf16 picks per shape):
//   4 waves, 2 stages, BK = 64 (64-80 KB: 2 blocks/CU)   — main-loop-bound launches: 3x3 convs, contracting 1x1 convs
//   4 waves, 2 stages, BK = 32 (ring 32-40 KB <= the epilogue slab: 3-4 blocks/CU) — expanding 1x1 convs, read-modify-write epilogues
//   4 waves, 1 stage,  BK = 64 (launches with ONE K tile: 1x1 convs with Cin = 64; LDS = the epilogue slab)
//   experiments: 256x128 tile (R3M_BF16_BIG), 8 waves with a 3-stage ring at 1 block/CU (R3M_BF16_RING: the L2 -> LDS DMA path
//     delivers 17 TB/s with 8 waves x 8 KB outstanding per CU, tools/micro/l2dma.hip, but one block per CU loses more in uncovered
//     prologue/epilogue than the deeper ring wins) — measurements in DESIGN.md section 5
// =====================================================================================================
//
// Register budget: the configurations whose LDS is just the 37-40 KB epilogue slab (32-wide K tiles; single stage) can share a CU
// three or four ways, but left alone the compiler gives the plain-store epilogue 117-152 VGPRs + 64 AGPRs = two waves per SIMD.
// __launch_bounds__'s second argument states the blocks per CU the kernel is budgeted for (168 registers per lane for three, 128
// for four — the latter with 8-56 bytes of scratch in the epilogue, still the faster choice for the 128x128 tile).
#ifndef R3M_GG16_OCC_WIDE
#define R3M_GG16_OCC_WIDE 4      // 128x128 tile (0 = compiler's choice)
#endif
#ifndef R3M_GG16_OCC_NARROW
#define R3M_GG16_OCC_NARROW 3    // 256x64 tile
#endif
template <int BM, int BN, int WM, int WN, int EPI, int NST, int BK = 64>
__global__ __launch_bounds__(WM * WN * 64, (BM == 256 && BN == 128) ? (BK == 32 ? 2 : 1)
                                           : !(BK == 32 || NST == 1) ? 1
                                           : (BN == 128 ? (R3M_GG16_OCC_WIDE ? R3M_GG16_OCC_WIDE : 1) : (R3M_GG16_OCC_NARROW ? R3M_GG16_OCC_NARROW : 1)))
void gather_gemm_bf16_kernel(const GatherGemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int RB = BK * 2;                             // bytes per LDS row
  constexpr int RPI = 1024 / RB;                         // rows per DMA instruction (8 or 16)
  constexpr int SPR = RB / 16;                           // 16-byte slots per row (8 or 4)
  constexpr int NG = BK / 16;                            // MFMA K groups per tile (4 or 2)
  constexpr int AJ = BM / (RPI * NW), BJ = BN / (RPI * NW);  // DMA instructions per wave per tile
  static_assert(BK == 64 || BK == 32, "K tile of 64 or 32");
  static_assert(AJ * RPI * NW == BM && BJ * RPI * NW == BN, "every wave stages whole DMA row groups");
  static_assert(NST >= 1 && NST <= 3, "ring of 1 (single K tile launches only), 2 or 3 stages");
  constexpr int NP = AJ + BJ;
  constexpr int STAGE = (BM + BN) * RB;                  // bytes
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int gridN = (p.Nc + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / gridN, nt = lid % gridN;
  const int m0 = mt * BM, n0 = nt * BN;
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);

  const int srow = lane / SPR, pslot = lane % SPR;
  auto swz = [](int r) __attribute__((always_inline)) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
  const int Hb = p.simple_rows ? 1 : p.Hi, Wb = p.simple_rows ? 1 : p.Wi;
  // Round 3: both operands arrive through buffer descriptors (conv_dev.h buf_dma16; see wgrad_bf16_kernel for the measurement that
  // motivated it: these kernels are instruction-issue-bound and the per-lane 64-bit source pointers were most of the vector work).
  //   weights: constant per-lane offset; the (tap, K chunk) offset is wave-uniform and rides in the instruction's scalar offset;
  //   activations: per-lane 32-bit offset relative to the frame of the tile's first row, recomputed only when the TAP changes
  //     (padding -> out of range -> zeros); the K chunk inside the tap is the scalar offset again.
  const long long imgA = (long long)p.Hi * p.Wi * p.Ci * 2;               // bytes per frame of A (simple rows: unused)
  const int hwg = p.Hg * p.Wg;
  const int nfirst = p.simple_rows ? 0 : (m0 < p.M ? m0 / hwg : 0);
  const char* a_base = p.simple_rows ? Ab + (long long)m0 * p.Ci * 2 : Ab + (long long)nfirst * imgA;
  int a_bytes;
  {
    const long long rest = p.simple_rows ? (long long)(p.M - m0) * p.Ci * 2 : (long long)(p.N - nfirst) * imgA;
    a_bytes = rest <= 0 ? 0 : (rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
  }
  RowDesc ad[AJ];
  unsigned arel[AJ];                   // byte offset of the row's frame (simple rows: of the row) from a_base + swizzled slot, or BUF_OOB
  unsigned avoff[AJ];                  // offset of the current tap's pixel (per lane), refreshed by set_tap
  unsigned bvoff[BJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int r = wave * (BM / NW) + j * RPI + srow;
    const unsigned col = (unsigned)((pslot ^ swz(r)) * 16);
    const int m = m0 + r;
    ad[j] = decode_row(p, m);
    if (m >= p.M) arel[j] = BUF_OOB;
    else if (p.simple_rows) arel[j] = (unsigned)(r * p.Ci * 2) + col;
    else arel[j] = (unsigned)(m / hwg - nfirst) * (unsigned)imgA + col;
    avoff[j] = arel[j];
  }
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int r = wave * (BN / NW) + j * RPI + srow;
    const int n = min(n0 + r, p.Nc - 1);                  // columns past Nc are computed on a clamped row, never stored
    bvoff[j] = (unsigned)(n * p.T * p.Ci * 2) + (unsigned)((pslot ^ swz(r)) * 16);
  }
  const int b_bytes = p.Nc * p.T * p.Ci * 2;

  const int kpt = p.Ci / BK;          // K tiles per tap
  const int nk = p.ntaps * kpt;

  // issue cursor: the tile the next DMA pieces belong to
  int tap_n = 0, chunk_n = 0;
  int pack_cur = nk > 0 ? p.tap[0] : 0;
  int pack_next = p.ntaps > 1 ? p.tap[1] : pack_cur;
  auto set_tap = [&](int pack) __attribute__((always_inline)) {
    if (p.simple_rows) return;
    const int dy = (pack << 24) >> 24, dx = (pack << 16) >> 24;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int iy = ad[j].iy + dy, ix = ad[j].ix + dx;
      const bool in = ((unsigned)iy < (unsigned)Hb) && ((unsigned)ix < (unsigned)Wb) && (arel[j] < BUF_OOB);
      avoff[j] = in ? arel[j] + (unsigned)((iy * p.Wi + ix) * p.Ci * 2) : BUF_OOB;
    }
  };
  if (nk > 0) set_tap(pack_cur);
  auto advance = [&]() __attribute__((always_inline)) {
    if (++chunk_n == kpt) {
      chunk_n = 0;
      ++tap_n;
      pack_cur = pack_next;
      pack_next = p.tap[min(tap_n + 1, p.ntaps - 1)];
      set_tap(pack_cur);
    }
  };
  // one DMA piece of the cursor's tile into ring slot `stage`
  auto issue_piece = [&](int stage, auto pc_c) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_c)::value;
    const int c0b = chunk_n * RB;
    unsigned char* la = smem + stage * STAGE + wave * (BM / NW) * RB;
    unsigned char* lb = smem + stage * STAGE + BM * RB + wave * (BN / NW) * RB;
    if constexpr (pc < AJ) {
      constexpr int j = pc;
      if (R3M_PROBE(p) & 24) {
        const int dy = (pack_cur << 24) >> 24, dx = (pack_cur << 16) >> 24;
        if ((R3M_PROBE(p) & 8) && dx != 0) return;     // probe 8: stage the A tile of the centre-column taps only (bytes-per-flop what-if)
        if ((R3M_PROBE(p) & 16) && (dx != 0 || dy != 0)) return;   // probe 16: ... of the centre tap only
      }
      buf_dma16(a_base, a_bytes, la + j * 1024, avoff[j], c0b);
    } else {
      constexpr int j = pc - AJ;
      const int wt = pack_cur >> 16;
      buf_dma16(Bb, b_bytes, lb + j * 1024, bvoff[j], wt * p.Ci * 2 + c0b);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment addressing: row = lrow (+32 per MFMA tile), logical slot 2g+h, physical slot = logical ^ ((row>>1)&7)
  const int lrow = lane & 31, lh = lane >> 5;
  const int xr = swz(lrow);           // tile row offsets are multiples of 32: the swizzle key only depends on lrow
  int goff[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) goff[g] = ((2 * g + lh) ^ xr) * 16;
  const unsigned char* fragA0 = smem + (wm * TM * 32 + lrow) * RB;
  const unsigned char* fragB0 = smem + BM * RB + (wn * TN * 32 + lrow) * RB;

  // prologue: tiles 0 .. NST-2 (NST == 1: the launch has ONE K tile — 1x1 convs with Cin = 64 — and it is simply loaded; the
  // block then needs 37 instead of 64 KB of LDS, so three blocks share a CU instead of two)
  int istage = 0;                         // ring slot the cursor's tile goes to
  if (NST == 1 && nk > 0) static_for<NP>([&](auto pc) __attribute__((always_inline)) { issue_piece(0, pc); });
#pragma unroll
  for (int t = 0; t < NST - 1; ++t) {
    if (t < nk) {
      static_for<NP>([&](auto pc) __attribute__((always_inline)) { issue_piece(istage, pc); });
      advance();
      istage = (istage + 1 == NST) ? 0 : istage + 1;
    }
  }
  int cstage = 0;                         // ring slot of the tile being multiplied
  for (int kt = 0; kt < nk; ++kt) {
    // my pieces of tile kt have landed; the pieces of one newer tile (NST == 3) may still be in flight
    if (NST == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // everyone's have; and everyone is done reading ring slot istage
    const bool more = NST > 1 && (kt + NST - 1 < nk) && !(R3M_PROBE(p) & 2);   // probe 2: no DMA after the prologue
    const unsigned char* fa = fragA0 + cstage * STAGE;
    const unsigned char* fb = fragB0 + cstage * STAGE;
    // all DMA pieces right after the barrier: with the 16x faster MFMA there is no issue cost worth hiding, and the earlier
    // the loads leave the sooner they land (+2..8 % over spreading them between the MFMA groups; probe 4 = spread)
    const bool clustered = (R3M_PROBE(p) & 4) == 0;
    if (more && clustered) static_for<NP>([&](auto pc) __attribute__((always_inline)) { issue_piece(istage, pc); });
    static_for<NG>([&](auto g_c) __attribute__((always_inline)) {
      constexpr int g = decltype(g_c)::value;
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const bf16x8*>(fa + t * 32 * RB + goff[g]);
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const bf16x8*>(fb + t * 32 * RB + goff[g]);
      if (more && !clustered) {   // DMA pieces of tile kt + NST - 1, spread over the MFMA groups
        constexpr int P0 = g * NP / NG, P1 = (g + 1) * NP / NG;
        static_for<P1 - P0>([&](auto q_c) __attribute__((always_inline)) {
          issue_piece(istage, std::integral_constant<int, P0 + decltype(q_c)::value>{});
        });
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    });
    if (more) {
      advance();
      istage = (istage + 1 == NST) ? 0 : istage + 1;
    }
    cstage = (cstage + 1 == NST) ? 0 : cstage + 1;
  }
  __syncthreads();   // all fragment reads done before the epilogue reuses the stages

  if (R3M_PROBE(p) & 1) {   // probe 1: no epilogue (one store keeps the accumulators alive)
    if (acc[0][0][0] + acc[TM - 1][TN - 1][3] == 123.456f) reinterpret_cast<float*>(p.out)[0] = 1.f;
    return;
  }
  // a 256-row tile writes one partial row per 128-row half (same partial geometry as the 128-row tiles)
  if (EPI & EPI_STATS) gg_stats<BM, BN, WM, WN, (BM == 256 && BN == 128)>(p, acc, reinterpret_cast<float*>(smem), n0, mt);
  // the launcher sizes the dynamic LDS as max(ring, epilogue slabs): with NST == 1 the slabs are the larger
  constexpr int SMEM_F = (NST * STAGE < 40960 ? 40960 : NST * STAGE) / 4;
  if ((p.Nc & 7) == 0) gg_store_bf16<BM, BN, WM, WN, EPI, SMEM_F>(p, acc, reinterpret_cast<float*>(smem), m0, n0);
  else gg_epilogue<BM, BN, WM, WN, EPI & ~EPI_STATS, SMEM_F, bf16_t>(p, acc, reinterpret_cast<float*>(smem), m0, n0, mt);
}

// =====================================================================================================
// 3x3 / stride 1 / pad 1 convolutions (forward and dgrad) with a HALO tile: the nine taps of a tile of BM consecutive output
// pixels read input pixels m0 - (W+1) .. m0 + BM - 1 + (W+1) — one window of BM + 2W + 2 rows instead of nine BM-row tiles. The
// gather kernel above stages 9 x (BM + BN) rows per 64-channel chunk and is bound by exactly that L2 -> LDS staging rate
// (DESIGN.md section 5); here a chunk stages the window once plus nine BN-row weight tiles (0.57x the rows at 14x14 / 128x128).
//   LDS: [window of HRI x 1 KiB][1 KiB with a zero row][2 weight stages of BN x 128 B]; same 128-byte swizzled rows.
//   A fragment of tap (dy, dx) = halo rows shifted by dy*W + dx (the swizzle key follows the shifted row); a lane whose pixel
//   has no (y+dy, x+dx) inside the image reads the zero row instead -> the border arithmetic of the gather kernel (zero
//   contributions) without per-tap staging. Rows >= M read zeros for every tap (BatchNorm partials rely on that).
//   Weight tiles: ring of 2, one tile per (chunk, tap) step. ONE window buffer, reloaded between chunks: a second buffer (next
//   window arriving under the current chunk's taps) measured slower than the third block per CU the smaller footprint allows
//   (14x14 x 256: 0.372 vs 0.363 ms, 7x7 x 512: 0.304 vs 0.277 ms; the gather kernel: 0.440 / 0.369).
// =====================================================================================================
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256, BM / WM > 64 ? 2 : 1) void conv3x3_halo_bf16_kernel(const GatherGemmParams p, const int hri) {
  static_assert(WM * WN == 4 && BN / WN == 64 && (BM / WM == 64 || BM / WM == 128), "four waves, 64 or 128 rows x 64 columns each");
  constexpr int TM = BM / WM / 32, TN = 2;
  constexpr int BJ = BN / 32;                        // weight DMA instructions per wave per tile
  constexpr int WSTAGE = BN * 128;                   // bytes
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int gridN = (p.Nc + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / gridN, nt = lid % gridN;
  const int m0 = mt * BM, n0 = nt * BN;
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);
  const int W = p.Wi, HW = p.Hi * p.Wi;
  const int halo_bytes = hri * 1024;
  unsigned char* zrow = smem + halo_bytes;                            // 128 zero bytes (1 KiB reserved)
  unsigned char* wring = zrow + 1024;
  if (tid < 16) *reinterpret_cast<uint4*>(zrow + tid * 16) = make_uint4(0u, 0u, 0u, 0u);   // 256 zero bytes: see the fragment addressing below

  const int srow = lane >> 3, pslot = lane & 7;
  const long long hb = (long long)m0 - (W + 1);                        // pixel staged in halo row 0
  // Round 3: buffer descriptors (see gather_gemm_bf16_kernel). Window rows: offsets relative to the first in-tensor pixel of the
  // window, rows before the tensor get an out-of-range offset, rows past it fall off the descriptor; the 64-channel chunk is the
  // scalar offset. Weights: constant per-lane offset, (tap, chunk) in the scalar offset.
  const long long hb0 = hb > 0 ? hb : 0;
  const char* a_base = Ab + hb0 * p.Ci * 2;
  int a_bytes;
  {
    const long long rest = ((long long)p.M - hb0) * p.Ci * 2;
    a_bytes = rest <= 0 ? 0 : (rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
  }
  const unsigned hcol = (unsigned)pslot;                               // physical slot; the logical one depends on the row
  // DMA instruction i of the window of chunk c: halo rows 8i .. 8i+7
  auto issue_halo = [&](int c, int i) __attribute__((always_inline)) {
    const int hr = 8 * i + srow;
    const long long px = hb + hr;
    const unsigned voff = px >= hb0 ? (unsigned)((int)(px - hb0) * p.Ci * 2) + ((hcol ^ (unsigned)((hr >> 1) & 7)) * 16u) : BUF_OOB;
    buf_dma16(a_base, a_bytes, smem + i * 1024, voff, c * 128);
  };
  unsigned bvoff[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int r = wave * (BN / 4) + j * 8 + srow;
    const int n = min(n0 + r, p.Nc - 1);
    bvoff[j] = (unsigned)(n * p.T * p.Ci * 2) + (unsigned)((pslot ^ ((r >> 1) & 7)) * 16);
  }
  const int b_bytes = p.Nc * p.T * p.Ci * 2;
  auto issue_w = [&](int c, int pack, int stage) __attribute__((always_inline)) {
    const int wt = pack >> 16;
    unsigned char* lb = wring + stage * WSTAGE + wave * (BN / 4) * 128;
#pragma unroll
    for (int j = 0; j < BJ; ++j) buf_dma16(Bb, b_bytes, lb + j * 1024, bvoff[j], wt * p.Ci * 2 + c * 128);
  };

  // per-lane rows of the two A row tiles: halo row of the centre tap and a 9-bit validity mask (bit = tap position in p.tap[])
  const int lrow = lane & 31, lh = lane >> 5;
  int crow[TM];
  unsigned vmask[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int r = wm * (BM / WM) + t * 32 + lrow;
    crow[t] = r + W + 1;
    const int m = m0 + r;
    unsigned v = 0;
    if (m < p.M) {
      const int rem = m % HW;
      const int y = rem / W, x = rem - y * W;
      for (int k = 0; k < 9; ++k) {
        const int pk = p.tap[k];
        const int dy = (pk << 24) >> 24, dx = (pk << 16) >> 24;
        if ((unsigned)(y + dy) < (unsigned)p.Hi && (unsigned)(x + dx) < (unsigned)W) v |= 1u << k;
      }
    }
    vmask[t] = v;
  }
  const int zoff = (int)(zrow - smem);
  int goffb[4];
  const int xrb = (lrow >> 1) & 7;
#pragma unroll
  for (int g = 0; g < 4; ++g) goffb[g] = ((2 * g + lh) ^ xrb) * 16;
  const unsigned char* fragB0 = wring + (wn * 64 + lrow) * 128;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nc = p.Ci >> 6;
  // prologue: window of chunk 0, weight tile (0, tap 0)
  for (int i = wave; i < hri; i += 4) issue_halo(0, i);
  issue_w(0, p.tap[0], 0);
  int stage = 0;
  for (int c = 0; c < nc; ++c) {
#pragma unroll 1
    for (int k = 0; k < 9; ++k) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const bool last_tap = k == 8;     // next weight tile
      if (!last_tap) issue_w(c, p.tap[k + 1], stage ^ 1);
      else if (c + 1 < nc) issue_w(c + 1, p.tap[0], stage ^ 1);
      const int pk = p.tap[k];
      const int shift = ((pk << 24) >> 24) * W + ((pk << 16) >> 24);
      int abase[TM], akey[TM];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const bool ok = (vmask[t] >> k) & 1u;
        const int rv = crow[t] + shift;
        // Round 5: a masked lane reads its zeros at the SAME position of the 256-byte bank window its real row would have used
        // (row parity picks the 128-byte half, the swizzled slot the 16 bytes): the 16 lanes of a ds_read_b128 group then still hit
        // 16 distinct positions. With ONE zero row at a fixed position a masked lane collided with whichever lane owned that slot:
        // PMC showed 25 % of this kernel's LDS cycles as bank conflicts at 14 x 14 (profiles/r05_pw16_pmc_v5_halo.csv).
        abase[t] = ok ? rv * 128 : zoff + (rv & 1) * 128;
        akey[t] = (rv >> 1) & 7;
      }
      const unsigned char* fb = fragB0 + stage * WSTAGE;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const bf16x8*>(smem + abase[t] + (((2 * g + lh) ^ akey[t]) * 16));
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const bf16x8*>(fb + t * 32 * 128 + goffb[g]);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
      stage ^= 1;
    }
    if (c + 1 < nc) {
      __syncthreads();                                 // every wave is done with this chunk's window
      for (int i = wave; i < hri; i += 4) issue_halo(c + 1, i);
    }
  }
  __syncthreads();   // all fragment reads done before the epilogue reuses the LDS
  // a 256 x 128 tile writes one partial row per 128-row half (the partial geometry of the 128-row tiles)
  if (EPI & EPI_STATS) gg_stats<BM, BN, WM, WN, (BM == 256 && BN == 128)>(p, acc, reinterpret_cast<float*>(smem), n0, mt);
  gg_store_bf16<BM, BN, WM, WN, EPI, 40960 / 4>(p, acc, reinterpret_cast<float*>(smem), m0, n0);
}

// LDS bytes of the halo kernel for a tile of BM pixels at image width W
static inline int halo_lds_bytes(int BM, int BN, int W) {
  const int hri = ceil_div(BM + 2 * W + 2, 8);
  const int b = hri * 1024 + 1024 + 2 * BN * 128;
  return b < 40960 ? 40960 : b;
}

template <int BM, int BN, int WM, int WN, int EPI>
static int halo_launch_one(const GatherGemmParams& p, hipStream_t s) {
  const int hri = ceil_div(BM + 2 * p.Wi + 2, 8);
  const int lds = halo_lds_bytes(BM, BN, p.Wi);
  auto kern = conv3x3_halo_bf16_kernel<BM, BN, WM, WN, EPI>;
  static DynLdsOptIn optin;
  if (int e = ensure_dyn_lds(optin, reinterpret_cast<const void*>(kern), lds, "conv3x3_halo(bf16)")) return e;
  const int grid = ceil_div(p.M, BM) * ceil_div(p.Nc, BN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p, hri);
  return 0;
}

template <int BM, int BN, int WM, int WN>
static int halo_launch(const GatherGemmParams& p, hipStream_t s) {
  switch (p.flags) {
    case 0: return halo_launch_one<BM, BN, WM, WN, 0>(p, s);
    case EPI_STATS: return halo_launch_one<BM, BN, WM, WN, EPI_STATS>(p, s);
    case EPI_ACCUM: return halo_launch_one<BM, BN, WM, WN, EPI_ACCUM>(p, s);
    case EPI_MASKED_ADD: return halo_launch_one<BM, BN, WM, WN, EPI_MASKED_ADD>(p, s);
    case EPI_BNRED: return halo_launch_one<BM, BN, WM, WN, EPI_BNRED>(p, s);
    case EPI_BNRED | EPI_MASKED_ADD: return halo_launch_one<BM, BN, WM, WN, EPI_BNRED | EPI_MASKED_ADD>(p, s);
    case EPI_AFFINE | EPI_RELU: return halo_launch_one<BM, BN, WM, WN, EPI_AFFINE | EPI_RELU>(p, s);
    case EPI_AFFINE | EPI_ACCUM | EPI_RELU: return halo_launch_one<BM, BN, WM, WN, EPI_AFFINE | EPI_ACCUM | EPI_RELU>(p, s);
    default: set_last_error("conv3x3_halo(bf16): unsupported epilogue flag combination %d", p.flags); return 1;
  }
}

// 3x3 stride-1 launches go through the halo kernel; wide launches with enough rows use the 256 x 128 tile (0.55x the staged rows
// of the 128 x 128 halo tile again: 14x14 x 256 at 1280 frames 0.317 ms against 0.380, the gather kernel 0.450).
// R3M_BF16_HALO=0: gather kernel; =2: 128-row tiles only; =4: 256-row tile whatever M (tests)
static int gg16_halo() {
  const int v = R3M_ENV_INT("R3M_BF16_HALO", 1);
  return v;
}

static bool halo_eligible(const GatherGemmParams& p) {
  if (p.ntaps != 9 || p.simple_rows || p.is != 1 || p.os != 1 || p.ooy != 0 || p.oox != 0) return false;
  if (p.Hg != p.Hi || p.Wg != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi || (p.Ci & 63) || (p.Nc & 7)) return false;
  if (!(p.Nc % 128 == 0 || p.Nc == 64)) return false;
  if (p.flags != 0 && p.flags != EPI_STATS && p.flags != EPI_ACCUM && p.flags != EPI_MASKED_ADD && p.flags != EPI_BNRED &&
      p.flags != (EPI_BNRED | EPI_MASKED_ADD) && p.flags != (EPI_AFFINE | EPI_RELU) && p.flags != (EPI_AFFINE | EPI_ACCUM | EPI_RELU))
    return false;
  for (int k = 0; k < 9; ++k)
    if (p.dy[k] < -1 || p.dy[k] > 1 || p.dx[k] < -1 || p.dx[k] > 1) return false;
  return true;
}

// tile rows of the halo launch (0: the window does not fit the LDS -> gather kernel)
static int halo_tile_rows(const GatherGemmParams& p) {
  if (p.Nc == 64) return halo_lds_bytes(256, 64, p.Wi) <= 160 * 1024 ? 256 : 0;
  const int h = gg16_halo();
  const int min_m = h == 4 ? 0 : 65536;      // 7x7 x 512 at 1280 frames (M = 62720) measured the same either way
  if (h != 2 && p.M >= min_m && halo_lds_bytes(256, 128, p.Wi) <= 80 * 1024) return 256;
  return halo_lds_bytes(128, 128, p.Wi) <= 160 * 1024 ? 128 : 0;
}

static int launch_halo(const GatherGemmParams& p, int rows, hipStream_t s) {
  if (p.Nc == 64) return halo_launch<256, 64, 4, 1>(p, s);
  return rows == 256 ? halo_launch<256, 128, 2, 2>(p, s) : halo_launch<128, 128, 2, 2>(p, s);
}

static inline bool gg_wide(int Nc) { return (Nc % 128) == 0; }

template <int BM, int BN, int WM, int WN, int EPI, int NST, int BK>
static int gg16_launch_one(const GatherGemmParams& p, int grid, hipStream_t s) {
  constexpr int slab = WM * WN * (BM / WM > 64 ? 64 : BM / WM) * (BN / WN + 8) * 2;   // bf16 epilogue slabs (64 rows per wave and pass)
  constexpr int slab32 = WM * WN * 32 * (BN / WN + 4) * 4;                  // fp32 slabs of the read-modify-write epilogues
  constexpr int tiles = NST * (BM + BN) * BK * 2;
  constexpr int lds = tiles > slab ? (tiles > slab32 ? tiles : slab32) : (slab > slab32 ? slab : slab32);
  auto kern = gather_gemm_bf16_kernel<BM, BN, WM, WN, EPI, NST, BK>;
  static DynLdsOptIn optin;         // > 64 KiB of dynamic LDS needs the opt-in once per kernel (and device)
  if (int e = ensure_dyn_lds(optin, reinterpret_cast<const void*>(kern), lds, "gather_gemm(bf16)")) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, s, p);
  return 0;
}

template <int BM, int BN, int WM, int WN, int NST, int BK = 64>
static int gg16_launch(const GatherGemmParams& p, int grid, hipStream_t s) {
  switch (p.flags) {
    case 0: return gg16_launch_one<BM, BN, WM, WN, 0, NST, BK>(p, grid, s);
    case EPI_STATS: return gg16_launch_one<BM, BN, WM, WN, EPI_STATS, NST, BK>(p, grid, s);
    case EPI_ACCUM: return gg16_launch_one<BM, BN, WM, WN, EPI_ACCUM, NST, BK>(p, grid, s);
    case EPI_MASKED_ADD: return gg16_launch_one<BM, BN, WM, WN, EPI_MASKED_ADD, NST, BK>(p, grid, s);
    case EPI_BNRED: return gg16_launch_one<BM, BN, WM, WN, EPI_BNRED, NST, BK>(p, grid, s);
    case EPI_BNRED | EPI_MASKED_ADD: return gg16_launch_one<BM, BN, WM, WN, EPI_BNRED | EPI_MASKED_ADD, NST, BK>(p, grid, s);
    case EPI_AFFINE: return gg16_launch_one<BM, BN, WM, WN, EPI_AFFINE, NST, BK>(p, grid, s);                           // inference forward (round 6)
    case EPI_AFFINE | EPI_RELU: return gg16_launch_one<BM, BN, WM, WN, EPI_AFFINE | EPI_RELU, NST, BK>(p, grid, s);
    case EPI_AFFINE | EPI_ACCUM | EPI_RELU: return gg16_launch_one<BM, BN, WM, WN, EPI_AFFINE | EPI_ACCUM | EPI_RELU, NST, BK>(p, grid, s);
    default: set_last_error("gather_gemm(bf16): unsupported epilogue flag combination %d", p.flags); return 1;
  }
}

// R3M_BF16_RING=n: use the 8-wave / 3-stage configuration for launches with >= n K tiles. Default 0 = never: measured on
// ResNet-50 shapes it LOSES 20-30 % to two co-resident 4-wave blocks (one block per CU leaves its prologue and epilogue
// uncovered, and the 2 x 4 wave layout reads 1.5x the LDS bytes per tile). Kept for experiments.
static int gg16_ring_min() {
  const int v = R3M_ENV_INT("R3M_BF16_RING", 0);
  return v;
}

// R3M_BF16_SINGLE=0 disables the single-stage configuration of one-K-tile launches
static bool gg16_single() {
  const int v = R3M_ENV_INT("R3M_BF16_SINGLE", 1);
  return v != 0;
}

// K tile width of a multi-tile launch. 32 (16 KB stages, THREE blocks per CU instead of two) wins where the epilogue is a large
// part of a block's life — the expanding 1x1 convs and everything with a read-modify-write epilogue: -10..-33 % per launch on
// ResNet-50 — because a third resident block covers it; 64 (half the barriers per K) wins by 3..16 % where the main loop
// dominates: K >= 4 N, i.e. the 3x3 convs and the contracting 1x1 convs (per-shape table measured in round 2: profiles/r02_*).
// The 256x64 tile's 3x3 launches are the exception (measured -4 % with 32). R3M_BF16_BK=32 / 64 forces one width.
static bool gg16_bk32(const GatherGemmParams& p, bool wide) {
  const int v = R3M_ENV_INT("R3M_BF16_BK", 0);
  if (v == 32) return true;
  if (v == 64) return false;
  const long long ktot = (long long)p.ntaps * p.Ci;
  const bool mainloop_bound = ktot >= 4LL * p.Nc && (wide || p.ntaps == 1);
  return !mainloop_bound;
}

// 256 x 128 block tile (waves 2 x 2, 128 x 64 per wave) for the main-loop-bound wide launches: a 64 x 64 wave tile reads
// (64 + 64) x 32 B of fragments per 4 MFMAs — with 8 waves per CU that is exactly the LDS's 128 B/clk, so those launches sat at
// ~35 % of the MFMA peak; 128 x 64 per wave needs 0.75x the fragment bytes per flop. R3M_BF16_BIG=0 disables, =64 uses 64-wide
// K tiles (96 KB ring, one block per CU) instead of 32-wide (48 KB, two blocks).
static int gg16_big() {
  const int v = R3M_ENV_INT("R3M_BF16_BIG", 0);
  return v;
}

// which kernel family launch_gather_gemm_bf16 runs (a pure function of the launch parameters; r3m_debug_conv_route reports it without a GPU)
int gg16_route(const GatherGemmParams& p) {
  if (row16_eligible(p)) return 32;
  if (gg16_halo() && halo_eligible(p) && pw16_form(p) != 3 && halo_tile_rows(p)) return 31;
  if (pw16_form(p)) return 33;
  return 30;
}

int launch_gather_gemm_bf16(const GatherGemmParams& p, hipStream_t s) {
  R3M_REQUIRE(p.Ci % 64 == 0, "gather_gemm(bf16): Ci=%d must be a multiple of 64", p.Ci);
  R3M_REQUIRE(p.Nc % 4 == 0, "gather_gemm(bf16): Nc=%d must be a multiple of 4", p.Nc);
  // every loader here addresses the weights (and a frame's worth of activations) through 32-bit buffer offsets
  R3M_REQUIRE((long long)p.Nc * p.T * p.Ci * 2 < (long long)BUF_OOB, "gather_gemm(bf16): weight tensor of %lld bytes exceeds the 32-bit buffer range",
              (long long)p.Nc * p.T * p.Ci * 2);
  R3M_REQUIRE(p.simple_rows || (long long)p.Hi * p.Wi * p.Ci * 2 < (long long)BUF_OOB, "gather_gemm(bf16): one frame of %lld bytes exceeds the 32-bit buffer range",
              (long long)p.Hi * p.Wi * p.Ci * 2);
  const double flops = 2.0 * (double)p.M * (double)p.Nc * (double)p.ntaps * p.Ci;
  const int nk = p.ntaps * (p.Ci / 64);
  const bool ring = gg16_ring_min() > 0 && nk >= gg16_ring_min();
  const bool w8 = gg16_ring_min() < 0;       // experiment: 8 waves, 2 stages (2 blocks/CU = 16 waves/CU)
  int rc;
  if (row16_eligible(p)) {                     // round 6: 128-multiple-wide 3x3 / stride-1 launches -> persistent kernel-row kernel (conv_row16.hip)
    prof_begin(KC_GEMM_WIDE, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    rc = launch_conv3x3_row_bf16(p, s);
    prof_bytes(gather_gemm_alg_bytes(p, 2));
    prof_end(s);
    if (rc) return rc;
    return check_launch("conv3x3_row_bf16");
  }
  const int halo_rows = (gg16_halo() && halo_eligible(p) && pw16_form(p) != 3) ? halo_tile_rows(p) : 0;   // (3: the persistent window form takes it)
  if (halo_rows) {
    prof_begin(gg_wide(p.Nc) ? KC_GEMM_WIDE : KC_GEMM_NARROW, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    rc = launch_halo(p, halo_rows, s);
    prof_bytes(gather_gemm_alg_bytes(p, 2));
    prof_end(s);
    if (rc) return rc;
    return check_launch("conv3x3_halo_bf16");
  }
  if (pw16_form(p)) {                          // dense / parity-strided output rows: the persistent warp-specialised kernel (conv_pw16.hip)
    prof_begin(gg_wide(p.Nc) ? KC_GEMM_WIDE : KC_GEMM_NARROW, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    rc = launch_pw16(p, s);
    prof_bytes(gather_gemm_alg_bytes(p, 2));
    prof_end(s);
    if (rc) return rc;
    return check_launch("pw16_gemm");
  }
  if (gg_wide(p.Nc)) {
    const int grid = ceil_div(p.M, 128) * ceil_div(p.Nc, 128);
    prof_begin(KC_GEMM_WIDE, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    // R3M_BF16_BIG < 0: every multi-tile wide launch (tests)
    const bool big = gg16_big() < 0 ? nk >= 2 : gg16_big() != 0 && !gg16_bk32(p, true) && nk >= 4 && p.M >= 256 * 256;
    const int gridb = ceil_div(p.M, 256) * ceil_div(p.Nc, 128);
    rc = big ? (gg16_big() == 64 ? gg16_launch<256, 128, 2, 2, 2, 64>(p, gridb, s) : gg16_launch<256, 128, 2, 2, 2, 32>(p, gridb, s))
         : ring ? gg16_launch<128, 128, 2, 4, 3>(p, grid, s) : w8 ? gg16_launch<128, 128, 2, 4, 2>(p, grid, s)
         : (nk == 1 && gg16_single()) ? gg16_launch<128, 128, 2, 2, 1>(p, grid, s)
#ifdef R3M_PROBES
         : (gg16_bk32(p, true) && R3M_ENV_INT("R3M_BF16_NST3", 0)) ? gg16_launch<128, 128, 2, 2, 3, 32>(p, grid, s)   // 3-stage ring, 48 KB: 3 blocks/CU
#endif
         : gg16_bk32(p, true) ? gg16_launch<128, 128, 2, 2, 2, 32>(p, grid, s) : gg16_launch<128, 128, 2, 2, 2>(p, grid, s);
  } else {
    const int grid = ceil_div(p.M, 256) * ceil_div(p.Nc, 64);
    prof_begin(KC_GEMM_NARROW, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    rc = ring ? gg16_launch<256, 64, 4, 2, 3>(p, grid, s) : w8 ? gg16_launch<256, 64, 4, 2, 2>(p, grid, s)
         : (nk == 1 && gg16_single()) ? gg16_launch<256, 64, 4, 1, 1>(p, grid, s)
         : gg16_bk32(p, false) ? gg16_launch<256, 64, 4, 1, 2, 32>(p, grid, s) : gg16_launch<256, 64, 4, 1, 2>(p, grid, s);
  }
  prof_bytes(gather_gemm_alg_bytes(p, 2));
  prof_end(s);
  if (rc) return rc;
  return check_launch("gather_gemm_bf16");
}

// =====================================================================================================
// wgrad on bf16 operands: dW[co, tap, ci] (fp32 split-K partials) = sum_m dY[m, co] * X[pix(m) + off(tap), ci].
// Block tile BMt (co) x BNt (ci), K step 64 rows, 4 waves as 2 x 2. LDS image per operand: [64 k][BMt] bf16, a k row is
// BMt*2 bytes, one DMA instruction covers 1 KiB = 4 (128 wide) or 8 (64 wide) k rows. 64-byte channel groups are
// XOR-swizzled with the k row (on the DMA's global source side and again on the fragment read).
// Fragment of MFMA step s, half r: ds_read_b64_tr_b16 — in each 16-lane group lane c returns element (c & 3) of what
// lane 4j + (c >> 2) addressed, for j = 0..3. With lane q = 4j + i addressing k row kb + j, channels cb + 4i..4i+3, lane c
// receives channel cb + c at k = kb..kb+3: four consecutive k of one channel — half an MFMA operand.
// =====================================================================================================
// NT = 3 (round 3, "kernel-row" blocks of a 3-wide kernel): one block owns the THREE taps (kh, 0..2) of one kernel row for its
// (co, ci) tile — three accumulator sets (192 registers on the 128 x 128 tile). The dY tile of a K step is staged ONCE and its
// fragments are read ONCE for the three taps, the (oy, ox) walk of the staged X rows is shared (the taps differ by one pixel
// in x), and each tap still stages its own exactly-masked X rows (no register masks, any width / stride). Per tap this is 2/3 of
// the L2 -> LDS bytes (the bf16 128 x 128 tile needs ~39 TB/s of that path at the matrix peak; the chip delivers ~17), 2/3 of
// the DMA instructions and 2/3 of the LDS fragment reads of the per-tap form.
// FAST = 1: the launcher has checked that one K step advances a row by less than one frame ((BK / Wo + 1) <= Ho — every layer of
// the networks here), so the division form of the coordinate walk is not even compiled in (registers, code size).
template <int BMt, int BNt, int BK = 64, int NT = 1, int FAST = 0>   // BK = rows per K step: 64, or 32 (half the LDS: more blocks per CU)
__global__ __launch_bounds__(256, NT == 3 ? 2 : 1) void wgrad_bf16_kernel(const WgradParams p) {
  static_assert(BK == 64 || BK == 32, "K step of 64 or 32 rows");
  static_assert(NT == 1 || NT == 3, "one tap, or the three taps of a kernel row");
  // compile-time where the launcher's choice is fixed: a kernel-row block (NT = 3) never has 1x1 "simple" rows, and the
  // interleaved DMA issue is a probe-build switch only (shipped builds issue all pieces right after the barrier)
  const bool simple_rows = NT == 1 && p.simple_rows;
#ifdef R3M_PROBES
  const bool interleave = p.interleave != 0;
#else
  constexpr bool interleave = false;
#endif
  constexpr int WR = BK / 4;                                   // k rows staged per wave per stage
  constexpr int TM = BMt / 64, TN = BNt / 64;
  constexpr int A_ROWB = BMt * 2, B_ROWB = BNt * 2;            // bytes per k row
  constexpr int A_RPI = 1024 / A_ROWB, B_RPI = 1024 / B_ROWB;  // k rows per DMA instruction
  constexpr int AJ = WR / A_RPI, BJ = WR / B_RPI;              // instructions per wave per stage (per tap for B)
  constexpr int NP = AJ + BJ;                                  // piece GROUPS: a B group issues NT instructions
  constexpr int B_TILE = BK * B_ROWB;
  constexpr int STAGE = BK * A_ROWB + NT * B_TILE;
  __shared__ __attribute__((aligned(256))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int T = p.KH * p.KW;
  const int TG = T / NT;                                       // tap groups per tile (NT = 3: kernel rows)
  const int lid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int bx = lid % p.gx, by = lid / p.gx;   // by = split index: consecutive logical blocks read the same rows
  const int tap0 = (bx % TG) * NT;
  const int tile = bx / TG;
  const int tn_ = tile % p.tilesN, tm_ = tile / p.tilesN;
  const int co0 = tm_ * BMt, ci0 = tn_ * BNt;
  const int kh = tap0 / p.KW, kw0 = tap0 - kh * p.KW;
  const int ms = by * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);
  const int hw = p.Ho * p.Wo;
  const char* dYb = reinterpret_cast<const char*>(p.dY);
  const char* Xb = reinterpret_cast<const char*>(p.X);

  // swizzle key of a k row: 4 consecutive rows must land in 4 different 64-byte bank quarters
  //   128 wide (256-byte rows): key = row & 3;   64 wide (128-byte rows, two per bank line): key = (row >> 1) & 1
  // DMA lane -> (k row within the instruction, physical 16-byte slot) -> logical slot = physical ^ 4*key
  const int a_k = lane / (A_ROWB / 16), a_ps = lane % (A_ROWB / 16);
  const int b_k = lane / (B_ROWB / 16), b_ps = lane % (B_ROWB / 16);
  const int a_key = (BMt == 128) ? (a_k & 3) : ((a_k >> 1) & 1);
  const int b_key = (BNt == 128) ? (b_k & 3) : ((b_k >> 1) & 1);
  // Round 3: LDS DMA through buffer descriptors (conv_dev.h, buf_dma16). PMC on the ResNet-34 3x3 layers showed this kernel
  // ISSUE-bound, not bandwidth-bound: 6.8 vector instructions per MFMA (the per-lane 64-bit source pointers of global_load_lds:
  // 64-bit multiply-adds, pointer selects, a 64-bit advance per piece), waves 28 % issuing / 35 % stalled on dependent VALU / 37 %
  // parked, matrix pipe busy 0.37. With a descriptor the per-lane part is a 32-bit byte offset and out-of-range lanes read zeros:
  //   dY (and X of 1x1 stride-1 layers): CONSTANT per-lane offsets, the descriptor advances one K step on the scalar unit and
  //     rows past the split fall off its end — no vector instruction per piece;
  //   X of 3x3 / strided layers: per-lane (oy, ox, frame offset) walk in 32-bit arithmetic, padding taps get an out-of-range
  //     offset; rows past the split need no test (their dY rows are zeros).
  const unsigned a_chan = (unsigned)((co0 + (a_ps ^ (4 * a_key)) * 8) * 2);   // byte offset of this lane's 8 channels inside a dY row
  const unsigned b_chan = (unsigned)((ci0 + (b_ps ^ (4 * b_key)) * 8) * 2);

  const char* a_base = dYb + (long long)ms * p.Co * 2;          // descriptor: rows [ms + BK * step, me) of dY
  int a_left = (int)((long long)(me - ms) * p.Co * 2);
  const int a_stepb = BK * p.Co * 2;
  unsigned a_voff[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) a_voff[j] = (unsigned)((wave * WR + j * A_RPI + a_k) * p.Co * 2) + a_chan;

  const int q64 = BK / p.Wo, r64 = BK - q64 * p.Wo;          // (oy, ox) advance of one K step
  const bool fast_adv = FAST || (q64 + 1) <= p.Ho;
  const long long img = (long long)p.Hi * p.Wi * p.Ci * 2;
  const unsigned imgb = (unsigned)img;
  const int n0 = ms / hw;                                     // 3x3 / strided: offsets are relative to the split's first frame
  const char* b_base = simple_rows ? Xb + (long long)ms * p.Ci * 2 : Xb + (long long)n0 * img;
  int b_left;
  {
    const long long rest = simple_rows ? (long long)(me - ms) * p.Ci * 2 : (long long)(p.N - n0) * img;
    b_left = rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB;
  }
  const int b_stepb = BK * p.Ci * 2;
  const int pixb = p.Ci * 2;                                    // bytes between the X rows of neighbouring taps (one pixel)
  const int rowb = p.Wi * pixb;
  const int kh_p = kh - p.pad, kw_p = kw0 - p.pad;
  // 3x3 / strided layers: a lane tracks the INPUT coordinates of its row's output pixel (ys = oy * stride, xs = ox * stride) and
  // their byte position pos = ys * rowb + xs * pixb by additions only (a K step advances (oy, ox) by a block-uniform amount),
  // so there is no integer multiply in the K loop (v_mul_lo_u32 issues at a quarter of the rate: four per piece were ~60 cycles).
  const int xs_wrap = p.Wo * p.stride, ys_wrap = p.Ho * p.stride;
  const int adv_xs = r64 * p.stride, adv_ys = q64 * p.stride;
  const int adv_pos = adv_ys * rowb + adv_xs * pixb;
  const int wrapx_pos = p.stride * rowb - xs_wrap * pixb;       // ox wrapped: one output row down, Wo pixels back
  const int wrapy_pos = ys_wrap * rowb;                         // oy wrapped: next frame (b_off carries the frame)
  const int tap_pos = kh_p * rowb + kw_p * pixb;
  int b_m[BJ];
  unsigned b_off[BJ];      // simple rows: constant per-lane offset; otherwise byte offset of the row's frame from b_base (+ channels)
  int ys[BJ], xs[BJ], pos[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    b_m[j] = ms + wave * WR + j * B_RPI + b_k;
    ys[j] = 0; xs[j] = 0; pos[j] = 0;
    if (simple_rows) {
      b_off[j] = (unsigned)((wave * WR + j * B_RPI + b_k) * p.Ci * 2) + b_chan;
    } else {
      const int n = b_m[j] / hw;
      const int rem = b_m[j] - n * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      ys[j] = oy * p.stride; xs[j] = ox * p.stride;
      pos[j] = ys[j] * rowb + xs[j] * pixb;
      b_off[j] = (unsigned)(n - n0) * imgb + b_chan;
    }
  }

  // NT = 3 runs at the register limit: there the descriptor words are pinned to scalar registers (conv_dev.h buf_dma16_uniform)
  auto dma = [&](const char* base, int bytes, unsigned char* lds, unsigned voff) __attribute__((always_inline)) {
    if constexpr (NT == 3) buf_dma16_uniform(base, bytes, lds, voff);
    else buf_dma16(base, bytes, lds, voff);
  };
  auto issue_piece = [&](int stage, auto pc_c) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_c)::value;
    if constexpr (pc < AJ) {
      constexpr int j = pc;
      unsigned char* la = smem + stage * STAGE + (wave * WR + j * A_RPI) * A_ROWB;
      dma(a_base, a_left, la, a_voff[j]);
    } else {
      constexpr int j = pc - AJ;
      unsigned char* lb = smem + stage * STAGE + BK * A_ROWB + (wave * WR + j * B_RPI) * B_ROWB;
      if (simple_rows) {
        dma(b_base, b_left, lb, b_off[j]);
      } else {
        const int iy = ys[j] + kh_p, ix0 = xs[j] + kw_p;
        const bool rowok = (unsigned)iy < (unsigned)p.Hi;
        const unsigned off0 = b_off[j] + (unsigned)(pos[j] + tap_pos);           // garbage when the tap is padding: not used then
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bool in = rowok && ((unsigned)(ix0 + t) < (unsigned)p.Wi);
          dma(b_base, b_left, lb + t * B_TILE, in ? off0 + (unsigned)(t * pixb) : BUF_OOB);
        }
        if (FAST || fast_adv) {
          int x = xs[j] + adv_xs, y = ys[j] + adv_ys, ps = pos[j] + adv_pos;
          const bool cx = x >= xs_wrap;
          x = cx ? x - xs_wrap : x;
          y = cx ? y + p.stride : y;
          ps = cx ? ps + wrapx_pos : ps;
          const bool cy = y >= ys_wrap;
          y = cy ? y - ys_wrap : y;
          ps = cy ? ps - wrapy_pos : ps;
          b_off[j] = cy ? b_off[j] + imgb : b_off[j];
          xs[j] = x; ys[j] = y; pos[j] = ps;
        } else if constexpr (!FAST) {
          b_m[j] += BK;
          const int n = b_m[j] / hw;
          const int rem = b_m[j] - n * hw;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          ys[j] = oy * p.stride; xs[j] = ox * p.stride;
          pos[j] = ys[j] * rowb + xs[j] * pixb;
          b_off[j] = (unsigned)(n - n0) * imgb + b_chan;
        }
      }
    }
    if constexpr (pc == NP - 1) {             // after the last piece of a K step: the linear descriptors move on (scalar unit)
      a_base += a_stepb;
      a_left = a_left > a_stepb ? a_left - a_stepb : 0;
      if (simple_rows) {
        b_base += b_stepb;
        b_left = b_left > b_stepb ? b_left - b_stepb : 0;
      }
    }
  };

  f32x16 acc[NT][TM][TN];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.f;

  // transpose-read addressing. Lane l: group-local q = l & 15 addresses k row 8*(l>>5) + (q>>2) (+16s + 4r as an immediate),
  // channels [16*((l>>4)&1) + 4*(q&3), +4) of its MFMA tile; the tile's 64-byte group index is XORed with the row key.
  const int q = lane & 15;
  const int frow = 8 * (lane >> 5) + (q >> 2);
  const int fkeyA = (BMt == 128) ? (frow & 3) : ((frow >> 1) & 1);
  const int fkeyB = (BNt == 128) ? (frow & 3) : ((frow >> 1) & 1);
  const int fcol = (16 * ((lane >> 4) & 1) + 4 * (q & 3)) * 2;       // byte offset inside the tile's 64-byte group
  int fa_off[TM], fb_off[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) fa_off[t] = frow * A_ROWB + (((wm * TM + t) ^ fkeyA) * 64) + fcol;
#pragma unroll
  for (int t = 0; t < TN; ++t) fb_off[t] = BK * A_ROWB + frow * B_ROWB + (((wn * TN + t) ^ fkeyB) * 64) + fcol;

  auto tr_read = [&](const unsigned char* ptr) __attribute__((always_inline)) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
  };
  auto frag = [&](const unsigned char* base, int rowb, int sidx) __attribute__((always_inline)) {
    const s16x4 lo = tr_read(base + (16 * sidx) * rowb);
    const s16x4 hi = tr_read(base + (16 * sidx + 4) * rowb);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  auto mfma_stage = [&](const unsigned char* st, int dma_stage) __attribute__((always_inline)) {
    // the next K step's DMA: all pieces right after the barrier (p.interleave = 1: spread between the MFMA groups instead)
    if (dma_stage >= 0 && !interleave) static_for<NP>([&](auto pc) __attribute__((always_inline)) { issue_piece(dma_stage, pc); });
    static_for<BK / 16>([&](auto s_c) __attribute__((always_inline)) {
      constexpr int sidx = decltype(s_c)::value;
      bf16x8 a[TM];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = frag(st + fa_off[t], A_ROWB, sidx);
      if (dma_stage >= 0 && interleave) {
        constexpr int P0 = sidx * NP / (BK / 16), P1 = (sidx + 1) * NP / (BK / 16);
        static_for<P1 - P0>([&](auto q_c) __attribute__((always_inline)) {
          issue_piece(dma_stage, std::integral_constant<int, P0 + decltype(q_c)::value>{});
        });
      }
#pragma unroll
      for (int tp = 0; tp < NT; ++tp) {
        bf16x8 b[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = frag(st + tp * B_TILE + fb_off[t], B_ROWB, sidx);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tp][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tp][tm][tn], 0, 0, 0);
      }
    });
  };

  const int nk = (me - ms + BK - 1) / BK;
  if (nk > 0) static_for<NP>([&](auto pc) __attribute__((always_inline)) { issue_piece(0, pc); });
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    mfma_stage(smem + cur * STAGE, (kt + 1 < nk) ? (cur ^ 1) : -1);
  }

  float* out = p.out + (long long)by * p.Co * T * p.Ci;
  const int lrow = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int tp = 0; tp < NT; ++tp)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int ci = ci0 + (wn * TN + tn) * 32 + lrow;
          out[((long long)co * T + tap0 + tp) * p.Ci + ci] = acc[tp][tm][tn][r];
        }
      }
}

// =====================================================================================================
// 3x3 / stride 1 / pad 1 weight gradient with ALL NINE TAPS per block: dW[co, tap, ci] = sum_m dY[m, co] * X[m + shift(tap), ci].
// The per-tap kernel above stages a dY tile and an X tile for every tap (18 rows per contraction row); here a K step of 64
// pixels stages the dY rows once and ONE X window of 64 + 2W + 2 rows that all taps read at their row shift (3.9 rows per
// contraction row at 56x56, 2.5 at 14x14). Block = one 64 (co) x 64 (ci) tile x 9 taps, four waves of 32 x 32 x 9 (144
// accumulator registers per lane), split-K partials as before.
// The border rule varies ALONG the contraction (a pixel at x = 0 has no left neighbour: dx = -1 drops pixels with x = 0, dx = +1
// those with x = W-1, dy = -1 / +1 those with y = 0 / H-1; rows of other frames inside the window are exactly those the dy masks
// remove), so it cannot be a row redirect as in the forward halo kernel — the X fragments are masked in registers. A wave issues
// one instruction per 4 cycles whatever its kind, so the budget is ~135 instructions per k group (9 MFMAs): general per-element
// masks (8-bit drop masks expanded to 16-bit lanes: 550 instructions, even on the scalar unit) ran at 390 TFLOP/s. Restricted to
// image widths that are multiples of 8 the masks collapse to "element 0", "element 7" and "all" per lane half (see below).
// =====================================================================================================
__global__ __launch_bounds__(256, 2) void wgrad3x3_halo_bf16_kernel(const WgradParams p, const int xri) {
  constexpr int BK = 64;
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int stage_bytes = BK * 128 + xri * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int bx = lid % p.gx, by = lid / p.gx;          // by = split index: consecutive logical blocks read the same rows
  const int tn_ = bx % p.tilesN, tm_ = bx / p.tilesN;
  const int co0 = tm_ * 64, ci0 = tn_ * 64;
  const int ms = by * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);
  const int W = p.Wi, H = p.Hi;
  const char* dYb = reinterpret_cast<const char*>(p.dY);
  const char* Xb = reinterpret_cast<const char*>(p.X);

  // DMA lane -> (k row within the instruction, physical 16-byte slot); logical slot = physical ^ 4 * key, key = (row >> 1) & 1
  const int d_k = lane >> 3, d_ps = lane & 7;
  const int d_key = (d_k >> 1) & 1;
  const int a_cb = (co0 + (d_ps ^ (4 * d_key)) * 8) * 2;
  const int b_cb = (ci0 + (d_ps ^ (4 * d_key)) * 8) * 2;
  const char* zl = reinterpret_cast<const char*>(g_zero_bytes) + (lane & 15) * 16;
  int a_m[2];
  const char* a_ptr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    a_m[j] = ms + wave * 16 + j * 8 + d_k;
    a_ptr[j] = dYb + (long long)a_m[j] * p.Co * 2 + a_cb;
  }
  const long long a_step = (long long)BK * p.Co * 2;
  auto issue = [&](int stage, int mk) __attribute__((always_inline)) {
    unsigned char* st = smem + stage * stage_bytes;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dma16(sel_ptr(a_ptr[j], zl, a_m[j] < me), st + (wave * 16 + j * 8) * 128);
      a_m[j] += BK;
      a_ptr[j] += a_step;
    }
    for (int i = wave; i < xri; i += 4) {
      const long long q = (long long)mk - (W + 1) + 8 * i + d_k;
      const bool in = q >= 0 && q < (long long)p.M;
      const char* src = Xb + ((in ? q : 0) * p.Ci) * 2 + b_cb;
      dma16(sel_ptr(src, zl, in), st + BK * 128 + i * 1024);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // transpose-read addressing (see wgrad_bf16_kernel): lane -> k row 8*(lane>>5) + (q>>2), 4 channels of its 32-channel group
  const int q16 = lane & 15;
  const int frow = 8 * (lane >> 5) + (q16 >> 2);
  const int fcol = (16 * ((lane >> 4) & 1) + 4 * (q16 & 3)) * 2;
  const int fa_off = frow * 128 + ((wm ^ ((frow >> 1) & 1)) * 64) + fcol;
  int fb_off[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int rowb = frow + (W + 1) + (t / 3 - 1) * W + (t % 3 - 1);       // window row of this lane's first k row at tap t
    fb_off[t] = BK * 128 + rowb * 128 + ((wn ^ ((rowb >> 1) & 1)) * 64) + fcol;
  }
  auto tr_read = [&](const unsigned char* ptr) __attribute__((always_inline)) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
  };
  auto frag = [&](const unsigned char* base, int sidx) __attribute__((always_inline)) {
    const s16x4 lo = tr_read(base + (16 * sidx) * 128);
    const s16x4 hi = tr_read(base + (16 * sidx + 4) * 128);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  // image coordinates of the first pixel of the K step (pixel ms + kt*64): block-uniform, kept in scalar registers
  const int invW = 65536 / W + 1;                       // (v * invW) >> 16 == v / W for v < 4096
  int ox, oy;
  {
    const int row = ms / W;
    ox = __builtin_amdgcn_readfirstlane(ms - row * W);
    oy = __builtin_amdgcn_readfirstlane(row % H);
  }
  const int adv_q = BK / W, adv_r = BK - adv_q * W;
  const bool hi_half = lane >= 32;

  const int nk = (me - ms + BK - 1) / BK;
  if (nk > 0) issue(0, ms);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue(cur ^ 1, ms + (kt + 1) * BK);
    const unsigned char* st = smem + cur * stage_bytes;
    static_for<4>([&](auto s_c) __attribute__((always_inline)) {
      constexpr int sidx = decltype(s_c)::value;
      const bf16x8 a = frag(st + fa_off, sidx);
      // Border masks. W and H*W are multiples of 8 (launcher) and K steps start at multiples of 64 pixels, so the 8 consecutive
      // pixels a lane half holds never straddle an image row: x = 0 can only be its element 0, x = W-1 only its element 7, and
      // the first / last image row covers the run entirely or not at all. Four flags per half on the scalar unit, four per-lane
      // mask dwords, at most four ANDs per tap.
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      unsigned sL[2], sR[2], sT[2], sB[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int xs = ox + 16 * sidx + 8 * h, ys = oy;
        const int qd = (xs * invW) >> 16;
        xs -= qd * W;
        ys += qd;
        ys = ys >= H ? ys - H : ys;     // qd <= 7 (W >= 8, 64-pixel K steps) and H >= 4 (launcher): two wraps cover oy + qd < 3 H
        ys = ys >= H ? ys - H : ys;
        sL[h] = xs == 0 ? 0xffff0000u : 0xffffffffu;          // drop element 0 (left neighbour of x = 0)
        sR[h] = xs + 8 == W ? 0x0000ffffu : 0xffffffffu;      // drop element 7 (right neighbour of x = W-1)
        sT[h] = ys == 0 ? 0u : 0xffffffffu;                   // first image row: nothing above
        sB[h] = ys == H - 1 ? 0u : 0xffffffffu;               // last image row: nothing below
      }
      const unsigned mL = hi_half ? sL[1] : sL[0], mR = hi_half ? sR[1] : sR[0];
      const unsigned mT = hi_half ? sT[1] : sT[0], mB = hi_half ? sB[1] : sB[0];
      static_for<9>([&](auto t_c) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        constexpr int kh = t / 3, kw = t % 3;
        u32x4_t u = __builtin_bit_cast(u32x4_t, frag(st + fb_off[t], sidx));
        if constexpr (kh != 1) {
          const unsigned my = kh == 0 ? mT : mB;
          u[0] &= (kw == 0 ? (my & mL) : my);
          u[1] &= my;
          u[2] &= my;
          u[3] &= (kw == 2 ? (my & mR) : my);
        } else {
          if constexpr (kw == 0) u[0] &= mL;
          if constexpr (kw == 2) u[3] &= mR;
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, u), acc[t], 0, 0, 0);
      });
    });
    // next K step: 64 pixels further
    ox += adv_r;
    oy += adv_q;
    if (ox >= W) { ox -= W; ++oy; }
    while (oy >= H) oy -= H;
  }

  float* out = p.out + (long long)by * p.Co * 9 * p.Ci;
  const int lrow = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int ci = ci0 + wn * 32 + lrow;
      out[((long long)co * 9 + t) * p.Ci + ci] = acc[t][r];
    }
}

// 3x3 stride-1 weight gradients at image widths that are multiples of 8 go through the all-taps kernel (56x56 x 64 channels at
// 1280 frames: 0.834 -> 0.470 ms; ResNet-50 step -1.1 %, ResNet-34 -4.2 %). R3M_WG16_HALO=0: per-tap kernel
static int wg16_halo() {
  const int v = R3M_ENV_INT("R3M_WG16_HALO", 1);
  return v;
}

static inline int wgrad_halo_lds_bytes(int W) { return 2 * (64 * 128 + ceil_div(64 + 2 * W + 2, 8) * 1024); }

static bool wgrad_halo_eligible(const WgradParams& p) {
  // widths that are multiples of 8 (56 x 56: the 64-channel layers, where the per-tap kernel is furthest from its roof)
  return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Ho == p.Hi && p.Wo == p.Wi && (p.Wi & 7) == 0 && p.Wi >= 8 &&
         p.Hi >= 4 && p.Wi <= 1024 && wgrad_halo_lds_bytes(p.Wi) <= 80 * 1024;
}

static inline bool wg_wide(int Co, int Ci) { return (Co % 128 == 0) && (Ci % 128 == 0); }

// Round 3: 3-wide kernels on the 128 x 128 tile run one block per KERNEL ROW (three taps, dY staged and read once): ResNet-34's
// 128 / 256 / 512-channel 3x3 weight gradients, stride 1 and 2. R3M_WG16_ROWS=0 (probe builds): per-tap blocks.
static bool wg16_rows(int KW, bool wide) {
  const int v = R3M_ENV_INT("R3M_WG16_ROWS", 1);
  return v && wide && KW == 3;
}

// split-K factor: enough blocks to fill the chip ~4 (wide) / ~10 (narrow) times, rows per split a multiple of 64
int wgrad_bf16_pick_split(int M, int Co, int Ci, int T) {
  const bool wide = wg_wide(Co, Ci);
  const int tiles = wide ? (Co / 128) * (Ci / 128) * T : ceil_div(Co, 64) * ceil_div(Ci, 64) * T;
  const int tgt = R3M_ENV_INT("R3M_WG16_BLOCKS", 0);
  // the all-taps kernel runs one block per (tile, split) for all nine taps: 512 splits of the 64-channel layers fill the chip
  const int narrow_target = (T == 9 && wg16_halo()) ? 512 * 9 : 2560;
  int blocks_per_split = tiles > 0 ? tiles : 1;
  int wide_target = 1024;
  if (wg16_rows(T == 9 ? 3 : 0, wide)) {
    blocks_per_split = tiles / 3;     // `tiles` counts taps; a kernel-row block covers three
    wide_target = 512;                // exactly one round of the 2 blocks a CU holds (same box: 802 -> 835, 717 -> 750 TFLOP/s vs 1024)
  }
  int split = (wide ? (tgt > 0 ? tgt : wide_target) : narrow_target) / blocks_per_split;
  const int max_split = ceil_div(M, 256);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  return split;
}

int launch_wgrad_bf16(const WgradParams& p0, int splitK, hipStream_t s) {
  WgradParams p = p0;
  R3M_REQUIRE(p.Co % 64 == 0 && p.Ci % 64 == 0, "wgrad(bf16): Co=%d, Ci=%d must be multiples of 64", p.Co, p.Ci);
  const int T = p.KH * p.KW;
  const bool wide = wg_wide(p.Co, p.Ci);
  p.rows_per_split = ceil_div(ceil_div(p.M, splitK), 64) * 64;
  p.tilesN = wide ? p.Ci / 128 : p.Ci / 64;
  const int tilesM = wide ? p.Co / 128 : p.Co / 64;
  const double flops = 2.0 * (double)p.M * p.Co * (double)p.Ci * T;
  prof_begin(wide ? KC_WGRAD_WIDE : KC_WGRAD_NARROW, flops, p.M, p.Co, p.Ci, T, s);
  p.gx = tilesM * p.tilesN * T;
  {   // buffer addressing: a block's operands are reached through 32-bit offsets from the first row / frame of its split
    const long long lim = 0x7FFFF000LL;
    const long long a_span = (long long)p.rows_per_split * p.Co * 2;
    const long long frames = (long long)p.rows_per_split / ((long long)p.Ho * p.Wo) + 2;
    const long long b_span = p.simple_rows ? (long long)p.rows_per_split * p.Ci * 2 : frames * p.Hi * p.Wi * p.Ci * 2;
    R3M_REQUIRE(a_span < lim && b_span < lim, "wgrad(bf16): one split spans %lld / %lld bytes (limit 2 GiB): raise splitK (%d)", a_span, b_span, splitK);
  }
  {
    const int il = R3M_ENV_INT("R3M_WG_INTERLEAVE", 0);
    p.interleave = il;
    const int xc = R3M_ENV_INT("R3M_WG_XCD", 1);
    p.xcd = xc;
  }
  if (wg16_halo() && wgrad_halo_eligible(p)) {
    p.tilesN = p.Ci / 64;
    p.gx = (p.Co / 64) * p.tilesN;
    const int lds = wgrad_halo_lds_bytes(p.Wi);
    static DynLdsOptIn optin;
    if (int e = ensure_dyn_lds(optin, reinterpret_cast<const void*>(wgrad3x3_halo_bf16_kernel), lds, "wgrad3x3_halo(bf16)")) return e;
    hipLaunchKernelGGL(wgrad3x3_halo_bf16_kernel, dim3(p.gx * splitK), dim3(256), lds, s, p, ceil_div(64 + 2 * p.Wi + 2, 8));
    prof_bytes(2.0 * ((double)p.M * p.Co + (double)p.N * p.Hi * p.Wi * p.Ci) + 4.0 * (double)splitK * p.Co * T * p.Ci);
    prof_end(s);
    return check_launch("wgrad3x3_halo_bf16");
  }
  // K steps of 32 rows for the 128x128 tile (32 KB of stages instead of 64: -16 % measured over ResNet-50), 64 rows for the
  // 64x64 tile (32 rows measured +5 % there). R3M_WG16_BK=64 / =32 forces one step size on both (experiments).
  const int bk = R3M_ENV_INT("R3M_WG16_BK", 0);
  const bool bk32 = bk == 32 || (bk != 64 && wide);
  if (wg16_rows(p.KW, wide)) {   // 3-wide kernels on the 128 x 128 tile: one block = the three taps of a kernel row
    p.gx = tilesM * p.tilesN * p.KH;
    if ((32 / p.Wo + 1) <= p.Ho) hipLaunchKernelGGL((wgrad_bf16_kernel<128, 128, 32, 3, 1>), dim3(p.gx * splitK), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<128, 128, 32, 3>), dim3(p.gx * splitK), dim3(256), 0, s, p);
    prof_bytes(2.0 * ((double)p.M * p.Co + (double)p.N * p.Hi * p.Wi * p.Ci) + 4.0 * (double)splitK * p.Co * T * p.Ci);
    prof_end(s);
    return check_launch("wgrad_bf16 (kernel rows)");
  }
  const dim3 grid(p.gx * splitK);
  if (wide && bk32) hipLaunchKernelGGL((wgrad_bf16_kernel<128, 128, 32>), grid, dim3(256), 0, s, p);
  else if (wide) hipLaunchKernelGGL((wgrad_bf16_kernel<128, 128>), grid, dim3(256), 0, s, p);
  else if (bk32) hipLaunchKernelGGL((wgrad_bf16_kernel<64, 64, 32>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((wgrad_bf16_kernel<64, 64>), grid, dim3(256), 0, s, p);
  prof_bytes(2.0 * ((double)p.M * p.Co + (double)p.N * p.Hi * p.Wi * p.Ci) + 4.0 * (double)splitK * p.Co * T * p.Ci);
  prof_end(s);
  return check_launch("wgrad_bf16");
}

// ---- weight images: fp32 master [Co][T][Ci] -> bf16 copy (forward) / bf16 [Ci][T][Co] (dgrad) ----
__global__ __launch_bounds__(256) void convert_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) st4t(dst + i * 4, ld4t(src + i * 4));
}

int launch_convert_bf16(const float* src, void* dst, long long n, hipStream_t s) {
  R3M_REQUIRE(n % 4 == 0, "convert_bf16: n=%lld must be a multiple of 4", n);
  hipLaunchKernelGGL(convert_bf16_kernel, dim3(ceil_div(n / 4, 256)), dim3(256), 0, s, src, reinterpret_cast<bf16_t*>(dst), n / 4);
  return check_launch("convert_bf16");
}

__global__ __launch_bounds__(256) void transpose_w_bf16_kernel(const float* __restrict__ W, bf16_t* __restrict__ Wt, int Co, int T, int Ci) {
  // Wt[ci][t][co] = W[co][t][ci]; a 32 x 32 (co, ci) tile per block through LDS, both sides coalesced
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Co && ci < Ci) ? W[((long long)co * T + t) * Ci + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Ci && co < Co) Wt[((long long)ci * T + t) * Co + co] = (bf16_t)tile[tx][r];
  }
}

int launch_transpose_w_bf16(const float* W, void* Wt, int Co, int T, int Ci, hipStream_t s) {
  hipLaunchKernelGGL(transpose_w_bf16_kernel, dim3(ceil_div(Ci, 32), ceil_div(Co, 32), T), dim3(256), 0, s, W,
                     reinterpret_cast<bf16_t*>(Wt), Co, T, Ci);
  return check_launch("transpose_w_bf16");
}

}  // namespace r3m
