// r3m_amd — bf16 3x3 / stride 1 / pad 1 convolutions (forward and dgrad) of the 128-multiple-wide layers, round 6:
// one PERSISTENT eight-wave block per CU, kernel-row K steps, double-buffered 32-channel windows.
// Call site in the reference: the torchvision BasicBlock / Bottleneck conv3x3 behind /root/reference/r3m/models/models_r3m.py:99.
//
// What the per-tile halo kernel (conv_bf16.hip, conv3x3_halo_bf16_kernel) left on the table:
//   * one tap = 32 MFMAs (1024 matrix cycles) per wave between two `s_waitcnt vmcnt(0)` + `s_barrier`, each followed by a scalar
//     table load and ~65 vector address instructions before the first fragment can be requested;
//   * ~7 non-MFMA instructions per MFMA (XOR-swizzled 128-byte LDS rows: every fragment address is computed);
//   * ONE window buffer: every 64-channel chunk ends with an exposed HBM round trip that only the CU's second block covers;
//   * 16 KB of weights staged per 256 x 128 x 64 MACs: 20 B/clk/CU of L2 -> LDS traffic at the MFMA peak.
// This kernel:
//   * tile 512 x 128, eight waves as 4 x 2 with 128 x 64 wave tiles (0.75 ds_read_b128 per MFMA), one block per CU;
//   * K step = one KERNEL ROW of one 32-channel chunk: 3 taps x 2 MFMA K groups x 8 = 48 MFMAs (1536 matrix cycles) per wave — with
//     two waves per SIMD 3072 cycles per SIMD — between barriers; weight stages of 3 x 128 rows in a ring of two;
//   * LDS rows are PADDED, not swizzled: 32 channels = 64 B + 16 B pad = 80 B (stride 20 banks: the 16 lanes of a ds_read_b128
//     group start at 16 distinct multiples of 4 banks, whatever common row shift a tap adds). The pad is produced by the DMA
//     itself: a 1 KiB LDS-DMA piece is 64 x 16 B units, unit u belongs to row u / 5, slot u % 5, and slot 4 fetches an
//     out-of-range offset. Fragment addresses are then `base(tap, row tile) + immediate`: the 36 bases of a tile (9 taps x 4 row
//     tiles) are computed once per tile and every chunk, kernel row, K group, window buffer and weight stage rides in the
//     ds_read offset field — the main loop has no vector ALU instruction at all;
//   * two window buffers (chunk c + 1 arrives under chunk c; at a tile's last chunk: the next tile's first window) and the next
//     tile's first weight stage requested during the last step: no tile starts cold;
//   * the epilogue's transposition slabs live in the window / weight buffers of odd parity, which are idle after a tile's last
//     step (the channel count is a multiple of 64, so the last chunk and the last step have odd parity) while the even ones fill.
// Accumulation order (32-channel chunk, kernel row, tap, K group of 16) differs from the halo kernel's (64-channel chunk, tap, K
// group): results agree to fp32 round-off of the accumulators, i.e. <= 1 bf16 ulp on stored elements (tests/test_gpu_bf16.py).
// The order does not depend on M, so plans of different frame counts produce the same rows.
#include "common.h"
#include "conv_dev.h"

namespace r3m {

namespace {

constexpr int R_BM = 512;
constexpr int R_RB = 80;                       // bytes per LDS row: 32 bf16 + 16 B pad
constexpr unsigned R_PAD = 0xC0000000u;        // offset of a pad unit: stays out of range after any +-1 GiB adjustment
constexpr int R_DUMMY = 160 * 1024 - 1024;     // the CU's last KiB: where DMA pieces without a destination land (all lanes out of range)

// BN = 128: waves 4 x 2, wave tile 128 x 64, windows of <= 45 KiB (W <= 30: every 128-multiple-wide 3x3 layer of the ResNets)
// BN = 64 : waves 8 x 1, wave tile  64 x 64, windows of <= 49 KiB (W <= 56: layer1 of ResNet-18 / 34, conv2 of ResNet-50's layer1)
template <int BN>
struct RowCfg {
  static constexpr int WN = BN / 64, WM = 8 / WN, TM = R_BM / WM / 32, TN = 2;
  static constexpr int ZOFF = (BN == 128 ? 45 : 49) * 1024;   // zero region of a window buffer, behind its pieces
  static constexpr int WB = ZOFF + 512;                       // window buffer stride
  static constexpr int PPT = BN * 5 / 64;                     // DMA pieces per tap of a weight stage (10 / 5)
  static constexpr int SB = 3 * BN * R_RB;                    // weight stage: 3 taps x BN columns x 80 B
  static constexpr int S0 = 2 * WB;                           // LDS map: [W0][W1][S0][S1][spare ... dummy KiB]
  static constexpr int END = 2 * WB + 2 * SB;
  static constexpr int NWS = (ZOFF / 1024 + 7) / 8;           // window pieces per wave and chunk (6 / 7) ...
  static constexpr int NW0 = (NWS + 1) / 2, NW1 = NWS - NW0;  // ... issued during the chunk's first / second kernel row
  static constexpr int NSW = (3 * PPT + 7) / 8;               // weight pieces per wave and step (4 / 2)
  static constexpr int RED = R_DUMMY - 4096;                  // BN = 64: scratch of the statistics combine
  static_assert(END <= RED, "one CU's LDS");
  static_assert(WB + 2 * 32 + 16 <= 65535 && SB + 2 * BN * R_RB + 32 * R_RB + 32 + 16 <= 65535, "buffer parities ride in the ds_read offset field");
};

template <int N>
__device__ __forceinline__ void r_wait_barrier() {
  // the wave's DMA pieces except the youngest N have landed (loads retire in order), all its LDS reads are done, then the block
  // barrier. Not __syncthreads(): its fence would drain vmcnt.
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int BN, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3_row_bf16_kernel(const GatherGemmParams p, const int npw, const int ntiles) {
  using C = RowCfg<BN>;
  constexpr int TM = C::TM, TN = C::TN, WN = C::WN;
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane & 31, lh = lane >> 5;
  const int W = p.Wi, HW = p.Hi * p.Wi;
  const int gridN = p.Nc / BN;
  const int G = gridDim.x;
  const int rowb = p.Ci * 2;                    // bytes of one pixel (A) / one tap of one column (B)
  const int nc32 = p.Ci >> 5;
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);
  const int b_bytes = p.Nc * p.T * rowb;

  if (tid < 64) *reinterpret_cast<uint4*>(smem + (tid >> 5) * C::WB + C::ZOFF + (tid & 31) * 16) = make_uint4(0u, 0u, 0u, 0u);

  // the nine taps as scalars: (dy & 255) | (dx & 255) << 8 | weight tap << 16
  int pk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) pk[k] = p.tap[k];

  // Weight pieces: a stage is 3 taps x PPT pieces (unit u = 64 * piece-in-tap + lane -> column u / 5, slot u % 5, slot 4 = pad).
  // Every wave issues the same NSW instructions per step (no branch splits the MFMA stream they are spread over):
  //   BN = 128: of every tap, wave w stages piece w (three pieces, their tap a compile-time property of the issue site); the fourth is
  //             piece 8 + (w & 1) of tap w >> 1 (waves 6, 7: piece w of the first tap once more — the same bytes to the same place);
  //   BN = 64 : pieces w and 8 + w of the stage's 15 (wave 7: piece 14 twice); which tap they belong to depends on the wave.
  // Weight-tap byte offsets that depend on the wave are loaded once into scalars: wsl[slot][kernel row].
  constexpr int NDYN = BN == 128 ? 1 : 2;       // slots whose tap depends on the wave
  unsigned bv[2];
  int gdyn[NDYN], wsl[NDYN][3];
  {
    int q[2];
    if constexpr (BN == 128) {
      gdyn[0] = wave < 6 ? (wave >> 1) * 10 + 8 + (wave & 1) : wave;
      q[0] = wave;
      q[1] = wave < 6 ? 8 + (wave & 1) : wave;
    } else {
      gdyn[0] = wave;
      gdyn[1] = wave < 7 ? 8 + wave : 14;
      q[0] = gdyn[0] % C::PPT;
      q[1] = gdyn[1] % C::PPT;
    }
#pragma unroll
    for (int sI = 0; sI < NDYN; ++sI)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) wsl[sI][kh] = __builtin_amdgcn_readfirstlane((p.tap[3 * kh + gdyn[sI] / C::PPT] >> 16) * rowb);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned u = 64u * (unsigned)q[i] + (unsigned)lane;
      const unsigned col = (u * 52429u) >> 18;   // u / 5 for u < 2^16
      const unsigned sl = u - 5u * col;
      bv[i] = sl < 4u ? col * (unsigned)(p.T * rowb) + sl * 16u : R_PAD;
    }
  }

  // window descriptor of row tile mt: base pointer / byte count of the buffer resource and the byte offset of the rows in front of the tensor
  struct Win { const char* base; int bytes; unsigned skipb; };
  auto win_of = [&](int mt) __attribute__((always_inline)) {
    Win t;
    const long long hb = (long long)mt * R_BM - (W + 1);      // pixel staged in window row 0
    const long long hb0 = hb > 0 ? hb : 0;
    t.base = Ab + hb0 * rowb;
    const long long rest = ((long long)p.M - hb0) * rowb;
    t.bytes = rest <= 0 ? 0 : (rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
    t.skipb = (unsigned)((hb0 - hb) * rowb);                  // rows in front of the tensor: their offsets wrap out of range
    return t;
  };
  // Window piece I of this wave = piece wave + 8 I of the window of 32-channel chunk c (tile described by base / bytes / skipb) into
  // buffer `buf`; unit u = 64 * piece + lane belongs to window row u / 5, slot u % 5 (slot 4 = pad). Pieces past the window's last go
  // to the dummy KiB with out-of-range lanes. Offsets are recomputed at every issue (6 vector instructions per piece, six or seven
  // pieces per chunk): kept in registers across the loop the compiler spilled them to scratch, and a scratch reload is a memory
  // round trip in the middle of a step. `zc` is zero at run time but depends on the chunk counter: it keeps the computation inside
  // the chunk loop without an asm statement (which would also cut the scheduling region the DMA pieces are spread over).
  auto issue_win_piece = [&](auto i_c, const char* a_base, int a_bytes, unsigned skipb, int c, int buf, int zc) __attribute__((always_inline)) {
    constexpr int I = decltype(i_c)::value;
    const int lane_o = lane + zc, wave_o = wave + zc;
    const int pi = wave_o + 8 * I;
    const bool live = pi < npw;
    const unsigned u = 64u * (unsigned)pi + (unsigned)lane_o;
    const unsigned r = (u * 52429u) >> 18;
    const unsigned sl = u - 5u * r;
    const unsigned vo = (sl < 4u && live) ? r * (unsigned)rowb + sl * 16u - skipb : R_PAD;
    unsigned char* dst = smem + (live ? buf * C::WB + pi * 1024 : R_DUMMY);
    buf_dma16_uniform(a_base, a_bytes, dst, vo, __builtin_amdgcn_readfirstlane(c * 64));
  };
  // Weight piece I (0 .. NSW - 1) of this wave for the stage of (chunk c, kernel row KHN), column tile n0, into stage `st`
  // (readfirstlane: under scalar-register pressure the compiler parks uniform values in vector registers and would wrap the DMA in a waterfall loop)
  auto issue_w_piece = [&](auto i_c, auto khn_c, int n0, int c, int st, int zc) __attribute__((always_inline)) {
    constexpr int I = decltype(i_c)::value, KHN = decltype(khn_c)::value;
    const int wave_o = wave + zc;                  // (LDS destinations computed here, not hoisted into live scalars)
    const int so = n0 * p.T * rowb + c * 64;
    unsigned char* stage = smem + C::S0 + st * C::SB;
    if constexpr (BN == 128 && I < 3) {
      buf_dma16(Bb, b_bytes, stage + (I * 10 + wave_o) * 1024, bv[0], __builtin_amdgcn_readfirstlane(so + (pk[3 * KHN + I] >> 16) * rowb));
    } else {
      constexpr int D = BN == 128 ? 0 : I;
      buf_dma16(Bb, b_bytes, stage + (gdyn[D] + zc) * 1024, bv[BN == 128 ? 1 : I], __builtin_amdgcn_readfirstlane(so + wsl[D][KHN]));
    }
  };

  // tiles tix, tix + G, ...: (row tile, column tile) advance by (gq, gr) with a carry — no division in the loop
  int tix = xcd_remap(blockIdx.x, G);
  if (tix >= ntiles) return;
  int cmt = tix / gridN, cnt = tix - cmt * gridN;
  const int gq = G / gridN, gr = G - gq * gridN;
  {
    const Win w0 = win_of(cmt);
    static_for<C::NWS>([&](auto i_c) __attribute__((always_inline)) { issue_win_piece(i_c, w0.base, w0.bytes, w0.skipb, 0, 0, 0); });
    static_for<C::NSW>([&](auto i_c) __attribute__((always_inline)) { issue_w_piece(i_c, std::integral_constant<int, 0>{}, cnt * BN, 0, 0, 0); });
  }

  const int bfrag0 = C::S0 + (wn * 64 + lrow) * R_RB + lh * 16;

  while (true) {
    // per-tile fragment bases: window row of (tap, row tile) or, where the tap falls outside the image (or the row outside the
    // tensor), the position of the zero region that shares its bank offset. Two 16-bit LDS offsets per register (taps k, k + 1 of
    // the first eight; the ninth alone).
    unsigned addrA[5][TM];
    {
      int lrow_o = lrow;
      asm volatile("" : "+v"(lrow_o));     // everything below is recomputed per tile: hoisted, the tile-invariant candidates were spilled to scratch
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int r = wm * (TM * 32) + t * 32 + lrow_o;
        const int m = cmt * R_BM + r;
        const int rem = m % HW;
        const int y = rem / W, x = rem - y * W;
        const int xc = (r + W + 1) * R_RB + lh * 16;
        const int vy = m < p.M ? y : -4, vx = x;               // rows past M: no tap is inside the image
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int dy = (pk[k] << 24) >> 24, dx = (pk[k] << 16) >> 24;
          const bool ok = ((unsigned)(vy + dy) < (unsigned)p.Hi) & ((unsigned)(vx + dx) < (unsigned)W);
          const int xa = xc + (dy * W + dx) * R_RB;
          const unsigned a16 = (unsigned)(ok ? xa : C::ZOFF + (xa & 255));
          if (k & 1) addrA[k >> 1][t] |= a16 << 16;
          else addrA[k >> 1][t] = a16;
        }
      }
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int tnext = tix + G;
    const bool has_next = tnext < ntiles;
    int nmt = cmt + gq, nnt = cnt + gr;                    // the next tile of this block
    if (nnt >= gridN) { nnt -= gridN; ++nmt; }
    if (!has_next) { nmt = cmt; nnt = cnt; }               // (none: the last chunk re-requests this tile's first window and stage — harmless)

    for (int cc = 0; cc < nc32; cc += 2) {
      static_for<2>([&](auto par_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        const int c = cc + PAR;
        const int zc = c >> 24;                             // 0 (see issue_win_piece)
        const bool more = c + 1 < nc32;                     // another chunk of this tile follows
        // what is requested during this chunk: the window of the next chunk (or the next tile's first), and at the last kernel row
        // the first weight stage of that chunk
        const Win nw = win_of(more ? cmt : nmt);
        const int nc = more ? c + 1 : 0;
        const int nn0 = (more ? cnt : nnt) * BN;
        static_for<3>([&](auto kh_c) __attribute__((always_inline)) {
          constexpr int KH = decltype(kh_c)::value;
          constexpr int SP = (PAR + KH) & 1;               // stage of step 3 c + KH (c has parity PAR)
          // This step's weights have landed — behind them only the window pieces issued during the previous step may still fly
          // (KH == 0: the next window has landed too) — and the other stage / window buffer is free.
          if (R3M_PROBE(p) & 16) {}
          else if constexpr (KH == 0) r_wait_barrier<0>();
          else if constexpr (KH == 1) r_wait_barrier<C::NW0>();
          else r_wait_barrier<C::NW1>();
          // DMA of this step, spread over its six MFMA groups: weight pieces first (they are needed at the next barrier), then the
          // window pieces of the chunk's first two kernel rows
          constexpr int WPG = (C::NSW + 1) / 2;             // weight pieces in groups 0 and 1 (2 + 2 / 1 + 1)
          constexpr int NWK = KH == 0 ? C::NW0 : (KH == 1 ? C::NW1 : 0), WK0 = KH == 0 ? 0 : C::NW0;
          auto dma = [&](auto gi_c) __attribute__((always_inline)) {
            constexpr int GI = decltype(gi_c)::value;
            if (R3M_PROBE(p) & 2) return;
            static_for<WPG>([&](auto q_c) __attribute__((always_inline)) {
              constexpr int I = GI * WPG + decltype(q_c)::value;
              if constexpr (GI < 2 && I < C::NSW) {
                if constexpr (KH < 2) issue_w_piece(std::integral_constant<int, I>{}, std::integral_constant<int, KH + 1>{}, cnt * BN, c, SP ^ 1, zc);
                else issue_w_piece(std::integral_constant<int, I>{}, std::integral_constant<int, 0>{}, nn0, nc, SP ^ 1, zc);
              }
            });
            if constexpr (GI >= 2 && GI - 2 < NWK) issue_win_piece(std::integral_constant<int, WK0 + GI - 2>{}, nw.base, nw.bytes, nw.skipb, nc, PAR ^ 1, zc);
          };
          // 3 taps x 2 K groups of 16 channels; fragments of group i + 1 requested before the MFMAs of group i
          bf16x8 fa[2][TM], fb[2][TN];
          auto load = [&](auto set_c, auto grp_c) __attribute__((always_inline)) {
            constexpr int S = decltype(set_c)::value, GI = decltype(grp_c)::value;
            constexpr int KW = GI >> 1, GG = GI & 1, K = 3 * KH + KW;
            if ((R3M_PROBE(p) & 4) && (GI || KH || PAR)) return;      // probe: the very first fragments forever
#pragma unroll
            for (int t = 0; t < TM; ++t) {
              const unsigned a = (K & 1) ? (addrA[K >> 1][t] >> 16) : (addrA[K >> 1][t] & 0xffffu);
              fa[S][t] = *reinterpret_cast<const bf16x8*>(smem + a + (PAR * C::WB + GG * 32));
            }
#pragma unroll
            for (int t = 0; t < TN; ++t)
              fb[S][t] = *reinterpret_cast<const bf16x8*>(smem + bfrag0 + (SP * C::SB + KW * (BN * R_RB) + t * (32 * R_RB) + GG * 32));
          };
          load(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
          static_for<6>([&](auto gi_c) __attribute__((always_inline)) {
            constexpr int GI = decltype(gi_c)::value;
            if constexpr (GI + 1 < 6) load(std::integral_constant<int, (GI + 1) & 1>{}, std::integral_constant<int, GI + 1>{});
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
              for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[GI & 1][tm], fb[GI & 1][tn], acc[tm][tn], 0, 0, 0);
            dma(gi_c);
            // issue order: one fragment read of the NEXT group behind each MFMA of this group (the scheduler would otherwise sink
            // every read to just in front of its first use and wait out an LDS round trip per MFMA pair), the group's DMA pieces
            // in front of its last MFMAs
            constexpr int NM = TM * TN, ND = GI + 1 < 6 ? TM + TN : 0;
            constexpr int NV = GI < 2 ? (GI * WPG + WPG <= C::NSW ? WPG : (GI * WPG < C::NSW ? C::NSW - GI * WPG : 0)) : (GI - 2 < NWK ? 1 : 0);
            constexpr int PAIRS = ND < NM - 1 ? ND : NM - 1;
#pragma unroll
            for (int q = 0; q < PAIRS; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (ND > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, ND - PAIRS, 0);
            if constexpr (NV > 0) __builtin_amdgcn_sched_group_barrier(0x020, NV, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM - PAIRS, 0);
          });
        });
      });
    }
    r_wait_barrier<0>();   // every wave is done with the odd window buffer and weight stage (they hold the store slabs now); nothing of this wave in flight

    if (R3M_PROBE(p) & 1) {   // probe: no epilogue (one store keeps the accumulators alive)
      if (acc[0][0][0] + acc[TM - 1][TN - 1][3] == 123.456f) reinterpret_cast<float*>(p.out)[0] = 1.f;
    } else {
      if constexpr ((EPI & EPI_STATS) != 0) {
        if constexpr (BN == 128) {
          gg_stats<R_BM, BN, C::WM, WN, true>(p, acc, nullptr, cnt * BN, cmt);        // one partial row per wave row (128 result rows)
        } else {
          // 64-wide outputs: one partial row per 256 result rows (the geometry gather_gemm_grid_m promises): four waves combine
          float* red = reinterpret_cast<float*>(smem + C::RED);                        // [8 waves][2][64]
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float v = acc[tm][tn][r];
                sm += v;
                sq = fmaf(v, v, sq);
              }
            sm += __shfl_xor(sm, 32);
            sq += __shfl_xor(sq, 32);
            if (lane < 32) {
              red[(wave * 2 + 0) * 64 + tn * 32 + lane] = sm;
              red[(wave * 2 + 1) * 64 + tn * 32 + lane] = sq;
            }
          }
          __syncthreads();
          if (tid < 128) {
            const int half = tid >> 6, col = tid & 63;
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              sm += red[((half * 4 + w) * 2 + 0) * 64 + col];
              sq += red[((half * 4 + w) * 2 + 1) * 64 + col];
            }
            const long long prow = (long long)cmt * 2 + half;
            if (prow * 256 < p.M) {
              p.stats[(prow * 2 + 0) * p.Nc + col] = sm;
              p.stats[(prow * 2 + 1) * p.Nc + col] = sq;
            }
          }
        }
      }
      // store slabs: waves 0-4 in W1, waves 5-7 in S1 and the spare LDS behind it (gg_store_bf16 addresses `base + wave * slab`)
      constexpr bool RMW = (EPI & (EPI_ACCUM | EPI_MASKED_ADD)) != 0;
      constexpr int SLAB = RMW ? 32 * (64 + 4) * 4 : 64 * (64 + 8) * 2;
      static_assert(5 * SLAB <= C::ZOFF && C::S0 + C::SB + 3 * SLAB <= C::RED, "store slabs fit the idle buffers");
      unsigned char* sb = wave < 5 ? smem + C::WB : smem + C::S0 + C::SB - 5 * SLAB;
      gg_store_bf16<R_BM, BN, C::WM, WN, EPI, (1 << 20)>(p, acc, reinterpret_cast<float*>(sb), cmt * R_BM, cnt * BN);
    }
    if (!has_next) break;
    tix = tnext;
    cmt = nmt;
    cnt = nnt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the harmless last requests still target this block's LDS: land before the block retires
}

int row16_cu_count() {
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

template <int BN, int EPI>
int row16_launch_one(const GatherGemmParams& p, hipStream_t s) {
  const int npw = ceil_div((R_BM + 2 * p.Wi + 2) * 5, 64);
  const long long tiles_ll = (long long)ceil_div(p.M, R_BM) * (p.Nc / BN);
  R3M_REQUIRE(tiles_ll < 0x7FFFFFFFLL, "conv3x3_row(bf16): too many tiles");
  const int ntiles = (int)tiles_ll;
  const int cus = row16_cu_count();
  const int grid = ntiles < cus ? ntiles : cus;
  auto kern = conv3x3_row_bf16_kernel<BN, EPI>;
  static DynLdsOptIn optin;
  if (int e = ensure_dyn_lds(optin, reinterpret_cast<const void*>(kern), 160 * 1024, "conv3x3_row(bf16)")) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, s, p, npw, ntiles);
  return 0;
}

template <int BN>
int row16_launch(const GatherGemmParams& p, hipStream_t s) {
  switch (p.flags) {
    case 0: return row16_launch_one<BN, 0>(p, s);
    case EPI_STATS: return row16_launch_one<BN, EPI_STATS>(p, s);
    case EPI_ACCUM: return row16_launch_one<BN, EPI_ACCUM>(p, s);
    case EPI_MASKED_ADD: return row16_launch_one<BN, EPI_MASKED_ADD>(p, s);
    case EPI_BNRED: return row16_launch_one<BN, EPI_BNRED>(p, s);
    case EPI_BNRED | EPI_MASKED_ADD: return row16_launch_one<BN, EPI_BNRED | EPI_MASKED_ADD>(p, s);
    case EPI_AFFINE | EPI_RELU: return row16_launch_one<BN, EPI_AFFINE | EPI_RELU>(p, s);                               // inference forward
    case EPI_AFFINE | EPI_ACCUM | EPI_RELU: return row16_launch_one<BN, EPI_AFFINE | EPI_ACCUM | EPI_RELU>(p, s);
    default: set_last_error("conv3x3_row(bf16): unsupported epilogue flag combination %d", p.flags); return 1;
  }
}

int g_row16_mode = 1;     // r3m_debug_set_conv3x3_bf16: 0 = the per-tile halo kernels everywhere

}  // namespace

int row16_set_mode(int mode) {
  const int old = __atomic_exchange_n(&g_row16_mode, mode, __ATOMIC_RELAXED);
  return old;
}

// 3x3 / stride 1 / pad 1, dense NHWC rows, input channels a multiple of 64; 128-multiple-wide outputs at W <= 30, 64-wide at W <= 56
bool row16_eligible(const GatherGemmParams& p) {
  if (!__atomic_load_n(&g_row16_mode, __ATOMIC_RELAXED)) return false;
  if (p.ntaps != 9 || p.simple_rows || p.is != 1 || p.os != 1 || p.ooy != 0 || p.oox != 0) return false;
  if (p.Hg != p.Hi || p.Wg != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi || (p.Ci & 63)) return false;
  if (!((p.Nc & 127) == 0 || p.Nc == 64)) return false;
  if (ceil_div((R_BM + 2 * p.Wi + 2) * 5, 64) * 1024 > ((p.Nc & 127) == 0 ? RowCfg<128>::ZOFF : RowCfg<64>::ZOFF)) return false;
  if (p.flags != 0 && p.flags != EPI_STATS && p.flags != EPI_ACCUM && p.flags != EPI_MASKED_ADD && p.flags != EPI_BNRED &&
      p.flags != (EPI_BNRED | EPI_MASKED_ADD) && p.flags != (EPI_AFFINE | EPI_RELU) && p.flags != (EPI_AFFINE | EPI_ACCUM | EPI_RELU))
    return false;
  if ((long long)p.Nc * p.T * p.Ci * 2 >= (long long)BUF_OOB) return false;
  for (int k = 0; k < 9; ++k)
    if (p.dy[k] < -1 || p.dy[k] > 1 || p.dx[k] < -1 || p.dx[k] > 1) return false;
  return true;
}

int launch_conv3x3_row_bf16(const GatherGemmParams& p, hipStream_t s) {
  return (p.Nc & 127) == 0 ? row16_launch<128>(p, s) : row16_launch<64>(p, s);
}

}  // namespace r3m
