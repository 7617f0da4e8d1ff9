// r3m_amd — convolution hot path for gfx950 (MI355X): NHWC fp32 implicit GEMM on the f32-input MFMA
// (v_mfma_f32_32x32x2_f32, exact fp32 == fmaf chain), LDS-staged operand tiles, XCD-aware block order.
//
// Replaces what the reference reaches through torchvision's ResNet -> ATen conv2d / cuDNN
// (call sites /root/reference/r3m/models/models_r3m.py:44-52,99): forward, dgrad and wgrad of every
// 1x1 / 3x3 / 7x7 convolution of ResNet-18/34/50, plus nn.Linear of the language
// reward head (/root/reference/r3m/models/models_language.py:43-51), which is the same GEMM with a bias epilogue.
//
//   * gather-GEMM : out[m, n] = sum_{tap, c} in[pix(m) + off(tap), c] * W[n, tap, c]
//       forward conv (taps = kh,kw; input stride = conv stride), dgrad (taps flipped; stride-2 dgrad is run as
//       4 output-parity classes so no MFMA work is spent on structural zeros), Linear (1 tap).
//       Epilogues: raw store (+ BatchNorm sum / sum-of-squares partials), accumulate, masked residual-gradient add,
//       bias (+ReLU), ReLU-mask.  Kernels (128x128 tiles, or 256x64 for 64-channel outputs):
//         gather_gemm_glds2_kernel : the production kernel. global -> LDS directly (global_load_lds_dwordx4), XOR-swizzled 128-byte
//                                    rows, low-VALU K loop (pointer arrays, two K tiles per iteration), DMA pieces issued between MFMAs
//         gather_gemm_glds_kernel  : generic direct-to-LDS variant (odd Cin/32; timing probes)
//         gather_gemm_kernel       : global -> VGPR -> LDS fallback, 36-float padded rows (R3M_GG_GLDS=0)
//   * wgrad_glds_kernel / wgrad_kernel : dW[co, tap, ci] = sum_m dY[m, co] * in[pix(m) + off(tap), ci]   (split-K over m, XCD-aware order)
//   * stem_prep / stem_fwd / stem_wgrad : conv1 7x7/2 straight from the frames (no im2col in HBM); legacy im2col route kept in the C ABI
//   (bf16 plans run conv_bf16.hip / stem_bf16.hip instead.)
//
// All staging is branch-free: taps that fall outside the image and rows past the end read a valid dummy address (a
// zero line / a clamped pixel) instead of being skipped, so the loader is straight-line code the compiler can interleave
// with the MFMA stream; the tap table is a dword array in the kernarg segment (scalar loads).
#include "common.h"
#include "conv_dev.h"
#include "augment_dev.h"
#include <cstdlib>
#include <cstring>
#include <utility>

namespace r3m {

__device__ __attribute__((aligned(128))) float g_zero_line[2048 + 64];


// =====================================================================================================
// gather-GEMM, direct-to-LDS staging (the 128x128 work-horse).
// Block 128 x 128, K step 32, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles of 32x32.
// LDS: two stages of {A[128][32], B[128][32]} floats, rows of exactly 128 B (what global_load_lds needs: the 64 lanes
// of one instruction land at base + lane*16, i.e. 8 consecutive rows), 16-byte slots XOR-swizzled by ((row>>1)&7): the
// swizzle is applied to the per-lane GLOBAL source address and again on the fragment read, never to the LDS
// destination. With it the ds_read_b128 fragment reads (16-lane groups, rows distinct mod 16) are conflict-free.
// Fragments: lane half h = lane>>5 reads k = 8g+4h..+3 of group g; MFMA step j contracts k = {8g+j, 8g+4+j} — A and B use
// the same permutation of k, the sum is unchanged.
// Pipeline: tile t+1's DMA is issued right after the barrier that publishes tile t and lands during tile t's 64 MFMAs
// per wave; one barrier per K step, no staging registers, no ds_write instructions.
// =====================================================================================================
template <int EPI>
__global__ __launch_bounds__(256) void gather_gemm_glds_kernel(const GatherGemmParams p) {
  constexpr int BM = 128, BN = 128, WM = 2, WN = 2, TM = 2, TN = 2;
  constexpr int STAGE = (BM + BN) * 32;                   // floats per stage (32 KiB)
  __shared__ __attribute__((aligned(128))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gridN = (p.Nc + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / gridN, nt = lid % gridN;  // column tiles of one row panel are neighbours -> same XCD L2
  const int m0 = mt * BM, n0 = nt * BN;

  // staging: wave w fills rows [32w, 32w+32) of A and of B, 8 rows (1 KiB) per instruction
  const int srow = lane >> 3;          // row within the 8-row group
  const int pslot = lane & 7;          // physical 16-byte slot written by this lane
  const int Hb = p.simple_rows ? 1 : p.Hi, Wb = p.simple_rows ? 1 : p.Wi;
  RowDesc ad[4];
  int acol[4];                         // logical k offset (floats) this lane fetches for A/B row group j
  unsigned arow_ok = 0;
  long long bbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 32 + j * 8 + srow;                 // row inside the tile
    acol[j] = (pslot ^ ((r >> 1) & 7)) * 4;
    const int m = m0 + r;
    ad[j] = decode_row(p, m);
    if (m < p.M) arow_ok |= 1u << j;
    const int n = min(n0 + r, p.Nc - 1);                    // columns past Nc are computed on a clamped row, never stored
    bbase[j] = (long long)n * p.T * p.Ci;
  }
  const float* zline = g_zero_line + pslot * 4;

  const int kpt = p.Ci >> 5;          // K tiles per tap
  const int nk = p.ntaps * kpt;

  auto issue_tile = [&](int pack, int chunk, int stage) {
    const int dy = (pack << 24) >> 24, dx = (pack << 16) >> 24, wt = pack >> 16;
    const int c0 = chunk * 32;
    const long long woff = (long long)wt * p.Ci + c0;
    float* la = smem + stage * STAGE + wave * 32 * 32;
    float* lb = la + BM * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = ad[j].iy + dy, ix = ad[j].ix + dx;
      const bool in = ((unsigned)iy < (unsigned)Hb) && ((unsigned)ix < (unsigned)Wb) && ((arow_ok >> j) & 1u);
      const int iyc = min(max(iy, 0), Hb - 1), ixc = min(max(ix, 0), Wb - 1);
      const float* src = p.A + ad[j].base + ((long long)iyc * p.Wi + ixc) * p.Ci + c0 + acol[j];
      // bitwise select keeps the loader straight-line (a ?: here is turned back into an exec-masked branch)
      const unsigned long long msk = in ? ~0ull : 0ull;
      src = reinterpret_cast<const float*>((reinterpret_cast<unsigned long long>(src) & msk) |
                                           (reinterpret_cast<unsigned long long>(zline) & ~msk));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(la + j * 8 * 32), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* src = p.B + bbase[j] + woff + acol[j];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lb + j * 8 * 32), 16, 0, 0);
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
  const int xr = (lrow >> 1) & 7;
  int goff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) goff[g] = ((2 * g + lh) ^ xr) * 4;
  const float* fragA0 = smem + (wm * 64 + lrow) * 32;
  const float* fragB0 = smem + BM * 32 + (wn * 64 + lrow) * 32;

  int tap_n = 0, chunk_n = 0;
  int pack_cur = nk > 0 ? p.tap[0] : 0;
  int pack_next = p.ntaps > 1 ? p.tap[1] : pack_cur;
  if (nk > 0) issue_tile(pack_cur, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = (R3M_PROBE(p) == 0) ? (kt & 1) : 0;       // timing probes read stage 0 only
    if (R3M_PROBE(p) != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my share of tile kt has landed
    if (R3M_PROBE(p) != 2) __syncthreads();                   // everyone's has; and everyone is done reading stage cur^1
    if (kt + 1 < nk && R3M_PROBE(p) != 1) {
      if (++chunk_n == kpt) {
        chunk_n = 0;
        ++tap_n;
        pack_cur = pack_next;
        pack_next = p.tap[min(tap_n + 1, p.ntaps - 1)];
      }
      issue_tile(pack_cur, chunk_n, (R3M_PROBE(p) == 0) ? (cur ^ 1) : 1);
    }
    const float* fa = fragA0 + cur * STAGE;
    const float* fb = fragB0 + cur * STAGE;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const f32x4*>(fa + t * 32 * 32 + goff[g]);
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const f32x4*>(fb + t * 32 * 32 + goff[g]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
    }
  }
  __syncthreads();   // all fragment reads done before the epilogue reuses the stages

  gg_epilogue<BM, BN, WM, WN, EPI, 2 * STAGE>(p, acc, smem, m0, n0, mt);
}

// =====================================================================================================
// gather-GEMM, direct-to-LDS staging, low-VALU main loop (used whenever Ci/32 is even — every ResNet layer).
// On gfx950 the f32-input MFMA runs at the fp32 VECTOR rate and, measured here, does NOT overlap with VALU work of
// co-resident waves: every VALU instruction in the K loop costs MFMA time (loads issued but never waited for cost the
// same 12 % as the full pipeline; removing the loader entirely gives 139-149 TF). So this variant strips the loop of
// vector ALU work: per-lane source POINTERS are kept per staged row and advanced by 256 B every second K step, the odd
// step uses the instruction's immediate offset (+128 B), validity is folded into the pointer once per tap (invalid rows
// walk along a zero buffer), the K loop is unrolled by two so LDS stage and fragment offsets are immediates, and the
// wave-uniform LDS destinations live in SGPRs. ~8 VALU instructions per K step instead of ~160.
// =====================================================================================================
template <int BM, int BN, int STG, int IMM>
__device__ __forceinline__ void glds_issue(const float* const (&pa)[BM / 32], const float* const (&pb)[BN / 32], float* smem,
                                           int wave_s) {
  constexpr int STAGE = (BM + BN) * 32;
  // the instruction's immediate offset is added to BOTH the global address and the LDS address (M0 base + offset +
  // lane*16), so the LDS destination is pre-biased by -IMM
  float* la = smem + STG * STAGE + wave_s * (BM / 4) * 32 - IMM / 4;
  float* lb = smem + STG * STAGE + BM * 32 + wave_s * (BN / 4) * 32 - IMM / 4;
#pragma unroll
  for (int j = 0; j < BM / 32; ++j)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[j],
                                     (__attribute__((address_space(3))) void*)(la + j * 8 * 32), 16, IMM, 0);
#pragma unroll
  for (int j = 0; j < BN / 32; ++j)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[j],
                                     (__attribute__((address_space(3))) void*)(lb + j * 8 * 32), 16, IMM, 0);
}

template <int BM, int BN, int STG>
__device__ __forceinline__ void glds_mfma(f32x16 (&acc)[2][2], const float* const (&fa)[4], const float* const (&fb)[4]) {
  constexpr int STAGE = (BM + BN) * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const f32x4*>(fa[g] + STG * STAGE + t * 32 * 32);
#pragma unroll
    for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const f32x4*>(fb[g] + STG * STAGE + t * 32 * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
  }
}

// MFMAs of the tile in stage STG_M with the NEXT tile's DMA pieces interleaved: one piece after every 64/NP MFMAs. A DMA
// instruction costs the issuing wave ~60-180 cycles of issue time (MI355X_MICROARCH.md); spread between MFMAs (each holds
// the matrix pipe 64 cycles) that cost hides behind the wave's own MFMAs instead of delaying their start.
template <int BM, int BN, int STG_M, int STG_D, int IMM>
__device__ __forceinline__ void glds_mfma_dma(f32x16 (&acc)[2][2], const float* const (&fa)[4], const float* const (&fb)[4],
                                              const float* const (&pa)[BM / 32], const float* const (&pb)[BN / 32], float* smem,
                                              int wave_s, bool do_dma) {
  constexpr int STAGE = (BM + BN) * 32;
  constexpr int AJ = BM / 32, BJ = BN / 32, NP = AJ + BJ;   // DMA pieces per wave per tile
  static_assert(NP == 8 || NP == 10, "piece schedule assumes 8 (128x128) or 10 (256x64) pieces");
  float* la = smem + STG_D * STAGE + wave_s * (BM / 4) * 32 - IMM / 4;
  float* lb = smem + STG_D * STAGE + BM * 32 + wave_s * (BN / 4) * 32 - IMM / 4;
  auto dma_one = [&](int pc) {
    if (pc < AJ)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[pc],
                                       (__attribute__((address_space(3))) void*)(la + pc * 8 * 32), 16, IMM, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[pc - AJ],
                                       (__attribute__((address_space(3))) void*)(lb + (pc - AJ) * 8 * 32), 16, IMM, 0);
  };
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const f32x4*>(fa[g] + STG_M * STAGE + t * 32 * 32);
#pragma unroll
    for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const f32x4*>(fb[g] + STG_M * STAGE + t * 32 * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
      if (do_dma) {
        // 16 slots (one per 4 MFMAs); use every second slot for 8 pieces, and slots 0..9 of the odd/even mix for 10
        const int slot = g * 4 + j;   // compile-time after unrolling: the piece index must be too (no scratch arrays)
        const bool fire = (NP == 8) ? ((slot & 1) == 1) : (slot < 12 && (slot % 6) != 5);
        const int piece = (NP == 8) ? (slot >> 1) : (slot - (slot > 5 ? 1 : 0));
        if (fire) {
          __builtin_amdgcn_sched_barrier(0);
          dma_one(piece);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}

// Block BM x BN with 4 waves laid out WM x WN, every wave a 64 x 64 sub-tile: <128,128,2,2> and <256,64,4,1>.
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void gather_gemm_glds2_kernel(const GatherGemmParams p) {
  constexpr int TM = 2, TN = 2;
  static_assert(BM / WM == 64 && BN / WN == 64 && WM * WN == 4, "wave tile is 64 x 64");
  constexpr int STAGE = (BM + BN) * 32;
  constexpr int AJ = BM / 32, BJ = BN / 32;     // DMA instructions per wave per stage (8 rows each)
  __shared__ __attribute__((aligned(128))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR: LDS destinations stay scalar
  const int wm = wave_s / WN, wn = wave_s % WN;
  const int gridN = (p.Nc + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / gridN, nt = lid % gridN;
  const int m0 = mt * BM, n0 = nt * BN;

  const int srow = lane >> 3, pslot = lane & 7;
  const int Hb = p.simple_rows ? 1 : p.Hi, Wb = p.simple_rows ? 1 : p.Wi;
  RowDesc ad[AJ];
  int acol[AJ];
  unsigned arow_ok = 0;
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int r = wave_s * (BM / 4) + j * 8 + srow;
    acol[j] = (pslot ^ ((r >> 1) & 7)) * 4;
    const int m = m0 + r;
    ad[j] = decode_row(p, m);
    if (m < p.M) arow_ok |= 1u << j;
  }
  const float* bptr[BJ];               // weight row pointers (tap 0, chunk 0)
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int r = wave_s * (BN / 4) + j * 8 + srow;
    const int n = min(n0 + r, p.Nc - 1);
    bptr[j] = p.B + (long long)n * p.T * p.Ci + (pslot ^ ((r >> 1) & 7)) * 4;
  }

  const int kpt = p.Ci >> 5;          // K tiles per tap (even)
  const int hpt = kpt >> 1;           // tile pairs per tap
  const int npairs = p.ntaps * hpt;

  const float* pa[AJ];
  const float* pb[BJ];
  auto set_tap = [&](int pack) {
    const int dy = (pack << 24) >> 24, dx = (pack << 16) >> 24, wt = pack >> 16;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int iy = ad[j].iy + dy, ix = ad[j].ix + dx;
      const bool in = ((unsigned)iy < (unsigned)Hb) && ((unsigned)ix < (unsigned)Wb) && ((arow_ok >> j) & 1u);
      const int iyc = min(max(iy, 0), Hb - 1), ixc = min(max(ix, 0), Wb - 1);
      const float* src = p.A + ad[j].base + ((long long)iyc * p.Wi + ixc) * p.Ci + acol[j];
      const unsigned long long msk = in ? ~0ull : 0ull;
      pa[j] = reinterpret_cast<const float*>((reinterpret_cast<unsigned long long>(src) & msk) |
                                             (reinterpret_cast<unsigned long long>(g_zero_line + acol[j]) & ~msk));
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) pb[j] = bptr[j] + (long long)wt * p.Ci;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lrow = lane & 31, lh = lane >> 5;
  const int xr = (lrow >> 1) & 7;
  const float* fa[4];
  const float* fb[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int go = ((2 * g + lh) ^ xr) * 4;
    fa[g] = smem + (wm * 64 + lrow) * 32 + go;
    fb[g] = smem + BM * 32 + (wn * 64 + lrow) * 32 + go;
  }

  int tap_n = 0, cp = 0;
  if (npairs > 0) {
    set_tap(p.tap[0]);
    glds_issue<BM, BN, 0, 0>(pa, pb, smem, wave_s);               // tile 0
  }
  int pack_next = p.ntaps > 1 ? p.tap[1] : 0;
  if (R3M_PROBE(p) == 8) {   // clustered DMA issue (kept for A/B: R3M_GG_DEBUG=8)
    for (int pr = 0; pr < npairs; ++pr) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      glds_issue<BM, BN, 1, 128>(pa, pb, smem, wave_s);
      glds_mfma<BM, BN, 0>(acc, fa, fb);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (pr + 1 < npairs) {
        if (++cp == hpt) {
          cp = 0;
          ++tap_n;
          set_tap(pack_next);
          pack_next = p.tap[min(tap_n + 1, p.ntaps - 1)];
        } else {
#pragma unroll
          for (int j = 0; j < AJ; ++j) pa[j] += 64;
#pragma unroll
          for (int j = 0; j < BJ; ++j) pb[j] += 64;
        }
        glds_issue<BM, BN, 0, 0>(pa, pb, smem, wave_s);
      }
      glds_mfma<BM, BN, 1>(acc, fa, fb);
    }
  } else {
    for (int pr = 0; pr < npairs; ++pr) {
      // ---- even tile (stage 0): its MFMAs carry the DMA of the odd tile (same tap, next 32 channels, stage 1) ----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      glds_mfma_dma<BM, BN, 0, 1, 128>(acc, fa, fb, pa, pb, smem, wave_s, true);
      // ---- odd tile (stage 1): carries the DMA of the next pair's even tile (stage 0) ----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const bool more = pr + 1 < npairs;
      if (more) {
        if (++cp == hpt) {
          cp = 0;
          ++tap_n;
          set_tap(pack_next);
          pack_next = p.tap[min(tap_n + 1, p.ntaps - 1)];
        } else {
#pragma unroll
          for (int j = 0; j < AJ; ++j) pa[j] += 64;
#pragma unroll
          for (int j = 0; j < BJ; ++j) pb[j] += 64;
        }
      }
      glds_mfma_dma<BM, BN, 1, 0, 0>(acc, fa, fb, pa, pb, smem, wave_s, more);
    }
  }
  __syncthreads();

  gg_epilogue<BM, BN, WM, WN, EPI, 2 * STAGE>(p, acc, smem, m0, n0, mt);
}

// =====================================================================================================
// gather-GEMM 128 x 128 with 16-wide K tiles (64-byte LDS rows) for launches whose main loop is SHORT: the expanding 1x1
// convolutions (K = Cin = 64 / 128 / 256 against N = 4 Cin) and the 1x1 downsample convolutions. There a block's life is
// prologue latency + 2-8 K steps + a 64 KB epilogue, and with the 64 KB of the 32-wide two-stage ring only two blocks share a
// CU, so nothing hides the one's loads / stores behind the other's (measured round 1: 62.6 / 87 / 110 TFLOP/s for K = 64 / 128 /
// 256 against 117 for the class; these launches sit at the HBM ridge — 25 FLOP/B for K = 64). With 16-wide tiles the ring is
// 2 x 16 KB and the LDS footprint is the 34 KB epilogue slab: three to four blocks per CU (registers: 128 per lane).
// Layout: rows of 16 floats = four 16-byte slots, slot XOR ((row >> 2) & 3) — a ds_read_b128 lane group (16 lanes, MI355X
// guide) then touches 16 distinct slots of the 256-byte bank row; one DMA instruction lands 16 rows (lane -> row lane>>2,
// slot lane&3; the swizzle is applied to the global source column). Fragments and MFMA order as in the 32-wide kernel
// (k = 8g + 4h .. +3 per lane half h, groups g = 0, 1), so results are bit-identical to it.
// =====================================================================================================
template <int EPI>
__global__ __launch_bounds__(256, 4) void gather_gemm_k16_kernel(const GatherGemmParams p) {
  constexpr int BM = 128, BN = 128, WM = 2, WN = 2;
  constexpr int STAGE = (BM + BN) * 16;                  // floats per stage (16 KiB)
  constexpr int SMEM = 4 * 32 * (64 + 4);                // the epilogue slab (8704 floats) >= the two stages (8192)
  static_assert(SMEM >= 2 * STAGE, "ring must fit under the epilogue slab");
  __shared__ __attribute__((aligned(128))) float smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_s / WN, wn = wave_s % WN;
  const int gridN = (p.Nc + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / gridN, nt = lid % gridN;
  const int m0 = mt * BM, n0 = nt * BN;

  // staging: a wave owns 32 A rows and 32 B rows = 2 + 2 DMA instructions of 16 rows per tile
  const int srow = lane >> 2;
  const int scol = ((lane & 3) ^ ((lane >> 4) & 3)) * 4;   // (row >> 2) & 3 == (lane >> 4) & 3: row = 32 w + 16 j + (lane >> 2)
  const int Hb = p.simple_rows ? 1 : p.Hi, Wb = p.simple_rows ? 1 : p.Wi;
  RowDesc ad[2];
  unsigned arow_ok = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wave_s * 32 + j * 16 + srow;
    ad[j] = decode_row(p, m);
    if (m < p.M) arow_ok |= 1u << j;
  }
  const float* bptr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = min(n0 + wave_s * 32 + j * 16 + srow, p.Nc - 1);
    bptr[j] = p.B + (long long)n * p.T * p.Ci + scol;
  }
  const int kpt = p.Ci >> 4;          // 16-wide tiles per tap (even: Ci is a multiple of 32)
  const int hpt = kpt >> 1;
  const int npairs = p.ntaps * hpt;

  const float* pa[2];
  const float* pb[2];
  auto set_tap = [&](int pack) {
    const int dy = (pack << 24) >> 24, dx = (pack << 16) >> 24, wt = pack >> 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int iy = ad[j].iy + dy, ix = ad[j].ix + dx;
      const bool in = ((unsigned)iy < (unsigned)Hb) && ((unsigned)ix < (unsigned)Wb) && ((arow_ok >> j) & 1u);
      const int iyc = min(max(iy, 0), Hb - 1), ixc = min(max(ix, 0), Wb - 1);
      const float* src = p.A + ad[j].base + ((long long)iyc * p.Wi + ixc) * p.Ci + scol;
      const unsigned long long msk = in ? ~0ull : 0ull;
      pa[j] = reinterpret_cast<const float*>((reinterpret_cast<unsigned long long>(src) & msk) |
                                             (reinterpret_cast<unsigned long long>(g_zero_line + scol) & ~msk));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) pb[j] = bptr[j] + (long long)wt * p.Ci;
  };
  // stage STG, the instruction's immediate IMM (bytes) is added to BOTH addresses -> LDS destination pre-biased by -IMM
  auto issue = [&](auto stg_c, auto imm_c) __attribute__((always_inline)) {
    constexpr int STG = decltype(stg_c)::value, IMM = decltype(imm_c)::value;
    float* la = smem + STG * STAGE + wave_s * 32 * 16 - IMM / 4;
    float* lb = smem + STG * STAGE + BM * 16 + wave_s * 32 * 16 - IMM / 4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[j],
                                       (__attribute__((address_space(3))) void*)(la + j * 16 * 16), 16, IMM, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[j],
                                       (__attribute__((address_space(3))) void*)(lb + j * 16 * 16), 16, IMM, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lrow = lane & 31, lh = lane >> 5;
  const int xr = (lrow >> 2) & 3;
  const float* fa[2];
  const float* fb[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int go = ((2 * g + lh) ^ xr) * 4;
    fa[g] = smem + (wm * 64 + lrow) * 16 + go;
    fb[g] = smem + BM * 16 + (wn * 64 + lrow) * 16 + go;
  }
  auto mfma_tile = [&](auto stg_c) __attribute__((always_inline)) {
    constexpr int STG = decltype(stg_c)::value;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const f32x4*>(fa[g] + STG * STAGE + t * 32 * 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const f32x4*>(fb[g] + STG * STAGE + t * 32 * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I64 = std::integral_constant<int, 64>;

  int tap_n = 0, cp = 0;
  if (npairs > 0) {
    set_tap(p.tap[0]);
    issue(I0{}, I0{});                                    // tile 0 -> stage 0
  }
  int pack_next = p.ntaps > 1 ? p.tap[1] : 0;
  for (int pr = 0; pr < npairs; ++pr) {
    // even tile (stage 0): issue the odd tile (same tap, next 16 channels: +64 B immediate) first, then the MFMAs
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(I1{}, I64{});
    mfma_tile(I0{});
    // odd tile (stage 1): issue the next pair's even tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (pr + 1 < npairs) {
      if (++cp == hpt) {
        cp = 0;
        ++tap_n;
        set_tap(pack_next);
        pack_next = p.tap[min(tap_n + 1, p.ntaps - 1)];
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) { pa[j] += 32; pb[j] += 32; }
      }
      issue(I0{}, I0{});
    }
    mfma_tile(I1{});
  }
  __syncthreads();
  gg_epilogue<BM, BN, WM, WN, EPI, SMEM>(p, acc, smem, m0, n0, mt);
}

// =====================================================================================================
// gather-GEMM, register staging (256x64 tiles for 64-channel layers; also the 128x128 fallback R3M_GG_GLDS=0).
// Operand tiles live in LDS as [row][k] with a 36-float row stride (conflict-free b128 writes and fragment reads).
// Single LDS stage; the next tile's global loads fly during the MFMA phase.
// =====================================================================================================
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gather_gemm_kernel(const GatherGemmParams p) {
  constexpr int NT = WM * WN * 64;      // threads
  constexpr int RPP = NT / 8;           // staging rows per pass (8 lanes x float4 cover one 32-float row)
  constexpr int S = 36;
  constexpr int STAGE = (BM + BN) * S;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int AJ = BM / RPP;  // float4 staging loads per thread (A)
  constexpr int BJ = BN / RPP;  // float4 staging loads per thread (B)
  __shared__ __attribute__((aligned(16))) float smem[STAGE];
  float* sA = smem;
  float* sB = smem + BM * S;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gridN = (p.Nc + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / gridN, nt = lid % gridN;
  const int m0 = mt * BM, n0 = nt * BN;

  const int c4 = tid & 7;   // which float4 of the 32-float k slice
  const int r0 = tid >> 3;  // staging row (0..RPP-1), + RPP*j

  const int Hb = p.simple_rows ? 1 : p.Hi, Wb = p.simple_rows ? 1 : p.Wi;
  RowDesc ad[AJ];
  unsigned arow_ok = 0;
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + r0 + RPP * j;
    ad[j] = decode_row(p, m);
    if (m < p.M) arow_ok |= 1u << j;
  }
  long long bbase[BJ];
  unsigned brow_ok = 0;
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    int n = n0 + r0 + RPP * j;
    if (n < p.Nc) brow_ok |= 1u << j; else n = p.Nc - 1;
    bbase[j] = (long long)n * p.T * p.Ci;
  }

  const int kpt = p.Ci >> 5;          // K tiles per tap
  const int nk = p.ntaps * kpt;

  f32x4 ra[AJ], rb[BJ];
  unsigned a_ok = 0;                  // validity bits of the tile currently held in ra[]
  auto load_tile = [&](int pack, int chunk) {
    const int dy = (pack << 24) >> 24, dx = (pack << 16) >> 24, wt = pack >> 16;
    const int c0 = chunk * 32 + c4 * 4;
    const long long woff = (long long)wt * p.Ci + c0;
    unsigned ok = 0;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int iy = ad[j].iy + dy, ix = ad[j].ix + dx;
      const bool in = ((unsigned)iy < (unsigned)Hb) && ((unsigned)ix < (unsigned)Wb);
      ok |= (in ? 1u : 0u) << j;
      const int iyc = min(max(iy, 0), Hb - 1), ixc = min(max(ix, 0), Wb - 1);
      ra[j] = ldg4(p.A + ad[j].base + ((long long)iyc * p.Wi + ixc) * p.Ci + c0);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) rb[j] = ldg4(p.B + bbase[j] + woff);
    a_ok = ok & arow_ok;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lrow = lane & 31;
  const int lh4 = (lane >> 5) * 4;
  const float* fragA = sA + (wm * TM * 32 + lrow) * S + lh4;
  const float* fragB = sB + (wn * TN * 32 + lrow) * S + lh4;

  int tap_n = 0, chunk_n = 0;
  int pack_cur = nk > 0 ? p.tap[0] : 0;
  int pack_next = p.ntaps > 1 ? p.tap[1] : pack_cur;
  if (nk > 0) load_tile(pack_cur, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<f32x4*>(sA + (r0 + RPP * j) * S + c4 * 4) = ((a_ok >> j) & 1u) ? ra[j] : zero4;
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<f32x4*>(sB + (r0 + RPP * j) * S + c4 * 4) = ((brow_ok >> j) & 1u) ? rb[j] : zero4;
    __syncthreads();
    if (kt + 1 < nk) {
      if (++chunk_n == kpt) {
        chunk_n = 0;
        ++tap_n;
        pack_cur = pack_next;
        pack_next = p.tap[min(tap_n + 1, p.ntaps - 1)];
      }
      load_tile(pack_cur, chunk_n);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const f32x4*>(fragA + t * 32 * S + g * 8);
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const f32x4*>(fragB + t * 32 * S + g * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
    }
    __syncthreads();
  }

  gg_epilogue<BM, BN, WM, WN, EPI, STAGE>(p, acc, smem, m0, n0, mt);
}

// =====================================================================================================
// 3x3 / stride 1 / pad 1 convolutions (forward and dgrad) on 128-wide outputs with an INPUT WINDOW in LDS (round 3).
// The gather kernel stages the 128-row A tile once per tap: 4 of its 8 DMA instructions per 64 MFMAs, and on this chip DMA issue
// is what keeps the fp32 main loop at 0.83 instead of the 0.9+ of a no-load loop (the weight gradient gained 6 % from 8 -> 5.3
// DMA instructions per 64 MFMAs). Here the nine taps of a tile of 128 consecutive output pixels read ONE window of 128 + 2W + 2
// input pixels per 32-channel chunk — staged once per chunk, double-buffered, one DMA instruction per tap step — plus the nine
// weight tiles: 4.7 DMA instructions per 64 MFMAs.
//   LDS (80 KB, two blocks per CU): 2 window buffers of 192 rows x 128 B (same 16-byte-slot XOR swizzle as the gather kernel),
//   2 weight stages of 128 x 128 B. Window row 191 is never inside the window (W <= 28): its DMA lanes are out of range, it holds
//   zeros, and a lane whose pixel has no (y + dy, x + dx) inside the image reads IT — the gather kernel's border arithmetic.
//   The A-fragment addresses of all 9 taps x 2 row tiles x 4 K groups are per-lane constants (72 registers), so the K loop has no
//   vector work: DMA offsets are constants, the (tap, chunk) position rides in the instructions' scalar offset.
// Summation order: (chunk, tap) instead of the gather kernel's (tap, chunk) — an fp32 reassociation, inside every gate.
// =====================================================================================================
// In use: <128,128,2,2,192> (W <= 28: the 128/256/512-channel layers; 80 KB, two blocks of 4 waves per CU): 126 -> 131, 133 -> 140,
// 133 -> 141 TFLOP/s at 14^2 / 28^2 / 7^2 (profiles/r03_win_ab.txt). The template also instantiates as <256,64,4,2,376> (the
// 64-channel layers at 56 x 56: 112 KB, ONE block of 8 waves per CU); measured 122 against the gather kernel's 124 there
// (a 370-row window per 2 chunks of 9 taps amortises less, and one block per CU exposes the chunk boundary) — not dispatched.
template <int BM, int BN, int WIN_ROWS>
struct WinCfg {
  static constexpr int WIN_BYTES = WIN_ROWS * 128;
  static constexpr int LDS = 2 * WIN_BYTES + 2 * BN * 128;
};

template <int BM, int BN, int WM, int WN, int WIN_ROWS, int EPI>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 4 ? 2 : 1) void conv3x3_win_kernel(const GatherGemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM == 2 && (TN == 1 || TN == 2), "wave tile 64 x 32 or 64 x 64");
  constexpr int WIN_BYTES = WIN_ROWS * 128;
  constexpr int BSTAGE = BN * 128;                               // bytes of one weight stage
  constexpr int LDSB = 2 * WIN_BYTES + 2 * BSTAGE;
  extern __shared__ __attribute__((aligned(128))) unsigned char wsm[];
  unsigned char* bst = wsm + 2 * WIN_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_s / WN, wn = wave_s % WN;
  const int gridN = (p.Nc + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / gridN, nt = lid % gridN;
  const int m0 = mt * BM, n0 = nt * BN;
  const int W = p.Wi, H = p.Hi, HW = H * W;
  const int HR = BM + 2 * W + 2;                                 // window rows in use (< WIN_ROWS)
  constexpr int ZR = WIN_ROWS - 1;                               // the zero row

  // ---- A fragment byte offsets inside a window buffer: [tap][row tile][K group] ----
  const int lrow = lane & 31, lh = lane >> 5;
  unsigned fa[9][TM][4];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int r = wm * (BM / WM) + t * 32 + lrow;
    const int m = m0 + r;
    int y = 0, x = 0;
    if (m < p.M) {
      const int rem = m % HW;
      y = rem / W;
      x = rem - y * W;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int pk = p.tap[k];
      const int dy = (pk << 24) >> 24, dx = (pk << 16) >> 24;
      const bool ok = (m < p.M) && ((unsigned)(y + dy) < (unsigned)H) && ((unsigned)(x + dx) < (unsigned)W);
      const unsigned wr = ok ? (unsigned)(r + (W + 1) + dy * W + dx) : (unsigned)ZR;
      const unsigned base = (wr << 7) | (((wr >> 1) & 7u) << 4);
#pragma unroll
      for (int g = 0; g < 4; ++g) fa[k][t][g] = base ^ ((unsigned)(2 * g + lh) << 4);
    }
  }
  // ---- B fragment byte offsets inside a weight stage ----
  unsigned fb[4];
  {
    const int xr = (lrow >> 1) & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) fb[g] = (unsigned)((wn * TN * 32 + lrow) * 128 + (((2 * g + lh) ^ xr) << 4));
  }

  // ---- DMA: window pieces (8 rows each; this wave owns pieces wave_s + NW q) and weight pieces (8 rows each) ----
  const int srow = lane >> 3, pslot = lane & 7;
  const long long px00 = (long long)m0 - (W + 1);                // pixel of window row 0
  const long long pxb = px00 > 0 ? px00 : 0;                     // descriptor base pixel
  const float* a_base = p.A + pxb * p.Ci;
  int a_bytes;
  {
    const long long rest = ((long long)p.M - pxb) * p.Ci * 4;
    a_bytes = rest <= 0 ? 0 : (rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB);
  }
  constexpr int NWP_ALL = WIN_ROWS / 8;                          // window pieces of a full buffer
  constexpr int NWQ = (NWP_ALL + NW - 1) / NW;                   // per wave (6)
  static_assert(NWQ <= 9, "one window piece per tap step");
  unsigned woff[NWQ];
#pragma unroll
  for (int q = 0; q < NWQ; ++q) {
    const int hr = 8 * (wave_s + NW * q) + srow;
    const long long px = px00 + hr;
    const bool ok = hr < HR && px >= 0;                          // px >= M falls off the descriptor
    woff[q] = ok ? (unsigned)((int)(px - pxb) * p.Ci * 4) + (unsigned)((pslot ^ ((hr >> 1) & 7)) << 4) : BUF_OOB;
  }
  const int nwp = (HR + 7) / 8;                                  // pieces that hold window rows; the last piece (zero row) always goes
  constexpr int BJ = BN / 8 / NW;                                // weight pieces per wave (4 or 1)
  static_assert(BJ >= 1 && BJ * 8 * NW == BN, "weight rows split over the waves");
  unsigned bvoff[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int r = wave_s * (BN / NW) + j * 8 + srow;
    const int n = min(n0 + r, p.Nc - 1);
    bvoff[j] = (unsigned)(n * p.T * p.Ci * 4) + (unsigned)((pslot ^ ((r >> 1) & 7)) << 4);
  }
  const int b_bytes = p.Nc * p.T * p.Ci * 4;

  auto issue_window_piece = [&](int q, int chunk, int buf) __attribute__((always_inline)) {
    const int i = wave_s + NW * q;
    if (R3M_PROBE(p) & 8) return;                                // timing probe: no window DMA (stale LDS)
    if (i < nwp || i == NWP_ALL - 1)
      buf_dma16(a_base, a_bytes, wsm + buf * WIN_BYTES + i * 1024, woff[q], chunk * 128);
  };
  auto issue_b_piece = [&](int j, int tapk, int chunk, int stage) __attribute__((always_inline)) {
    const int wt = p.tap[tapk] >> 16;
    if (R3M_PROBE(p) & 16) return;                               // timing probe: no weight DMA
    buf_dma16(p.B, b_bytes, bst + stage * BSTAGE + (wave_s * (BN / NW) + j * 8) * 128, bvoff[j], (wt * p.Ci + chunk * 32) * 4);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nchunks = p.Ci >> 5;
  // one tap step: the MFMAs of window buffer WB (tap K) x weight stage ST, carrying the DMA of the NEXT step's weights and one
  // window piece of the next chunk
  auto step = [&](auto k_c, auto wb_c, auto st_c, int chunk) __attribute__((always_inline)) {
    constexpr int K = decltype(k_c)::value, WB = decltype(wb_c)::value, ST = decltype(st_c)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool last_chunk = chunk + 1 >= nchunks;
    const bool more_b = !(last_chunk && K == 8);
    const int nk = K == 8 ? 0 : K + 1, nc = K == 8 ? chunk + 1 : chunk;
    static_for<4>([&](auto g_c) __attribute__((always_inline)) {
      constexpr int g = decltype(g_c)::value;
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t)
        a[t] = *reinterpret_cast<const f32x4*>(wsm + WB * WIN_BYTES + fa[K][t][g]);
#pragma unroll
      for (int t = 0; t < TN; ++t)
        b[t] = *reinterpret_cast<const f32x4*>(bst + ST * BSTAGE + t * 32 * 128 + fb[g]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
        if (j == 1) {
          __builtin_amdgcn_sched_barrier(0);
          if (g < BJ && more_b) issue_b_piece(g, nk, nc, ST ^ 1);
          if (g == 3 && K < NWQ && !last_chunk) issue_window_piece(K, chunk + 1, WB ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
  };
  auto chunk_steps = [&](auto wb_c, int chunk) __attribute__((always_inline)) {
    constexpr int WB = decltype(wb_c)::value;                  // window buffer = chunk parity; weight stage = (chunk + tap) parity
    static_for<9>([&](auto k_c) __attribute__((always_inline)) {
      constexpr int K = decltype(k_c)::value;
      step(k_c, wb_c, std::integral_constant<int, (WB + K) & 1>{}, chunk);
    });
  };

  if (nchunks > 0) {
#pragma unroll
    for (int q = 0; q < NWQ; ++q) issue_window_piece(q, 0, 0);
#pragma unroll
    for (int j = 0; j < BJ; ++j) issue_b_piece(j, 0, 0, 0);
  }
  int c = 0;
  for (; c + 1 < nchunks; c += 2) {
    chunk_steps(std::integral_constant<int, 0>{}, c);
    chunk_steps(std::integral_constant<int, 1>{}, c + 1);
  }
  if (c < nchunks) chunk_steps(std::integral_constant<int, 0>{}, c);
  __syncthreads();

  gg_epilogue<BM, BN, WM, WN, EPI, LDSB / 4>(p, acc, reinterpret_cast<float*>(wsm), m0, n0, mt);
}

// 3x3 / stride 1 / pad 1 on the full pixel grid with the taps being exactly {-1,0,1}^2, 128-wide outputs, W <= 28
static bool conv3x3_win_eligible(const GatherGemmParams& p, int maxW) {
  if (p.ntaps != 9 || p.is != 1 || p.os != 1 || p.simple_rows || p.Hg != p.Hi || p.Wg != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi) return false;
  if (p.Wi > maxW || p.Wi < 1 || (p.Ci & 31)) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    if (p.dy[t] < -1 || p.dy[t] > 1 || p.dx[t] < -1 || p.dx[t] > 1) return false;
    seen |= 1u << ((p.dy[t] + 1) * 3 + (p.dx[t] + 1));
  }
  return seen == 0x1ffu && (long long)p.Nc * p.T * p.Ci * 4 < 0x7FFFF000LL;
}

static inline bool gg_wide(int Nc) { return (Nc % 128) == 0; }

int gather_gemm_grid_m(int M, int Nc) { return gg_wide(Nc) ? ceil_div(M, 128) : ceil_div(M, 256); }

// R3M_GG_GLDS: 2 (default) low-VALU direct-to-LDS kernel, 1 generic direct-to-LDS kernel, 0 register staging
static int gg_use_glds() {
  const int v = R3M_ENV_INT("R3M_GG_GLDS", 2);
  return v;
}

#define GG_EPI_SWITCH(LAUNCH)                                                       \
  switch (p.flags) {                                                                \
    case 0: LAUNCH(0); break;                                                       \
    case EPI_STATS: LAUNCH(EPI_STATS); break;                                       \
    case EPI_ACCUM: LAUNCH(EPI_ACCUM); break;                                       \
    case EPI_MASKED_ADD: LAUNCH(EPI_MASKED_ADD); break;                             \
    case EPI_BIAS: LAUNCH(EPI_BIAS); break;                                         \
    case EPI_RELU: LAUNCH(EPI_RELU); break;                                         \
    case EPI_BIAS | EPI_RELU: LAUNCH(EPI_BIAS | EPI_RELU); break;                   \
    case EPI_MASK_OUT: LAUNCH(EPI_MASK_OUT); break;                                 \
    case EPI_BNRED: LAUNCH(EPI_BNRED); break;                                       \
    case EPI_BNRED | EPI_MASKED_ADD: LAUNCH(EPI_BNRED | EPI_MASKED_ADD); break;     \
    default:                                                                        \
      set_last_error("gather_gemm: unsupported epilogue flag combination %d", p.flags); \
      return 1;                                                                     \
  }

// algorithmic HBM bytes of one gather-GEMM launch: the input tensor once, the weights once, the result once (+ the tensors
// a read-modify-write epilogue adds: the old result / the residual gradient and its 1-bit mask)
double gather_gemm_alg_bytes(const GatherGemmParams& p, int elem) {
  const double in = p.simple_rows ? (double)p.M * p.Ci : (double)p.N * p.Hi * p.Wi * p.Ci;
  const double out = (double)p.M * p.Nc;
  double b = elem * (in + out + (double)p.Nc * p.ntaps * p.Ci);
  if (p.flags & EPI_ACCUM) b += elem * out;
  if (p.flags & EPI_MASKED_ADD) b += elem * out + out / 8.0;
  if (p.flags & EPI_MASK_OUT) b += elem * out;
  if (p.flags & EPI_BNRED) b += elem * out + (p.bn_bits ? out / 8.0 : 0.0) + 8.0 * bnred_partial_rows(p.M) * p.Nc;   // + y, mask, partials
  return b;
}

static thread_local unsigned* t_tile_ctr = nullptr;
static thread_local int t_tile_ctr_sets = 0;   // further sets of 8 counters behind t_tile_ctr (the parity-class launches of a stride-2 dgrad)
static int g_dynamic_tiles = 1;          // diagnostic (tools/occupy_ab.py): plain int, written before launches from the same thread
void gg_set_tile_counters(unsigned* ctr8, int sets) {
  t_tile_ctr = g_dynamic_tiles ? ctr8 : nullptr;
  t_tile_ctr_sets = t_tile_ctr ? sets : 0;
}
int gg_set_dynamic_tiles(int on) { const int old = g_dynamic_tiles; g_dynamic_tiles = on ? 1 : 0; return old; }

// Which kernel family a launch runs (a pure function of the launch parameters; taps must be packed). Also what
// r3m_debug_conv_route reports, so that the dispatch DESIGN.md describes is checked on CPU (tests/test_host.py).
enum : int { GG_ROUTE_WIN = 1, GG_ROUTE_PW = 10 /* + pw_gemm_form: 11 pointwise, 12 gather, 13 strided output */, GG_ROUTE_K16 = 20,
             GG_ROUTE_GLDS2 = 21, GG_ROUTE_OTHER = 22, GG_ROUTE_BF16 = 30 };
static int gg_route(const GatherGemmParams& p) {
  if (p.dtype == DT_BF16) return GG_ROUTE_BF16;
  const bool glds2 = gg_use_glds() == 2 && ((p.Ci >> 5) & 1) == 0 && p.Ci <= 2048;
  if (gg_wide(p.Nc)) {
    if (R3M_ENV_INT("R3M_GG_WIN", 1) && conv3x3_win_eligible(p, 28)) return GG_ROUTE_WIN;   // probe builds: 0 = gather kernel for 3x3 / stride 1 too
    if (R3M_ENV_INT("R3M_GG_PW", 1) && pw_gemm_eligible(p)) return GG_ROUTE_PW + pw_gemm_form(p);
    // short main loop + wide output: single tap, K <= 256, N >= 2 K (expanding / downsample 1x1 convolutions) -> 16-wide K tiles
    // R3M_GG_K16 (probe builds): 0 = never, 1 = the rule above, 2 = every wide launch (experiment: 4 blocks per CU everywhere)
    const int k16_mode = R3M_ENV_INT("R3M_GG_K16", 1);
    if ((p.Ci & 31) == 0 && (k16_mode == 2 || (k16_mode == 1 && p.ntaps == 1 && p.Ci <= 256 && p.Nc >= 2 * p.Ci))) return GG_ROUTE_K16;
    return glds2 ? GG_ROUTE_GLDS2 : GG_ROUTE_OTHER;
  }
  if (R3M_ENV_INT("R3M_GG_PW", 1) && pw_gemm_eligible(p)) return GG_ROUTE_PW + pw_gemm_form(p);
  return glds2 ? GG_ROUTE_GLDS2 : GG_ROUTE_OTHER;
}
// inference forward: can this launch apply eval-mode BatchNorm (+ residual) (+ ReLU) where it stores (EPI_AFFINE family)? True for the
// kernels every ResNet layer runs (the persistent kernel, the 3x3 window kernel; every bf16 kernel); the engine falls back to
// conv + bn_act_fwd for anything else (odd shapes of the fuzz tests).
bool gather_gemm_fuses_affine(const GatherGemmParams& p_in) {
  GatherGemmParams p = p_in;
  for (int t = 0; t < p.ntaps; ++t)
    p.tap[t] = (int)((unsigned)(unsigned char)p.dy[t] | ((unsigned)(unsigned char)p.dx[t] << 8) | ((unsigned)p.wt[t] << 16));
  if (p.dtype == DT_BF16) return (p.Nc & 7) == 0 && (p.Ci & 63) == 0;
  const int r = gg_route(p);
  if (r == GG_ROUTE_WIN) return p.flags == (EPI_AFFINE | EPI_RELU) || p.flags == (EPI_AFFINE | EPI_ACCUM | EPI_RELU);
  return r > GG_ROUTE_PW && r < GG_ROUTE_K16;
}
static thread_local int* t_route_out = nullptr;      // dry run (r3m_debug_conv_route): record the route of every launch, launch nothing
static thread_local int t_route_n = 0, t_route_cap = 0;
void gg_route_record_begin(int* out, int cap) { t_route_out = out; t_route_n = 0; t_route_cap = cap; }
int gg_route_record_end() { const int n = t_route_n; t_route_out = nullptr; t_route_n = t_route_cap = 0; return n; }

int launch_gather_gemm(const GatherGemmParams& p_in, hipStream_t s) {
  GatherGemmParams p = p_in;
  p.tile_ctr = t_tile_ctr;     // one set of counters serves ONE launch: the parity-class launches of a stride-2 dgrad take the next
  if (t_tile_ctr && --t_tile_ctr_sets > 0) t_tile_ctr += 8;   // set each, and the next layer must not reuse any of them
  else { t_tile_ctr = nullptr; t_tile_ctr_sets = 0; }
  if (p.dtype == DT_BF16) {
    {
      const int dbg = R3M_ENV_INT("R3M_GG_DEBUG", 0);
      p.debug = dbg;   // timing probes only (wrong results when != 0)
    }
    R3M_REQUIRE(p.ntaps >= 0 && p.ntaps <= MAX_TAPS, "gather_gemm: ntaps=%d", p.ntaps);
    R3M_REQUIRE(p.M > 0 && p.Nc > 0, "gather_gemm: empty problem M=%d Nc=%d", p.M, p.Nc);
    for (int t = 0; t < p.ntaps; ++t)
      p.tap[t] = (int)((unsigned)(unsigned char)p.dy[t] | ((unsigned)(unsigned char)p.dx[t] << 8) | ((unsigned)p.wt[t] << 16));
    if (t_route_out) {
      if (t_route_n < t_route_cap) t_route_out[t_route_n] = gg16_route(p);    // 30 gather, 31 halo, 32 kernel-row (GG_ROUTE_BF16 + family)
      ++t_route_n;
      return 0;
    }
    return launch_gather_gemm_bf16(p, s);
  }
  R3M_REQUIRE(p.Ci % 32 == 0, "gather_gemm: Ci=%d must be a multiple of 32", p.Ci);
  R3M_REQUIRE(p.Nc % 4 == 0, "gather_gemm: Nc=%d must be a multiple of 4", p.Nc);
  R3M_REQUIRE(p.ntaps >= 0 && p.ntaps <= MAX_TAPS, "gather_gemm: ntaps=%d", p.ntaps);
  R3M_REQUIRE(p.M > 0 && p.Nc > 0, "gather_gemm: empty problem M=%d Nc=%d", p.M, p.Nc);
  R3M_REQUIRE((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(p.out) & 15) == 0,
              "gather_gemm: operands must be 16-byte aligned");
  {
    const int dbg = R3M_ENV_INT("R3M_GG_DEBUG", 0);
    p.debug = dbg;   // timing probes only (wrong results when != 0)
  }
  for (int t = 0; t < p.ntaps; ++t)
    p.tap[t] = (int)((unsigned)(unsigned char)p.dy[t] | ((unsigned)(unsigned char)p.dx[t] << 8) | ((unsigned)p.wt[t] << 16));
  const double kdim = (double)p.ntaps * p.Ci;
  const double flops = 2.0 * (double)p.M * (double)p.Nc * kdim;
  const int route = gg_route(p);
  if (t_route_out) {
    if (t_route_n < t_route_cap) t_route_out[t_route_n] = route;
    ++t_route_n;
    return 0;
  }
  if (gg_wide(p.Nc)) {
    const int grid = ceil_div(p.M, 128) * ceil_div(p.Nc, 128);
    prof_begin(KC_GEMM_WIDE, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    {
      if (route == GG_ROUTE_WIN) {
        typedef WinCfg<128, 128, 192> Cfg;
#define LAUNCH_WIN(E)                                                                                                        \
  do {                                                                                                                       \
    static DynLdsOptIn oi;                                                                                                   \
    if (int e = ensure_dyn_lds(oi, reinterpret_cast<const void*>(conv3x3_win_kernel<128, 128, 2, 2, 192, E>), Cfg::LDS, "conv3x3_win")) return e; \
    hipLaunchKernelGGL((conv3x3_win_kernel<128, 128, 2, 2, 192, E>), dim3(grid), dim3(256), Cfg::LDS, s, p);                 \
  } while (0)
        switch (p.flags) {                       // inference forward (round 6): eval-mode BatchNorm (+ residual in `out`) + ReLU at the store
          case EPI_AFFINE | EPI_RELU: LAUNCH_WIN(EPI_AFFINE | EPI_RELU); break;
          case EPI_AFFINE | EPI_ACCUM | EPI_RELU: LAUNCH_WIN(EPI_AFFINE | EPI_ACCUM | EPI_RELU); break;
          default:
            GG_EPI_SWITCH(LAUNCH_WIN)
        }
#undef LAUNCH_WIN
        prof_bytes(gather_gemm_alg_bytes(p, 4));
        prof_end(s);
        return check_launch("conv3x3_win");
      }
    }
    if (route > GG_ROUTE_PW && route < GG_ROUTE_K16) {   // dense or parity-strided output rows: persistent kernel (conv_pw.hip)
      if (int e = launch_pw_gemm(p, s)) return e;
      prof_bytes(gather_gemm_alg_bytes(p, 4));
      prof_end(s);
      return check_launch("pw_gemm");
    }
    if (route == GG_ROUTE_K16) {
#define LAUNCH_K16(E) hipLaunchKernelGGL((gather_gemm_k16_kernel<E>), dim3(grid), dim3(256), 0, s, p)
      GG_EPI_SWITCH(LAUNCH_K16)
#undef LAUNCH_K16
    } else if (route == GG_ROUTE_GLDS2) {
#define LAUNCH_GLDS2(E) hipLaunchKernelGGL((gather_gemm_glds2_kernel<128, 128, 2, 2, E>), dim3(grid), dim3(256), 0, s, p)
      GG_EPI_SWITCH(LAUNCH_GLDS2)
#undef LAUNCH_GLDS2
#ifdef R3M_PROBES
    } else if (!gg_use_glds()) {   // register-staged 128x128 kernel: A/B only (R3M_GG_GLDS=0)
#define LAUNCH_REG(E) hipLaunchKernelGGL((gather_gemm_kernel<128, 128, 2, 2, E>), dim3(grid), dim3(256), 0, s, p)
      GG_EPI_SWITCH(LAUNCH_REG)
#undef LAUNCH_REG
#endif
    } else {                       // odd Ci/32 (not a ResNet shape; reachable through r3m_conv2d_fwd / r3m_linear_fwd): generic direct-to-LDS kernel
#define LAUNCH_GLDS(E) hipLaunchKernelGGL((gather_gemm_glds_kernel<E>), dim3(grid), dim3(256), 0, s, p)
      GG_EPI_SWITCH(LAUNCH_GLDS)
#undef LAUNCH_GLDS
    }
  } else {
    const int grid = ceil_div(p.M, 256) * ceil_div(p.Nc, 64);
    prof_begin(KC_GEMM_NARROW, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    if (route > GG_ROUTE_PW && route < GG_ROUTE_K16) {   // 64-wide output: persistent kernel, eight-wave 256 x 64 tile
      if (int e = launch_pw_gemm(p, s)) return e;
      prof_bytes(gather_gemm_alg_bytes(p, 4));
      prof_end(s);
      return check_launch("pw_gemm");
    }
    if (route == GG_ROUTE_GLDS2) {
#define LAUNCH_NARROW2(E) hipLaunchKernelGGL((gather_gemm_glds2_kernel<256, 64, 4, 1, E>), dim3(grid), dim3(256), 0, s, p)
      GG_EPI_SWITCH(LAUNCH_NARROW2)
#undef LAUNCH_NARROW2
    } else {
#define LAUNCH_NARROW(E) hipLaunchKernelGGL((gather_gemm_kernel<256, 64, 4, 1, E>), dim3(grid), dim3(256), 0, s, p)
      GG_EPI_SWITCH(LAUNCH_NARROW)
#undef LAUNCH_NARROW
    }
  }
  prof_bytes(gather_gemm_alg_bytes(p, 4));
  prof_end(s);
  return check_launch("gather_gemm");
}

#ifdef R3M_PROBES   // register-staged predecessor of wgrad_glds_kernel: kept for A/B in probe builds (R3M_WG_GLDS=0), not shipped
// =====================================================================================================
// wgrad: dW[co, tap, ci] = sum_m dY[m, co] * X[pix(m) + off(tap), ci].  GEMM M' = Co tile, N' = Ci tile,
// K' = rows m (split over blockIdx.y). Both operands arrive row(m)-major with channels contiguous, which is exactly
// the [k][i] LDS image the 32x32x2 MFMA wants for conflict-free ds_read_b32 fragment reads.
// Staging is branch-free (clamped addresses + select-to-zero at the LDS write); the (n, oy, ox) decode of the rows a thread
// stages is advanced incrementally (+32 rows per K step) instead of dividing.
// =====================================================================================================
template <int BMt, int BNt>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int BK = 32;
  constexpr int TM = BMt / 64, TN = BNt / 64;
  constexpr int AJ = BMt / 32, BJ = BNt / 32;
  constexpr int A_F4 = BMt / 4, B_F4 = BNt / 4;        // float4 per staged row
  constexpr int A_RPP = 256 / A_F4, B_RPP = 256 / B_F4;  // rows per pass
  __shared__ __attribute__((aligned(16))) float smem[BK * (BMt + BNt)];
  float* sA = smem;
  float* sB = smem + BK * BMt;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int T = p.KH * p.KW;
  const int lid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int bx = lid % p.gx, by = lid / p.gx;   // by = split index: consecutive logical blocks read the same rows
  const int tap = bx % T;  // the taps of one (co, ci) tile are neighbours: they re-read the same dY rows
  const int tile = bx / T;
  const int tn_ = tile % p.tilesN, tm_ = tile / p.tilesN;
  const int co0 = tm_ * BMt, ci0 = tn_ * BNt;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int ms = by * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);

  const int a_c = (tid % A_F4) * 4, a_r = tid / A_F4;
  const int b_c = (tid % B_F4) * 4, b_r = tid / B_F4;
  const bool a_cv = (co0 + a_c) < p.Co;
  const bool b_cv = (ci0 + b_c) < p.Ci;
  const int a_col = a_cv ? co0 + a_c : 0;
  const int b_col = b_cv ? ci0 + b_c : 0;
  const int hw = p.Ho * p.Wo;

  const int q32 = 32 / p.Wo, r32 = 32 - q32 * p.Wo;
  const bool fast_adv = (q32 + 1) <= p.Ho;        // one conditional subtract per axis is enough
  const long long img = (long long)p.Hi * p.Wi * p.Ci;
  long long xb[BJ];
  int xoy[BJ], xox[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int m = ms + b_r + j * B_RPP;
    if (p.simple_rows) {
      xb[j] = 0; xoy[j] = 0; xox[j] = 0;
    } else {
      const int n = m / hw;
      const int rem = m - n * hw;
      xoy[j] = rem / p.Wo;
      xox[j] = rem - xoy[j] * p.Wo;
      xb[j] = (long long)n * img;
    }
  }

  f32x4 ra[AJ], rb[BJ];
  unsigned a_ok = 0, b_ok = 0;
  auto load_tile = [&](int mk) {
    unsigned oka = 0, okb = 0;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = mk + a_r + j * A_RPP;
      const bool ok = (m < me) && a_cv;
      oka |= (ok ? 1u : 0u) << j;
      const int mc = min(m, me - 1);
      ra[j] = ldg4(p.dY + (long long)mc * p.Co + a_col);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int m = mk + b_r + j * B_RPP;
      const bool mok = (m < me) && b_cv;
      long long off;
      bool in = true;
      if (p.simple_rows) {
        off = (long long)min(m, me - 1) * p.Ci;
      } else {
        const int iy = xoy[j] * p.stride + kh - p.pad, ix = xox[j] * p.stride + kw - p.pad;
        in = ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
        const int iyc = min(max(iy, 0), p.Hi - 1), ixc = min(max(ix, 0), p.Wi - 1);
        off = xb[j] + ((long long)iyc * p.Wi + ixc) * p.Ci;
        off = (m < me) ? off : 0;       // rows past the split: any valid address, masked below
      }
      okb |= ((mok && in) ? 1u : 0u) << j;
      rb[j] = ldg4(p.X + off + b_col);
    }
    a_ok = oka; b_ok = okb;
    if (!p.simple_rows) {               // advance the decode to the next K step (+32 rows)
      if (fast_adv) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
          int ox = xox[j] + r32, oy = xoy[j] + q32;
          const bool cx = ox >= p.Wo;
          ox = cx ? ox - p.Wo : ox;
          oy = cx ? oy + 1 : oy;
          const bool cy = oy >= p.Ho;
          oy = cy ? oy - p.Ho : oy;
          xb[j] = cy ? xb[j] + img : xb[j];
          xox[j] = ox; xoy[j] = oy;
        }
      } else {
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
          const int m = mk + 32 + b_r + j * B_RPP;
          const int n = m / hw;
          const int rem = m - n * hw;
          xoy[j] = rem / p.Wo;
          xox[j] = rem - xoy[j] * p.Wo;
          xb[j] = (long long)n * img;
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lrow = lane & 31, lh = lane >> 5;
  const float* fragA = sA + lh * BMt + wm * TM * 32 + lrow;
  const float* fragB = sB + lh * BNt + wn * TN * 32 + lrow;

  if (ms < me) load_tile(ms);
  for (int mk = ms; mk < me; mk += BK) {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<f32x4*>(sA + (a_r + j * A_RPP) * BMt + a_c) = ((a_ok >> j) & 1u) ? ra[j] : zero4;
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<f32x4*>(sB + (b_r + j * B_RPP) * BNt + b_c) = ((b_ok >> j) & 1u) ? rb[j] : zero4;
    __syncthreads();
    if (mk + BK < me) load_tile(mk + BK);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = fragA[kk * 2 * BMt + t * 32];
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = fragB[kk * 2 * BNt + t * 32];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
    __syncthreads();
  }

  float* out = p.out + (long long)by * p.Co * T * p.Ci;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (co >= p.Co) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int ci = ci0 + (wn * TN + tn) * 32 + lrow;
        if (ci < p.Ci) out[((long long)co * T + tap) * p.Ci + ci] = acc[tm][tn][r];
      }
    }
}

#endif  // R3M_PROBES

// =====================================================================================================
// wgrad, direct-to-LDS staging through BUFFER addressing (round 3). Same GEMM as the register-staged probe kernel; the
// [k][channel] LDS image is lane-linear (a row of 64 or 128 floats = 256/512 B, one DMA instruction covers 4 or 2 consecutive
// k rows), so no swizzle is needed and the b32 fragment reads stay conflict-free. Two stages, one barrier per K step.
//
// Why buffer addressing: on gfx950 the f32-input MFMA shares the SIMD's fp32 lanes with the VALU, so every vector
// instruction in the K loop costs matrix time. `global_load_lds` needs a 64-bit per-lane address, i.e. per DMA
// piece a 64-bit add, the (oy, ox) walk, four compares and a pointer select in VALU — ~25 vector instructions per X piece of a
// 3x3 convolution, ~125 per 64 MFMAs. `buffer_load_dwordx4 ... lds` takes a wave-uniform 128-bit descriptor (base, bytes) in
// SGPRs plus a 32-bit per-lane offset, and lanes whose offset is >= the descriptor's byte count land ZEROS in LDS
// (tools/micro/bufload.hip). So:
//   * dY, and X of 1x1/stride-1 convolutions (rows are linear in m): the per-lane offset is a CONSTANT; a K step advances the
//     descriptor base by 32 rows and shrinks its byte count with four scalar instructions — rows past the end of the split
//     fall off the descriptor and read zeros. No vector instruction per piece at all.
//   * X of 3x3 / strided convolutions: a DMA instruction covers only 2 (128-wide tile) or 4 (64-wide) consecutive rows m, and
//     which rows is wave-uniform — the (frame, oy, ox) walk, the tap shift and the padding test of every staged row run on the
//     SCALAR unit (in the shadow of the MFMAs); a padding row gets an out-of-range offset. Per lane: pick its row's scalar
//     offset and add the channel offset = 3 (or 6) vector instructions per piece.
// =====================================================================================================
// NT = 3 ("kernel rows", round 3): one block owns the THREE taps (kh, 0..2) of one row of a 3-wide kernel for its (co, ci) tile,
// with three accumulator sets: the dY rows of a K step are staged and read from LDS ONCE for the three taps, the scalar cursor
// walk is shared (the taps differ by one pixel in x) — 2/3 of the DMA instructions and fragment reads per MFMA of the per-tap
// form. K steps of 16 rows keep the two stages at 64 KB (128-wide tile: 2 blocks per CU as before).
template <int BMt, int BNt, int BK = 32, int NT = 1, int IL = -1, int SR = -1>   // IL: DMA pieces spread between the MFMAs (1), in one burst (0), or p.interleave (-1); SR: 1x1 "simple rows" known at compile time (1 / 0) or p.simple_rows (-1)
__global__ __launch_bounds__(256, (NT == 3 && BMt == 128) ? 2 : 1) void wgrad_glds_kernel(const WgradParams p) {
  static_assert(BK == 32 || BK == 16, "K step of 32 or 16 rows");
  static_assert(NT == 1 || NT == 3, "one tap, or the three taps of a kernel row");
  // 128x128: waves 1 x 4, each 128 (co, interleaved: MFMA tile tm owns channels 4*i + tm) x 32 (ci) -> the A fragment of all
  // four tiles is ONE ds_read_b128 per K pair; 64x64: waves 2 x 2, each 32 x 32.
  constexpr bool WIDE = (BMt == 128);
  constexpr int TM = WIDE ? 4 : BMt / 64, TN = WIDE ? 1 : BNt / 64;
  constexpr int WR = BK / 4;                            // k rows staged per wave per stage
  constexpr int A_RPI = 256 / BMt, B_RPI = 256 / BNt;   // k rows covered by one 1 KiB DMA instruction
  constexpr int AJ = WR / A_RPI, BJ = WR / B_RPI;       // DMA pieces per wave per stage (a B piece = NT instructions)
  static_assert(AJ >= 1 && BJ >= 1, "a wave stages whole DMA instructions");
  constexpr int B_TILE = BK * BNt;                      // floats of one tap's X tile
  constexpr int STAGE = BK * BMt + NT * B_TILE;
  __shared__ __attribute__((aligned(128))) float smem[2 * STAGE];

  const bool simple_rows = NT == 1 && (SR < 0 ? p.simple_rows != 0 : SR != 0);   // a kernel-row block (NT = 3) never has 1x1 "simple" rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WIDE ? 0 : (wave_s >> 1), wn = WIDE ? wave_s : (wave_s & 1);
  const int T = p.KH * p.KW;
  const int TG = T / NT;                                // tap groups per tile (NT = 3: kernel rows)
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

  // lane -> (k row within the instruction, first channel); the per-lane offsets below never change in the K loop
  const int a_k = lane / (BMt / 4), a_c = (lane % (BMt / 4)) * 4;
  const int b_k = lane / (BNt / 4), b_c = (lane % (BNt / 4)) * 4;
  const unsigned a_chan = (co0 + a_c) < p.Co ? (unsigned)a_c * 4u : BUF_OOB;
  const unsigned b_chan = (ci0 + b_c) < p.Ci ? (unsigned)b_c * 4u : BUF_OOB;

  // A operand (dY): descriptor = [row ms + BK*step, end of the split) x channels from co0
  const float* a_base = p.dY + (long long)ms * p.Co + co0;
  int a_left = (int)(((long long)(me - ms) * p.Co - co0) * 4);      // bytes (host: a split spans < 2 GB)
  const int a_stepb = BK * p.Co * 4;
  unsigned a_voff[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) a_voff[j] = (unsigned)((wave_s * WR + j * A_RPI + a_k) * p.Co) * 4u + a_chan;

  // B operand (X)
  const long long img = (long long)p.Hi * p.Wi * p.Ci;
  const float* b_base;
  int b_left;
  const int b_stepb = BK * p.Ci * 4;
  unsigned b_voff[BJ];             // simple rows: constant per-lane offsets
  // non-simple rows: ONE scalar cursor (frame offset, oy, ox) that walks the WR consecutive rows this wave stages per K step,
  // then jumps the BK - WR rows to its rows of the next step; `c_left` = rows from the cursor to the end of the split
  int c_ox = 0, c_oy = 0, c_left = 0;
  unsigned c_f = 0;                // byte offset of the cursor row's frame from b_base
  constexpr int JUMP = BK - WR;
  const int qj = JUMP / p.Wo, rj = JUMP - qj * p.Wo;
  const unsigned imgb = (unsigned)(img * 4);
  const int pixb = p.Ci * 4;       // bytes between the X rows of neighbouring taps (one pixel)
  const int kh_p = kh - p.pad, kw_p = kw0 - p.pad;
  if (simple_rows) {
    b_base = p.X + (long long)ms * p.Ci + ci0;
    b_left = (int)(((long long)(me - ms) * p.Ci - ci0) * 4);
#pragma unroll
    for (int j = 0; j < BJ; ++j) b_voff[j] = (unsigned)((wave_s * WR + j * B_RPI + b_k) * p.Ci) * 4u + b_chan;
  } else {
    const int n0 = ms / hw;        // first frame of the split: 32-bit offsets are relative to it
    b_base = p.X + (long long)n0 * img + ci0;
    const long long rest = ((long long)(p.N - n0) * img - ci0) * 4;
    b_left = rest < (long long)BUF_OOB ? (int)rest : (int)BUF_OOB;
#pragma unroll
    for (int j = 0; j < BJ; ++j) b_voff[j] = 0;
    const int m = ms + wave_s * WR;
    const int n = m / hw;
    const int rem = m - n * hw;
    c_oy = rem / p.Wo;
    c_ox = rem - c_oy * p.Wo;
    c_f = (unsigned)(n - n0) * imgb;
    c_left = me - m;
  }
  bool b_is[B_RPI];                // lane masks: "my row is sub-row r of the instruction"
#pragma unroll
  for (int r = 0; r < B_RPI; ++r) b_is[r] = (b_k == r);

  // one DMA piece (pc < AJ: dY rows, else X rows of all NT taps) into `stage`
  auto issue_piece = [&](int stage, auto pc_c) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_c)::value;
    if (R3M_PROBE(p) & 1) return;                       // timing probes (probe builds only; wrong results)
    if ((R3M_PROBE(p) & 2) && pc >= AJ) return;
    if ((R3M_PROBE(p) & 4) && pc < AJ) return;
    if constexpr (pc < AJ) {
      constexpr int j = pc;
      float* la = smem + stage * STAGE + wave_s * WR * BMt;
      buf_dma16(a_base, a_left, la + j * A_RPI * BMt, a_voff[j]);
    } else {
      constexpr int j = pc - AJ;
      float* lb = smem + stage * STAGE + BK * BMt + wave_s * WR * BNt + j * B_RPI * BNt;
      if (simple_rows) {
        buf_dma16(b_base, b_left, lb, b_voff[j]);
      } else {
        unsigned so[NT][B_RPI];
#pragma unroll
        for (int r = 0; r < B_RPI; ++r) {     // scalar unit: tap shift, padding tests, row offset, cursor to the next row
          const int iy = c_oy * p.stride + kh_p, ix0 = c_ox * p.stride + kw_p;
          const bool rowok = ((unsigned)iy < (unsigned)p.Hi) && (c_left > 0);
          const unsigned off0 = c_f + (unsigned)((iy * p.Wi + ix0) * p.Ci) * 4u;
#pragma unroll
          for (int t = 0; t < NT; ++t)
            so[t][r] = (rowok && (unsigned)(ix0 + t) < (unsigned)p.Wi) ? off0 + (unsigned)(t * pixb) : BUF_OOB;
          c_left -= 1;
          c_ox += 1;
          if (c_ox == p.Wo) {
            c_ox = 0;
            c_oy += 1;
            if (c_oy == p.Ho) { c_oy = 0; c_f += imgb; }
          }
        }
        if constexpr (j == BJ - 1) {          // the wave's rows of this K step are issued: jump to its rows of the next one
          c_left -= JUMP;
          c_ox += rj;
          if (c_ox >= p.Wo) { c_ox -= p.Wo; c_oy += 1; }
          c_oy += qj;
          while (c_oy >= p.Ho) { c_oy -= p.Ho; c_f += imgb; }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          unsigned voff = so[t][0];
#pragma unroll
          for (int r = 1; r < B_RPI; ++r) voff = b_is[r] ? so[t][r] : voff;
          buf_dma16(b_base, b_left, lb + t * B_TILE, voff + b_chan);
        }
      }
    }
  };
  // after the last piece of a K step: both descriptors move on by BK rows (scalar)
  auto advance = [&]() __attribute__((always_inline)) {
    a_base += BK * p.Co;
    a_left = a_left > a_stepb ? a_left - a_stepb : 0;
    if (simple_rows) {
      b_base += BK * p.Ci;
      b_left = b_left > b_stepb ? b_left - b_stepb : 0;
    }
  };
  auto issue = [&](int stage) __attribute__((always_inline)) {
    static_for<AJ + BJ>([&](auto pc_c) __attribute__((always_inline)) { issue_piece(stage, pc_c); });
    advance();
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

  const int lrow = lane & 31, lh = lane >> 5;
  const float* fragA = smem + lh * BMt + (WIDE ? 4 * lrow : wm * TM * 32 + lrow);
  const float* fragB = smem + BK * BMt + lh * BNt + wn * TN * 32 + lrow;
  // MFMAs of one stage; when dma_stage >= 0 the next K step's DMA pieces are spread between them so that their issue cost
  // hides behind this wave's own MFMAs
  auto mfma_stage = [&](const float* fa, const float* fb, int dma_stage) __attribute__((always_inline)) {
    constexpr int NP = AJ + BJ;
    constexpr int EVERY = (BK / 2) / NP;     // K pairs between two pieces
    static_assert(EVERY >= 1, "at most one DMA piece per K pair");
    static_for<BK / 2>([&](auto kk_c) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_c)::value;
      float a[TM];
      if constexpr (WIDE) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(fa + kk * 2 * BMt);
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = a4[t];
      } else {
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = fa[kk * 2 * BMt + t * 32];
      }
#pragma unroll
      for (int tp = 0; tp < NT; ++tp) {
        float b[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = fb[tp * B_TILE + kk * 2 * BNt + t * 32];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tp][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tp][tm][tn], 0, 0, 0);
      }
      if constexpr ((kk % EVERY) == EVERY - 1 && kk / EVERY < NP) {
        if (dma_stage >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(dma_stage, std::integral_constant<int, kk / EVERY>{});
          if constexpr (kk / EVERY == NP - 1) advance();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
  };

  const int nk = (me - ms + BK - 1) / BK;
  if (nk > 0) issue(0);
  int kt = 0;
  const bool il = IL < 0 ? (p.interleave != 0) : (IL != 0);
  for (; kt + 1 < nk; kt += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (il) {
      mfma_stage(fragA, fragB, 1);
    } else {
      issue(1);
      mfma_stage(fragA, fragB, -1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (il) {
      mfma_stage(fragA + STAGE, fragB + STAGE, (kt + 2 < nk) ? 0 : -1);
    } else {
      if (kt + 2 < nk) issue(0);
      mfma_stage(fragA + STAGE, fragB + STAGE, -1);
    }
  }
  if (kt < nk) {   // odd tail
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    mfma_stage(fragA, fragB, -1);
  }

  float* out = p.out + (long long)by * p.Co * T * p.Ci;
#pragma unroll
  for (int tp = 0; tp < NT; ++tp)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = WIDE ? co0 + 4 * rho + tm : co0 + (wm * TM + tm) * 32 + rho;
        if (co >= p.Co) continue;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int ci = ci0 + (wn * TN + tn) * 32 + lrow;
          if (ci < p.Ci) out[((long long)co * T + tap0 + tp) * p.Ci + ci] = acc[tp][tm][tn][r];
        }
      }
}

// =====================================================================================================
// Stem, direct: x/255 -> Normalize -> conv 7x7 stride 2 pad 3, 3 -> 64 channels (the reference's first three steps,
// /root/reference/r3m/models/models_r3m.py:97-99 into torchvision's conv1), forward and weight gradient, straight from the
// NCHW fp32 frames — no im2col matrix in HBM (that cost 8 MB written + 16 MB re-read per frame).
// Geometry trick: for a fixed kernel row kh the 7 x 3 (kw, c) taps of one output pixel are 21 CONSECUTIVE floats of an
// interleaved [x][c] image row, starting at 6*ox. So with the (normalised, zero-padded) input rows staged in LDS as
// patch[y][(ix+3)*3 + c], the MFMA A-fragment of output pixel (oy, ox) for k = (kh, j) is patch[2*oy + kh][6*ox + j]:
// a per-lane base plus an immediate — no address arithmetic in the K loop. K is walked as 7 x 22 (j = 21 multiplies a
// zero weight), i.e. 154 instead of 147 MACs per output: 5 % padding instead of im2col's 160.
// =====================================================================================================
constexpr int ST_PS = 692;      // patch row stride (forward): 230 pixels x 3 channels (+2 pad)
constexpr int ST_PSW = 694;     // patch row stride (weight gradient): == 22 (mod 32). There 32 lanes read patch[kh * stride + jj] for 32
                                // CONSECUTIVE k = 22 kh + jj, which cross a kernel-row boundary; with 692 (== 20 mod 32) the lanes of
                                // the next kernel row landed on the banks of jj = 20, 21 (2-way conflict on every B read: PMC
                                // lds_conflict_frac 0.44, round 2); with 694 the bank is k mod 32 — conflict-free
constexpr int ST_KS = 155;      // LDS weight row stride (odd: conflict-free fragment reads)
constexpr int ST_K = 154;       // 7 kernel rows x 22

// pre-pass: frames NCHW fp32 0..255 -> normalised, channel-interleaved rows xn[f][iy][ix*3 + c] (exactly the reference's
// (x/255 - mean)/std with IEEE divisions, done once per frame; both stem kernels then stage plain row copies)
__global__ __launch_bounds__(256) void stem_prep_kernel(const float* __restrict__ x, float* __restrict__ xn, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread per (f, iy, ix)
  if (i >= total) return;
  const int ix = (int)(i % 224);
  const long long t = i / 224;
  const int iy = (int)(t % 224);
  const long long f = t / 224;
  float* o = xn + i * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = stem_normalize(x[((f * 3 + c) * 224 + iy) * 224 + ix], c);
}

// the same pre-pass reading the RAW clips through their crop boxes (rc / rctraj on the GPU): the cropped fp32
// frames [F,3,224,224] are never written — one gather-bilinear pass from uint8 (or float) straight into the normalised image
template <typename T>
__global__ __launch_bounds__(256) void stem_prep_crop_kernel(const T* __restrict__ raw, const int* __restrict__ boxes,
                                                              float* __restrict__ xn, long long total, int Hi, int Wi, int fpb) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread per (f, iy, ix)
  if (i >= total) return;
  const int ix = (int)(i % 224);
  const long long t = i / 224;
  const int iy = (int)(t % 224);
  const long long f = t / 224;
  const int* b = boxes + (f / fpb) * 4;
  const int top = b[0], left = b[1], bh = b[2], bw = b[3];
  float* o = xn + i * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    o[c] = stem_normalize(bilinear_sample(raw + (f * 3 + c) * (long long)Hi * Wi, Wi, top, left, bh, bw, iy, ix, 0, 0, 224, 224), c);
}

int launch_stem_prep_crop(const FrameSource& src, float* xn, int F, hipStream_t s) {
  const long long total = (long long)F * 224 * 224;
  if (src.is_u8)
    hipLaunchKernelGGL((stem_prep_crop_kernel<unsigned char>), dim3(ceil_div(total, 256)), dim3(256), 0, s,
                       static_cast<const unsigned char*>(src.frames), src.boxes, xn, total, src.Hi, src.Wi, src.frames_per_box);
  else
    hipLaunchKernelGGL((stem_prep_crop_kernel<float>), dim3(ceil_div(total, 256)), dim3(256), 0, s,
                       static_cast<const float*>(src.frames), src.boxes, xn, total, src.Hi, src.Wi, src.frames_per_box);
  return check_launch("stem_prep_crop");
}

int launch_stem_prep(const float* x_nchw, float* xn, int F, hipStream_t s) {
  const long long total = (long long)F * 224 * 224;
  hipLaunchKernelGGL(stem_prep_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, x_nchw, xn, total);
  return check_launch("stem_prep");
}

// stage `nrows` input rows iy0.. of frame f into patch[y][9 zeros | 672 data | zeros]: float4 row copies
template <int PS = ST_PS>
__device__ __forceinline__ void stem_load_patch(const float* __restrict__ xn, float* patch, long long f, int iy0, int nrows) {
  constexpr int TAIL = PS - 681;      // zero floats behind the 672 data floats (9 in front)
  for (int i = threadIdx.x; i < nrows * (9 + TAIL); i += 256) {
    const int y = i / (9 + TAIL), e = i - y * (9 + TAIL);
    patch[y * PS + (e < 9 ? e : 672 + e)] = 0.f;
  }
  for (int i = threadIdx.x; i < nrows * 168; i += 256) {
    const int y = i / 168, q = i - y * 168;
    const int iy = iy0 + y;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < 224u) v = ldg4(xn + ((f * 224 + iy) * 224) * 3 + q * 4);
    float* d = patch + y * PS + 9 + q * 4;        // 9-float left border: not 16-byte aligned -> scalar LDS stores
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
}

template <int EPI, class OT>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ xn, const float* __restrict__ w,
                                                        const GatherGemmParams p, int ntiles) {
  constexpr int SMEM = 13 * ST_PS + 64 * ST_KS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* patch = smem;                 // also the epilogue's scratch (8704 floats < 13*ST_PS): the weights behind it survive
  float* wl = smem + 13 * ST_PS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {   // weights once per (persistent) block
    const int n = tid >> 2, q = tid & 3;            // 4 threads per output channel
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
      for (int j = q; j < 22; j += 4) wl[n * ST_KS + kh * 22 + j] = (j < 21) ? w[n * 147 + kh * 21 + j] : 0.f;
  }
  const int lrow = lane & 31, lh = lane >> 5;
  int b_base[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) b_base[t] = (t * 32 + lrow) * ST_KS + lh;

  // Round 3: register prefetch — the 13 input rows of the NEXT tile (2184 float4, 9 per thread) are requested before this tile's
  // MFMAs and written to LDS after its epilogue (whose slabs alias the patch), so their latency rides under the matrix work.
  constexpr int PQ = (13 * 168 + 255) / 256;
  f32x4 pre[PQ];
  auto request = [&](int blk) __attribute__((always_inline)) {
    const long long f = blk / 49;
    const int iy0 = 2 * (((blk - (int)f * 49) * 256) / 112) - 3;
#pragma unroll
    for (int k = 0; k < PQ; ++k) {
      const int i = tid + 256 * k;
      const int y = i / 168, q = i - y * 168;
      const int iy = iy0 + y;
      pre[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < 13 * 168 && (unsigned)iy < 224u) pre[k] = ldg4(xn + ((f * 224 + iy) * 224) * 3 + q * 4);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
    for (int i = tid; i < 13 * 20; i += 256) {       // zero borders: the epilogue's slabs overwrote them
      const int y = i / 20, e = i - y * 20;
      patch[y * ST_PS + (e < 9 ? e : 672 + e)] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < PQ; ++k) {
      const int i = tid + 256 * k;
      if (i < 13 * 168) {
        const int y = i / 168, q = i - y * 168;
        float* d = patch + y * ST_PS + 9 + q * 4;     // 9-float left border: not 16-byte aligned -> scalar LDS stores
        d[0] = pre[k][0]; d[1] = pre[k][1]; d[2] = pre[k][2]; d[3] = pre[k][3];
      }
    }
  };
  if ((int)blockIdx.x < ntiles) {
    request(blockIdx.x);
    commit();
  }
  for (int blk = blockIdx.x; blk < ntiles; blk += gridDim.x) {
    const long long f = blk / 49;
    const int lm0 = (blk - (int)f * 49) * 256;    // first output pixel of this tile inside its frame (12544 = 49 * 256)
    const int oy0 = lm0 / 112;
    __syncthreads();                              // the patch of this tile (and, first time, the weights) is in LDS
    const int nblk = blk + gridDim.x;
    if (nblk < ntiles) request(nblk);
    int a_base[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int lm = lm0 + wave * 64 + t * 32 + lrow;
      const int oy = lm / 112, ox = lm - oy * 112;
      a_base[t] = 2 * (oy - oy0) * ST_PS + 6 * ox + lh;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
      for (int jp = 0; jp < 11; ++jp) {
        float a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = patch[a_base[t] + kh * ST_PS + 2 * jp];
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = wl[b_base[t] + kh * 22 + 2 * jp];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
    __syncthreads();
    gg_epilogue<256, 64, 4, 1, EPI, 13 * ST_PS, OT>(p, acc, smem, blk * 256, 0, blk);
    __syncthreads();   // the epilogue slabs alias the patch that is refilled now
    if (nblk < ntiles) commit();
  }
}

int launch_stem_fwd(const float* x_nchw, const float* w147, void* y, float* stats, int F, int dt, hipStream_t s) {
  GatherGemmParams p;
  memset(&p, 0, sizeof p);
  p.out = static_cast<float*>(y); p.stats = stats; p.dtype = dt;
  p.M = F * 12544; p.Nc = 64; p.os = 1;
  p.Hg = 112; p.Wg = 112; p.Ho = 112; p.Wo = 112;
  const double flops = 2.0 * (double)p.M * 64.0 * 147.0;
  prof_begin(KC_GEMM_NARROW, flops, p.M, 64, 147, 1, s);
  prof_bytes((double)F * 224 * 224 * 3 * 4 + (double)p.M * 64 * (dt == DT_BF16 ? 2 : 4));
  const int ntiles = F * 49;
  const int grid = ntiles < 512 ? ntiles : 512;   // persistent blocks (2 per CU): the 39 KB weight image is staged once per block
  if (dt == DT_BF16) {
    if (stats) hipLaunchKernelGGL((stem_fwd_kernel<EPI_STATS, bf16_t>), dim3(grid), dim3(256), 0, s, x_nchw, w147, p, ntiles);
    else hipLaunchKernelGGL((stem_fwd_kernel<0, bf16_t>), dim3(grid), dim3(256), 0, s, x_nchw, w147, p, ntiles);
  } else {
    if (stats) hipLaunchKernelGGL((stem_fwd_kernel<EPI_STATS, float>), dim3(grid), dim3(256), 0, s, x_nchw, w147, p, ntiles);
    else hipLaunchKernelGGL((stem_fwd_kernel<0, float>), dim3(grid), dim3(256), 0, s, x_nchw, w147, p, ntiles);
  }
  prof_end(s);
  return check_launch("stem_fwd");
}

// dW[co][kh*22 + j] partial of one block = sum over its output image rows of dY[m][co] * patch(m, kh, j).
// One output image row (112 pixels = 56 K pairs) per iteration: 7 input rows + the dY row in LDS, per-lane bases plus
// immediates (pixel step = 6 floats of the interleaved row). Waves: 2 (co halves) x 2 (k tiles {0,1,2} / {3,4}).
template <class T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dY,
                                                          float* __restrict__ partial, int total_rows) {
  __shared__ __attribute__((aligned(16))) float smem[112 * 64 + 7 * ST_PSW];
  float* dys = smem;                  // 16-byte aligned (float4 stores); the patch takes scalar stores
  float* patch = smem + 112 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int lrow = lane & 31, lh = lane >> 5;
  const int jt0 = wj ? 3 : 0;
  const int a_base = lh * 64 + wi * 32 + lrow;
  int b_base[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    int j = (jt0 + t) * 32 + lrow;
    if (j >= ST_K) j = 0;                       // columns 154..159 (and the unused third tile of the second wave column)
    const int kh = j / 22, jj = j - kh * 22;
    b_base[t] = kh * ST_PSW + jj + 6 * lh;
  }
  // two-level summation: `acc` covers one image row (112 products per element), `tot` adds the rows — short fp32 chains
  f32x16 acc[3], tot[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[t][r] = 0.f;

  // Round 3: register prefetch. The next output row's operands (7 input rows = 1176 float4, the dY row = 1792 float4: 5 + 7 per
  // thread) are requested BEFORE this row's MFMAs and written to LDS after them, so their global latency (~2 us of the ~6.5 us a
  // row took) rides under the matrix work instead of in front of it. The zero borders of the patch rows never change: written once.
  constexpr int PQ = (7 * 168 + 255) / 256, DQ = 112 * 16 / 256;     // 5, 7
  f32x4 pre_p[PQ], pre_d[DQ];
  auto request = [&](int row) __attribute__((always_inline)) {
    const long long f = row / 112;
    const int iy0 = 2 * (row - (int)f * 112) - 3;
#pragma unroll
    for (int k = 0; k < PQ; ++k) {
      const int i = tid + 256 * k;
      const int y = i / 168, q = i - y * 168;
      const int iy = iy0 + y;
      pre_p[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < 7 * 168 && (unsigned)iy < 224u) pre_p[k] = ldg4(x + ((f * 224 + iy) * 224) * 3 + q * 4);
    }
    const T* src = dY + (long long)row * 112 * 64;
#pragma unroll
    for (int k = 0; k < DQ; ++k) pre_d[k] = ld4t(src + (tid + 256 * k) * 4);
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PQ; ++k) {
      const int i = tid + 256 * k;
      if (i < 7 * 168) {
        const int y = i / 168, q = i - y * 168;
        float* d = patch + y * ST_PSW + 9 + q * 4;     // 9-float left border: not 16-byte aligned -> scalar LDS stores
        d[0] = pre_p[k][0]; d[1] = pre_p[k][1]; d[2] = pre_p[k][2]; d[3] = pre_p[k][3];
      }
    }
#pragma unroll
    for (int k = 0; k < DQ; ++k) *reinterpret_cast<f32x4*>(dys + (tid + 256 * k) * 4) = pre_d[k];
  };
  {
    constexpr int TAIL = ST_PSW - 681;
    for (int i = tid; i < 7 * (9 + TAIL); i += 256) {
      const int y = i / (9 + TAIL), e = i - y * (9 + TAIL);
      patch[y * ST_PSW + (e < 9 ? e : 672 + e)] = 0.f;
    }
  }
  int row = blockIdx.x;
  if (row < total_rows) {
    request(row);
    commit();
  }
  __syncthreads();
  for (; row < total_rows; row += gridDim.x) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int next = row + gridDim.x;
    if (next < total_rows) request(next);
    if (wj == 0) {
#pragma unroll
      for (int q = 0; q < 56; ++q) {
        const float a = dys[a_base + q * 128];
#pragma unroll
        for (int t = 0; t < 3; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, patch[b_base[t] + q * 12], acc[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 56; ++q) {
        const float a = dys[a_base + q * 128];
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, patch[b_base[t] + q * 12], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) tot[t] += acc[t];
    __syncthreads();                      // every wave is done reading this row's tiles
    if (next < total_rows) commit();
    __syncthreads();
  }
  float* out = partial + (long long)blockIdx.x * 64 * 160;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (t == 2 && wj) continue;                 // the second wave column owns k tiles 3 and 4 only
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      out[co * 160 + (jt0 + t) * 32 + lrow] = tot[t][r];
    }
  }
}

// persistent blocks: the kernel holds 130 VGPRs + 48 AGPRs -> TWO blocks per CU; 768 blocks (round 1) ran as one and a half rounds
// of resident blocks with equal work each, i.e. the last third of the time at half occupancy
#ifndef R3M_STEM_WG_BLOCKS
#define R3M_STEM_WG_BLOCKS 512
#endif
constexpr int STEM_WG_BLOCKS = R3M_STEM_WG_BLOCKS;
size_t stem_wgrad_ws_floats() { return (size_t)STEM_WG_BLOCKS * 64 * 160; }

// dw147[co][kh*21 + j] (+)= dw160[co][kh*22 + j]
__global__ void stem_unpack_dw22_kernel(const float* __restrict__ dw160, float* __restrict__ dw147, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 147) return;
  const int co = i / 147, k = i - co * 147;
  const int kh = k / 21, j = k - kh * 21;
  const float v = dw160[co * 160 + kh * 22 + j];
  dw147[i] = accumulate ? dw147[i] + v : v;
}

int launch_stem_wgrad(const float* x_nchw, const void* dY, float* dw147, float* ws /* stem_wgrad_ws_floats() + 64*160 */, int F,
                      int accumulate, int dt, hipStream_t s) {
  const int total_rows = F * 112;
  const int nb = total_rows < STEM_WG_BLOCKS ? total_rows : STEM_WG_BLOCKS;
  const double flops = 2.0 * (double)F * 12544.0 * 64.0 * 147.0;
  prof_begin(KC_WGRAD_NARROW, flops, F * 12544, 64, 147, 1, s);
  prof_bytes((double)F * 224 * 224 * 3 * 4 + (double)F * 12544 * 64 * (dt == DT_BF16 ? 2 : 4));
  if (dt == DT_BF16)
    hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, x_nchw, static_cast<const bf16_t*>(dY), ws, total_rows);
  else
    hipLaunchKernelGGL((stem_wgrad_kernel<float>), dim3(nb), dim3(256), 0, s, x_nchw, static_cast<const float*>(dY), ws, total_rows);
  prof_end(s);
  if (int e = check_launch("stem_wgrad")) return e;
  float* dw160 = ws + stem_wgrad_ws_floats();
  if (int e = launch_wgrad_reduce(ws, dw160, 64 * 160, nb, 0, s)) return e;
  hipLaunchKernelGGL(stem_unpack_dw22_kernel, dim3(ceil_div(64 * 147, 256)), dim3(256), 0, s, dw160, dw147, accumulate);
  return check_launch("stem_unpack_dw22");
}

// debugging aid: resident blocks per CU the runtime predicts for the main kernel variants
int debug_occupancy(int* out4) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gather_gemm_glds2_kernel<128, 128, 2, 2, EPI_STATS>, 256, 0) != hipSuccess) return 1;
  out4[0] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gather_gemm_kernel<128, 128, 2, 2, EPI_STATS>, 256, 0) != hipSuccess) return 1;
  out4[1] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gather_gemm_glds2_kernel<256, 64, 4, 1, EPI_STATS>, 256, 0) != hipSuccess) return 1;
  out4[2] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_glds_kernel<128, 128>, 256, 0) != hipSuccess) return 1;
  out4[3] = n;
  return 0;
}

#ifdef R3M_PROBES
static bool wg_use_glds() {
  const int v = R3M_ENV_INT("R3M_WG_GLDS", 1) != 0;
  return v == 1;
}
#endif

static inline bool wg_wide(int Co, int Ci) { return (Co % 128 == 0) && (Ci % 128 == 0); }
// 3-wide kernels run one block per kernel row (wgrad_glds_kernel NT = 3). R3M_WG_ROWS=0 (probe builds): per-tap blocks.
static bool wg_rows(int KW) {
  const int v = R3M_ENV_INT("R3M_WG_ROWS", 1);
  return v && KW == 3;
}

// Split-K factor: enough blocks for two full waves of resident blocks (128x128: 2 blocks/CU x 256 CUs; 64x64: 5/CU), as few
// splits as that allows (every split writes and re-reads a full dW slab), never fewer than 8 K steps per block.
int wgrad_pick_split(int M, int Co, int Ci, int T) {
  const bool wide = wg_wide(Co, Ci);
  const int bt = wide ? 128 : 64;
  long long tiles = (long long)ceil_div(Co, bt) * ceil_div(Ci, bt) * T;
  if (T == 9 && wg_rows(3)) tiles /= 3;     // kernel-row blocks cover three taps each
  const int tgt = R3M_ENV_INT("R3M_WG_BLOCKS", 0);                       // probe builds: block target override
  const long long target = tgt > 0 ? tgt : (wide ? 1024 : 2560);
  long long split = target / tiles;   // floor: never spill a few blocks into an extra wave
  const long long max_split = (M + 255) / 256;
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  long long rps = ((M + split - 1) / split + 31) / 32 * 32;
  return ceil_div(M, rps);
}

int launch_wgrad(const WgradParams& p0, int splitK, hipStream_t s) {
  WgradParams p = p0;
  R3M_REQUIRE(p.Ci % 4 == 0 && p.Co % 4 == 0, "wgrad: channel counts must be multiples of 4 (Co=%d Ci=%d)", p.Co, p.Ci);
  R3M_REQUIRE(splitK >= 1, "wgrad: splitK=%d", splitK);
  p.rows_per_split = ((p.M + splitK - 1) / splitK + 31) / 32 * 32;
  R3M_REQUIRE(ceil_div(p.M, p.rows_per_split) == splitK, "wgrad: splitK=%d does not tile M=%d", splitK, p.M);
  {   // buffer addressing: a block's operands are reached through 32-bit offsets from the first row / frame of its split
    const long long lim = 0x7FFFF000LL;
    const long long a_span = (long long)p.rows_per_split * p.Co * 4;
    const long long frames = (long long)p.rows_per_split / ((long long)p.Ho * p.Wo) + 2;
    const long long b_span = p.simple_rows ? (long long)p.rows_per_split * p.Ci * 4 : frames * p.Hi * p.Wi * p.Ci * 4;
    R3M_REQUIRE(a_span < lim && b_span < lim, "wgrad: one split spans %lld / %lld bytes (limit 2 GiB): raise splitK (%d)", a_span, b_span, splitK);
  }
  const int T = p.KH * p.KW;
  {
    // DMA pieces spread between the MFMAs (one per 2 K pairs) or issued in one burst before them. Round 3, buffer addressing,
    // same box (profiles/r03_wgrad_buffer_ab.txt): spreading wins where a piece carries scalar work — the (oy, ox) walk of 3x3 /
    // strided X rows on a 128-wide tile (115.8 -> 118-123 TFLOP/s) — and loses where it does not (1x1: 134 -> 130) and on the
    // 64-wide tile (107.5 -> 100). Probe builds: R3M_WG_INTERLEAVE = 0 / 1 forces it.
    const int il = R3M_ENV_INT("R3M_WG_INTERLEAVE", -1);
    p.interleave = il >= 0 ? il : (!p.simple_rows && wg_wide(p.Co, p.Ci));
    const int xc = R3M_ENV_INT("R3M_WG_XCD", 1);
    p.xcd = xc;
    p.debug = R3M_ENV_INT("R3M_WG_DEBUG", 0);       // probe builds only (R3M_ENV_INT is the default in shipped builds)
  }
  const double flops = 2.0 * (double)p.M * (double)p.Co * (double)T * p.Ci;
  if (wg_wide(p.Co, p.Ci)) {
    p.tilesN = ceil_div(p.Ci, 128);
    const int tiles = ceil_div(p.Co, 128) * p.tilesN * T;
    prof_begin(KC_WGRAD_WIDE, flops, p.M, p.Co, p.Ci, T, s);
    p.gx = tiles;
#ifdef R3M_PROBES
    if (!wg_use_glds()) hipLaunchKernelGGL((wgrad_kernel<128, 128>), dim3(tiles * splitK), dim3(256), 0, s, p);
    else
#endif
    if (wg_rows(p.KW) && R3M_ENV_INT("R3M_WG_WIN", 1) && wgrad_rowwin_eligible(p)) {   // 3x3 "same" convolutions: shared input window (wgrad_win.hip)
      if (int e = launch_wgrad_rowwin(p, splitK, s)) return e;
    } else if (wg_rows(p.KW)) {   // 3-wide kernels: one block per kernel row (three taps), K steps of 16 rows
      p.gx = ceil_div(p.Co, 128) * p.tilesN * p.KH;
      hipLaunchKernelGGL((wgrad_glds_kernel<128, 128, 16, 3, 1>), dim3(p.gx * splitK), dim3(256), 0, s, p);
    } else
    {
#ifdef R3M_PROBES
      hipLaunchKernelGGL((wgrad_glds_kernel<128, 128>), dim3(tiles * splitK), dim3(256), 0, s, p);
#else
      // shipped builds: the two switches of the per-tap kernel are fixed per launch kind -> two compile-time variants
      if (p.simple_rows) hipLaunchKernelGGL((wgrad_glds_kernel<128, 128, 32, 1, 0, 1>), dim3(tiles * splitK), dim3(256), 0, s, p);
      else hipLaunchKernelGGL((wgrad_glds_kernel<128, 128, 32, 1, 1, 0>), dim3(tiles * splitK), dim3(256), 0, s, p);
#endif
    }
  } else {
    p.tilesN = ceil_div(p.Ci, 64);
    const int tiles = ceil_div(p.Co, 64) * p.tilesN * T;
    prof_begin(KC_WGRAD_NARROW, flops, p.M, p.Co, p.Ci, T, s);
    p.gx = tiles;
#ifdef R3M_PROBES
    if (!wg_use_glds()) hipLaunchKernelGGL((wgrad_kernel<64, 64>), dim3(tiles * splitK), dim3(256), 0, s, p);
    else
#endif
    if (wg_rows(p.KW) && R3M_ENV_INT("R3M_WG_WIN", 1) && wgrad_rowwin_eligible(p)) {
      if (int e = launch_wgrad_rowwin(p, splitK, s)) return e;
    } else if (wg_rows(p.KW)) {
      p.gx = ceil_div(p.Co, 64) * p.tilesN * p.KH;
      hipLaunchKernelGGL((wgrad_glds_kernel<64, 64, 16, 3, 0>), dim3(p.gx * splitK), dim3(256), 0, s, p);
    } else
    {
#ifdef R3M_PROBES
      hipLaunchKernelGGL((wgrad_glds_kernel<64, 64>), dim3(tiles * splitK), dim3(256), 0, s, p);
#else
      if (p.simple_rows) hipLaunchKernelGGL((wgrad_glds_kernel<64, 64, 32, 1, 0, 1>), dim3(tiles * splitK), dim3(256), 0, s, p);
      else hipLaunchKernelGGL((wgrad_glds_kernel<64, 64, 32, 1, 0, 0>), dim3(tiles * splitK), dim3(256), 0, s, p);
#endif
    }
  }
  prof_bytes(4.0 * ((double)p.M * p.Co + (double)p.N * p.Hi * p.Wi * p.Ci + (double)splitK * p.Co * T * p.Ci));
  prof_end(s);
  return check_launch("wgrad");
}

// dW[i] (+)= sum_s partial[s][i]   — fixed summation order: deterministic gradients
// dW[i] (+)= sum over split-K slices of partial[s][i], fixed order (deterministic). A block is TX float4 columns x TY slice
// groups (TX * TY = 256): group g adds slices g, g+TY, ...; the groups are combined through LDS in group order. Small weight
// tensors (e.g. 64x64x9: 9216 float4, 284 slices) get TY = 16 so that the launch has hundreds of blocks and short load chains
// instead of 36 blocks walking 284 slices one after the other (that was up to 1 ms per launch).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                            long long n4, long long n, int splitK, int accumulate, int tx_log2) {
  __shared__ f32x4 red[256];
  const int TX = 1 << tx_log2, TY = 256 >> tx_log2;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> tx_log2;
  const long long i = (long long)blockIdx.x * TX + tx;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (i < n4)
    for (int sidx = ty; sidx < splitK; sidx += TY) v += ldg4(partial + sidx * n + i * 4);
  if (TY == 1) {
    if (i < n4) {
      if (accumulate) v += *reinterpret_cast<const f32x4*>(dW + i * 4);
      *reinterpret_cast<f32x4*>(dW + i * 4) = v;
    }
    return;
  }
  red[threadIdx.x] = v;
  __syncthreads();
  if (ty == 0 && i < n4) {
    for (int g = 1; g < TY; ++g) v += red[(g << tx_log2) + tx];
    if (accumulate) v += *reinterpret_cast<const f32x4*>(dW + i * 4);
    *reinterpret_cast<f32x4*>(dW + i * 4) = v;
  }
}
int launch_wgrad_reduce(const float* partial, float* dW, long long n, int splitK, int accumulate, hipStream_t s) {
  R3M_REQUIRE(n % 4 == 0, "wgrad_reduce: n=%lld must be a multiple of 4", n);
  const long long n4 = n / 4;
  int tx_log2 = 8;                                     // TX = 256, TY = 1
  while (tx_log2 > 4 && ceil_div(n4, 1 << tx_log2) < 1024 && (256 >> tx_log2) * 2 <= splitK) --tx_log2;   // more slice groups for small tensors
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(n4, 1 << tx_log2)), dim3(256), 0, s, partial, dW, n4, n, splitK, accumulate, tx_log2);
  return check_launch("wgrad_reduce");
}

// Wt[ci][t][co] = W[co][t][ci]   (dgrad wants the contraction index co contiguous)
__global__ __launch_bounds__(256) void transpose_w_kernel(const float* __restrict__ W, float* __restrict__ Wt, int Co, int T, int Ci) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Co && ci < Ci) ? W[((long long)co * T + t) * Ci + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Ci && co < Co) Wt[((long long)ci * T + t) * Co + co] = tile[tx][r];
  }
}

// Every dgrad weight image of a network in ONE launch (the engine ran one 5 us transpose per conv layer and step: 52 launches for
// ResNet-50). tab[l] = {w_off, wt_off, Co, T, Ci}: Wt_l[ci][t][co] = W_l[co][t][ci] with W_l at params + w_off and Wt_l at
// wt + wt_off (elements of the output type); tile0[l] = first 32 x 32 tile (block) of layer l, tile0[n] = grid size.
template <class OT>
__global__ __launch_bounds__(256) void transpose_w_all_kernel(const float* __restrict__ params, OT* __restrict__ wt,
                                                               const WtEntry* __restrict__ tab, const int* __restrict__ tile0, int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;                 // last l with tile0[l] <= blockIdx.x (block-uniform binary search)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WtEntry e = tab[lo];
  const int local = blockIdx.x - tile0[lo];
  const int tci = (e.Ci + 31) / 32, tco = (e.Co + 31) / 32;
  const int ci0 = (local % tci) * 32, co0 = ((local / tci) % tco) * 32, t = local / (tci * tco);
  const float* W = params + e.w_off;
  OT* Wt = wt + e.wt_off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < e.Co && ci < e.Ci) ? W[((long long)co * e.T + t) * e.Ci + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < e.Ci && co < e.Co) Wt[((long long)ci * e.T + t) * e.Co + co] = (OT)tile[tx][r];
  }
}

int launch_transpose_w_all(const float* params, void* wt, const WtEntry* tab, const int* tile0, int n, int tiles, int dt, hipStream_t s) {
  if (n <= 0 || tiles <= 0) return 0;
  if (dt == DT_BF16)
    hipLaunchKernelGGL((transpose_w_all_kernel<bf16_t>), dim3(tiles), dim3(256), 0, s, params, static_cast<bf16_t*>(wt), tab, tile0, n);
  else
    hipLaunchKernelGGL((transpose_w_all_kernel<float>), dim3(tiles), dim3(256), 0, s, params, static_cast<float*>(wt), tab, tile0, n);
  return check_launch("transpose_w_all");
}

int launch_transpose_w(const float* W, float* Wt, int Co, int T, int Ci, hipStream_t s) {
  hipLaunchKernelGGL(transpose_w_kernel, dim3(ceil_div(Ci, 32), ceil_div(Co, 32), T), dim3(256), 0, s, W, Wt, Co, T, Ci);
  return check_launch("transpose_w");
}

}  // namespace r3m
