// r3m_amd — fp32 weight gradient of 3x3 / stride-1 / pad-1 convolutions with a SHARED INPUT WINDOW (round 4, gfx950).
//
//   dW[co, (kh, kw), ci] = sum_m dY[m, co] * X[pixel(m) + (kh - 1, kw - 1), ci]
//
// Reference call site: the backward of torchvision's conv2 (Bottleneck) / conv1, conv2 (BasicBlock) reached from
// /root/reference/r3m/trainer.py:146 (`full_loss.backward()`).
//
// The kernel-row form of wgrad_glds_kernel (conv.hip, NT = 3) stages the X rows of a K step once PER TAP — three tiles of 16 rows,
// each DMA instruction preceded by the scalar (frame, oy, ox) walk and padding tests of its rows and three vector instructions.
// Probe builds (tools/gpu_wg_probe.sh, R3M_WG_DEBUG=2): that X staging is 11 % of the launch (15 % on the 64-wide tile). But for a
// "same" convolution (Hi = Ho, Wi = Wo) the input pixel of GEMM row m and tap (kh, kw) is FLAT pixel m + (kh - 1) Wi + (kw - 1) of
// the NHWC tensor — linear in m across image rows and frames — so the three taps of a kernel row read ONE window of BK + 2
// consecutive pixels per K step, shifted by one row each:
//   * X arrives like dY: constant per-lane offsets, a descriptor that advances by BK rows per K step (4 scalar instructions), rows
//     before the tensor / past its end fall off the descriptor and land zeros — NO vector instruction and NO scalar walk per piece;
//     34 window rows instead of 3 x 32: 2.75 DMA instructions per 64 MFMAs (kernel-row form: 5.3) with K steps of 32 rows;
//   * what the flat window gets wrong is the padding: (row m, tap kw = 0) at ox = 0 and (m, kw = 2) at ox = Wo - 1 read the
//     neighbouring image row's pixel, rows whose iy = oy + kh - 1 leaves the image read another row / frame. These (row, tap)
//     pairs are wave-uniform per k: three 32-bit masks per K step, built on the SCALAR unit from the step's first (oy, ox) by
//     walking its at most 33 / Wo + 2 image-row segments, select the B fragment or zero — one v_cndmask with a scalar lane mask per
//     tap and K pair (3 vector instructions per 12 MFMAs).
// Everything else (LDS image, MFMA order inside a K pair, split-K partial slabs, XCD-aware block order) is wgrad_glds_kernel's.
// Summation order over m is unchanged (K steps of 32 instead of 16 rows do not reorder a split's rows): results are bit-identical
// to the kernel-row form.
#include "common.h"
#include "conv_dev.h"

namespace r3m {

namespace {

// the B fragment of (K pair kk, tap) or zero: lanes 0-31 hold k row 2 kk, lanes 32-63 row 2 kk + 1; bit r of `mask` = row r of
// the K step is padding for this tap
template <int KK>
__device__ __forceinline__ float wgw_select(float b, unsigned mask) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lo = (mask >> (2 * KK)) & 1u ? 0u : 0xFFFFFFFFu;
  const unsigned hi = (mask >> (2 * KK + 1)) & 1u ? 0u : 0xFFFFFFFFu;
  const unsigned long long lanes = ((unsigned long long)hi << 32) | lo;
  // (the compiler's own select, so that its hazard recognizer sees a vector write in front of the MFMA that reads it: an inline-asm
  // v_cndmask directly followed by the MFMA handed it the OLD register contents — first tile of every tap wrong, found on the GPU)
  const float r = __builtin_amdgcn_inverse_ballot_w64(lanes) ? b : 0.f;
  return r;
#else
  return b;
#endif
}

}  // namespace

template <int BT, int BK>
__global__ __launch_bounds__(256, BT == 128 ? 2 : 4) void wgrad_rowwin_kernel(const WgradParams p) {
  static_assert((BT == 128 || BT == 64) && BK == 32, "128 x 128 or 64 x 64 tile, K steps of 32 rows (32-bit padding masks)");
  // 128 x 128: waves 1 x 4, each 128 (co, interleaved: MFMA tile tm owns channels 4 i + tm) x 32 (ci); 64 x 64: waves 2 x 2 of 32 x 32
  constexpr bool WIDE = BT == 128;
  constexpr int TM = WIDE ? 4 : 1;
  constexpr int WR = BK / 4;                   // dY rows a wave stages per K step
  constexpr int RPI = 256 / BT;                // rows one 1 KiB DMA instruction covers
  constexpr int AJ = WR / RPI;                 // dY instructions per wave and stage
  constexpr int XP = (BK + 2 + RPI - 1) / RPI; // X instructions per stage (window of BK + 2 rows), dealt round-robin to the 4 waves
  constexpr int XJ = (XP + 3) / 4;
  constexpr int WIN = XP * RPI;                // window rows allocated
  constexpr int STAGE = (BK + WIN) * BT;       // floats
  constexpr int NP = AJ + XJ;
  static_assert(NP <= BK / 2, "at most one DMA piece per K pair");
  extern __shared__ __attribute__((aligned(128))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WIDE ? 0 : (wave_s >> 1), wn = WIDE ? wave_s : (wave_s & 1);
  const int T = p.KH * p.KW;
  const int lid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int bx = lid % p.gx, by = lid / p.gx;      // by = split index: consecutive logical blocks read the same rows
  // (block coordinates are uniform, but the divisions above are expanded on the vector unit: say so, or everything derived from
  // them — descriptors, the padding-mask walk — stays in vector registers)
  const int kh = __builtin_amdgcn_readfirstlane(bx % p.KH);
  const int tile = bx / p.KH;
  const int tn_ = tile % p.tilesN, tm_ = tile / p.tilesN;
  const int co0 = __builtin_amdgcn_readfirstlane(tm_ * BT), ci0 = __builtin_amdgcn_readfirstlane(tn_ * BT);
  const int ms = __builtin_amdgcn_readfirstlane(by * p.rows_per_split);
  const int me = min(p.M, ms + p.rows_per_split);

  // lane -> (row within the instruction, first channel): loop constants
  const int l_k = lane / (BT / 4), l_c = (lane % (BT / 4)) * 4;

  // dY: descriptor = [row ms + BK step, end of the split) x channels from co0
  const float* a_base = p.dY + (long long)ms * p.Co + co0;   // (scalar: ms, co0 are)
  int a_left = (int)(((long long)(me - ms) * p.Co - co0) * 4);
  const int a_stepb = BK * p.Co * 4;
  unsigned a_voff[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) a_voff[j] = (unsigned)(((wave_s * WR + j * RPI + l_k) * p.Co + l_c) * 4);

  // X window of the step being staged: flat pixel rows xw0 .. xw0 + BK + 1 of the [M][Ci] image (M = N Hi Wi)
  int xw0 = ms + (kh - p.pad) * p.Wi - 1;
  unsigned x_voff[XJ];
#pragma unroll
  for (int q = 0; q < XJ; ++q) x_voff[q] = (unsigned)((((wave_s + 4 * q) * RPI + l_k) * p.Ci + l_c) * 4);

  // descriptor of the window being staged, set with its first piece (32-bit scalar arithmetic: a window touches < 64 rows)
  const float* x_base = p.X;
  int x_bytes = 0;
  unsigned x_neg = 0;
  auto issue_piece = [&](int stage, auto pc_c) __attribute__((always_inline)) {
    constexpr int pc = decltype(pc_c)::value;
    if (R3M_PROBE(p) & 1) return;                       // timing probes (probe builds only; wrong results)
    if constexpr (pc < AJ) {
      buf_dma16_uniform(a_base, a_left, smem + stage * STAGE + (wave_s * WR + pc * RPI) * BT, a_voff[pc]);
    } else {
      constexpr int q = pc - AJ;
      if constexpr (q == 0) {
        const int first = xw0 > 0 ? xw0 : 0;
        int rows = p.M - first;
        rows = rows < 0 ? 0 : (rows > 64 ? 64 : rows);
        x_base = p.X + (long long)first * p.Ci + ci0;
        x_bytes = rows > 0 ? (rows * p.Ci - ci0) * 4 : 0;
        x_neg = (unsigned)((xw0 < 0 ? xw0 : 0) * p.Ci * 4);
      }
      const int i = wave_s + 4 * q;                     // this wave's q-th window piece
      if (i >= XP) return;
      // a window that starts before the tensor (first K steps of frame 0, kh = 0): the descriptor sits at row 0 and the (negative)
      // distance is added to the lane offsets — rows before the tensor wrap to huge offsets, fall off the descriptor and land zeros
      buf_dma16_uniform(x_base, x_bytes, smem + stage * STAGE + BK * BT + i * RPI * BT, x_voff[q] + x_neg);
    }
  };
  auto advance = [&]() __attribute__((always_inline)) {   // both descriptors move on by BK rows (scalar)
    a_base += BK * p.Co;
    a_left = a_left > a_stepb ? a_left - a_stepb : 0;
    xw0 += BK;
  };

  // padding masks of the K step whose first row is (sy, sx): bit r = row r's tap is outside the image
  int sy, sx;
  {
    const int rem = ms % (p.Ho * p.Wo);
    sy = __builtin_amdgcn_readfirstlane(rem / p.Wo);
    sx = __builtin_amdgcn_readfirstlane(rem - sy * p.Wo);
  }
  unsigned mk[3];
  auto step_masks = [&]() __attribute__((always_inline)) {
    unsigned long long starts = sx == 0 ? 1ull : 0ull;  // rows with ox == 0 (bit BK: the row after the step)
    unsigned long long yb = 0;                          // rows whose iy = oy + kh - pad is outside the image
    int seg0 = 0, oy = sy, nxt = p.Wo - sx;             // rows [seg0, nxt) share oy
    while (true) {
      const int end = nxt < BK ? nxt : BK;
      if ((unsigned)(oy + kh - p.pad) >= (unsigned)p.Hi) yb |= ((1ull << end) - 1ull) & ~((1ull << seg0) - 1ull);
      if (nxt > BK) break;
      starts |= 1ull << nxt;
      if (nxt == BK) break;
      seg0 = nxt;
      nxt += p.Wo;
      oy = oy + 1 == p.Ho ? 0 : oy + 1;
    }
    mk[0] = (unsigned)starts | (unsigned)yb;            // kw = 0 reads ix = ox - 1
    mk[1] = (unsigned)yb;
    mk[2] = (unsigned)(starts >> 1) | (unsigned)yb;     // kw = 2 reads ix = ox + 1: padding where the NEXT row starts an image row
    sx += BK;                                           // on to the next step
    while (sx >= p.Wo) {
      sx -= p.Wo;
      sy = sy + 1 == p.Ho ? 0 : sy + 1;
    }
  };

  f32x16 acc[3][TM];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][a][r] = 0.f;

  const int lrow = lane & 31, lh = lane >> 5;
  const float* fragA = smem + lh * BT + (WIDE ? 4 * lrow : wm * 32 + lrow);
  const float* fragB = smem + BK * BT + lh * BT + wn * 32 + lrow;     // window row k + kw of the lane's k = 2 kk + lh
  // MFMAs of one stage; the next K step's DMA pieces go out between them, one per K pair
  auto mfma_stage = [&](const float* fa, const float* fb, int dma_stage) __attribute__((always_inline)) {
    static_for<BK / 2>([&](auto kk_c) __attribute__((always_inline)) {
      constexpr int kk = decltype(kk_c)::value;
      float a[TM];
      if constexpr (WIDE) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(fa + kk * 2 * BT);
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = a4[t];
      } else {
        a[0] = fa[kk * 2 * BT];
      }
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        const float b = wgw_select<kk>(fb[(kk * 2 + tp) * BT], mk[tp]);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) acc[tp][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b, acc[tp][tm], 0, 0, 0);
      }
      if constexpr (kk < NP) {
        if (dma_stage >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(dma_stage, std::integral_constant<int, kk>{});
          if constexpr (kk == NP - 1) advance();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
  };

  const int nk = (me - ms + BK - 1) / BK;
  if (nk > 0) {
    static_for<NP>([&](auto pc_c) __attribute__((always_inline)) { issue_piece(0, pc_c); });
    advance();
  }
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    step_masks();
    mfma_stage(fragA, fragB, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    step_masks();
    mfma_stage(fragA + STAGE, fragB + STAGE, (kt + 2 < nk) ? 0 : -1);
  }
  if (kt < nk) {   // odd tail
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    step_masks();
    mfma_stage(fragA, fragB, -1);
  }

  float* out = p.out + (long long)by * p.Co * T * p.Ci;
#pragma unroll
  for (int tp = 0; tp < 3; ++tp)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = WIDE ? co0 + 4 * rho + tm : co0 + wm * 32 + rho;
        const int ci = ci0 + wn * 32 + lrow;
        out[((long long)co * T + kh * p.KW + tp) * p.Ci + ci] = acc[tp][tm][r];
      }
}

// 3 x 3 "same" convolution (stride 1, pad 1: Ho = Hi, Wo = Wi), fp32, whole tiles
bool wgrad_rowwin_eligible(const WgradParams& p) {
  if (p.dtype != DT_F32 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.Ho != p.Hi || p.Wo != p.Wi) return false;
  if (p.Wo < 2 || p.Ho < 1 || p.M != p.N * p.Ho * p.Wo) return false;
  const bool wide = (p.Co % 128 == 0) && (p.Ci % 128 == 0);
  if (!wide && ((p.Co % 64) || (p.Ci % 64))) return false;
  // 32-bit offsets: a window / a dY tile spans 34 rows; one split's dY < 2 GiB is checked by the caller
  return 40LL * p.Ci * 4 < 0x7FFFF000LL && 40LL * p.Co * 4 < 0x7FFFF000LL;
}

// p: rows_per_split, tilesN, xcd set by launch_wgrad (conv.hip); sets p.gx. One block per (co tile, ci tile, kernel row, split).
int launch_wgrad_rowwin(WgradParams& p, int splitK, hipStream_t s) {
  const bool wide = (p.Co % 128 == 0) && (p.Ci % 128 == 0);
  if (wide) {
    constexpr int LDS = 2 * (32 + 34) * 128 * 4;
    static DynLdsOptIn oi;
    if (int e = ensure_dyn_lds(oi, reinterpret_cast<const void*>(wgrad_rowwin_kernel<128, 32>), LDS, "wgrad_rowwin")) return e;
    p.tilesN = p.Ci / 128;
    p.gx = (p.Co / 128) * p.tilesN * p.KH;
    hipLaunchKernelGGL((wgrad_rowwin_kernel<128, 32>), dim3(p.gx * splitK), dim3(256), LDS, s, p);
  } else {
    constexpr int LDS = 2 * (32 + 36) * 64 * 4;
    static DynLdsOptIn oi;
    if (int e = ensure_dyn_lds(oi, reinterpret_cast<const void*>(wgrad_rowwin_kernel<64, 32>), LDS, "wgrad_rowwin")) return e;
    p.tilesN = p.Ci / 64;
    p.gx = (p.Co / 64) * p.tilesN * p.KH;
    hipLaunchKernelGGL((wgrad_rowwin_kernel<64, 32>), dim3(p.gx * splitK), dim3(256), LDS, s, p);
  }
  return 0;
}

}  // namespace r3m
