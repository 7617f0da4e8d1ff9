// r3m_amd — convolution hot path for gfx950 (MI355X): NHWC fp32 implicit GEMM on the f32-input MFMA
// (v_mfma_f32_32x32x2_f32, exact fp32 == fmaf chain), LDS-staged operand tiles, XCD-aware block order.
//
// Replaces what the reference reaches through torchvision's ResNet -> ATen conv2d / cuDNN
// (call sites /root/reference/r3m/models/models_r3m.py:44-52,99): forward, dgrad and wgrad of every
// 1x1 / 3x3 / 7x7 convolution of ResNet-18/34/50, plus nn.Linear of the language
// reward head (/root/reference/r3m/models/models_language.py:43-51), which is the same GEMM with a bias epilogue.
//
//   * gather_gemm_kernel : out[m, n] = sum_{tap, c} in[pix(m) + off(tap), c] * W[n, tap, c]
//       forward conv (taps = kh,kw; input stride = conv stride), dgrad (taps flipped; stride-2 dgrad is run as
//       4 output-parity classes so no MFMA work is spent on structural zeros), Linear (1 tap).
//       Epilogues: raw store (+ BatchNorm sum / sum-of-squares partials), accumulate, masked residual-gradient add,
//       bias (+ReLU).
//   * wgrad_kernel       : dW[co, tap, ci] = sum_m dY[m, co] * in[pix(m) + off(tap), ci]   (split-K over m)
#include "common.h"
#include <cstdlib>

namespace r3m {

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed; speed only). Remap so that
// each XCD walks a contiguous range of logical tiles: tiles that share an operand panel then share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  const int xcd = bid % NX, idx = bid / NX;
  const int q = nwg / NX, r = nwg % NX;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// =====================================================================================================
// gather-GEMM: block tile BM x BN, K step 32, 4 waves laid out WM x WN, each wave (BM/WM) x (BN/WN) as
// 32x32 MFMA tiles. Operand tiles live in LDS as [row][k] with a 36-float row stride: each lane fetches
// its fragment with one ds_read_b128 (4 consecutive k); lane half h = lane>>5 takes k = 8g+4h..8g+4h+3, so MFMA
// step j of group g contracts k = {8g+j, 8g+4+j}. A and B use the same k permutation, the sum is unchanged.
// The 36-float stride makes both the b128 fragment reads and the b128 staging writes bank-conflict free.
// =====================================================================================================
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void gather_gemm_kernel(const GatherGemmParams p) {
  constexpr int S = 36;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int AJ = BM / 32;  // float4 staging loads per thread (A)
  constexpr int BJ = BN / 32;  // float4 staging loads per thread (B)
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * S];
  float* sA = smem;
  float* sB = smem + BM * S;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gridN = (p.Nc + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / gridN, nt = lid % gridN;  // column tiles of one row panel are neighbours -> same XCD L2
  const int m0 = mt * BM, n0 = nt * BN;

  const int c4 = tid & 7;   // which float4 of the 32-float k slice
  const int r0 = tid >> 3;  // staging row (0..31), + 32*j

  // ---- per-thread row descriptors (fixed for the whole K loop) ----
  long long abase[AJ];
  int aiy[AJ], aix[AJ];
  bool aval[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + r0 + 32 * j;
    aval[j] = m < p.M;
    abase[j] = 0; aiy[j] = 0; aix[j] = 0;
    if (aval[j]) {
      if (p.simple_rows) {
        abase[j] = (long long)m * p.Ci;
      } else {
        const int hw = p.Hg * p.Wg;
        const int n = m / hw;
        const int rem = m - n * hw;
        const int gy = rem / p.Wg;
        const int gx = rem - gy * p.Wg;
        abase[j] = (long long)n * p.Hi * p.Wi * p.Ci;
        aiy[j] = gy * p.is;
        aix[j] = gx * p.is;
      }
    }
  }
  long long bbase[BJ];
  bool bval[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int n = n0 + r0 + 32 * j;
    bval[j] = n < p.Nc;
    bbase[j] = (long long)n * p.T * p.Ci;
  }

  const int kpt = p.Ci >> 5;          // K tiles per tap
  const int nk = p.ntaps * kpt;

  f32x4 ra[AJ], rb[BJ];
  auto load_tile = [&](int kt) {
    const int ti = kt / kpt;
    const int c0 = (kt - ti * kpt) * 32 + c4 * 4;
    const int dy = p.dy[ti], dx = p.dx[ti];
    const long long woff = (long long)p.wt[ti] * p.Ci + c0;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (aval[j]) {
        if (p.simple_rows) {
          v = ldg4(p.A + abase[j] + c0);
        } else {
          const int iy = aiy[j] + dy, ix = aix[j] + dx;
          if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
            v = ldg4(p.A + abase[j] + ((long long)iy * p.Wi + ix) * p.Ci + c0);
        }
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (bval[j]) v = ldg4(p.B + bbase[j] + woff);
      rb[j] = v;
    }
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

  if (nk > 0) load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    // registers -> LDS (single LDS stage; the next tile's global loads fly during the MFMA phase below)
#pragma unroll
    for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4*>(sA + (r0 + 32 * j) * S + c4 * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(sB + (r0 + 32 * j) * S + c4 * 4) = rb[j];
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);
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

  // ---- BatchNorm statistic partials straight from the accumulators ----
  if (EPI & EPI_STATS) {
    // rows >= M were staged as zeros -> their accumulators are exactly 0 and add nothing to either sum
    float* red = smem;  // [WM][2][BN]; the K loop ended with a barrier, the tiles are dead
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[tm][tn][r];
          s += v;
          ss = fmaf(v, v, ss);
        }
      s += __shfl_xor(s, 32);
      ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        const int c = (wn * TN + tn) * 32 + lane;
        red[(wm * 2 + 0) * BN + c] = s;
        red[(wm * 2 + 1) * BN + c] = ss;
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        s += red[(w * 2 + 0) * BN + tid];
        ss += red[(w * 2 + 1) * BN + tid];
      }
      const int col = n0 + tid;
      if (col < p.Nc) {
        p.stats[((long long)mt * 2 + 0) * p.Nc + col] = s;
        p.stats[((long long)mt * 2 + 1) * p.Nc + col] = ss;
      }
    }
    __syncthreads();
  }

  // ---- output: each wave transposes its 32 x (TN*32) accumulator slabs through a private LDS slab so that every store
  // instruction writes 4 rows x 256 contiguous bytes (dwordx4 per lane) instead of 2 rows x 128 B of single dwords ----
  constexpr int CW = TN * 32;          // columns owned by the wave
  constexpr int CS = CW + 4;           // padded slab row stride (floats)
  constexpr int F4 = CW / 4;           // float4 per slab row
  constexpr int RPI = 64 / F4;         // rows covered per store instruction
  static_assert(4 * 32 * CS <= (BM + BN) * S, "epilogue slab must fit in the operand tiles' LDS");
  float* slab = smem + wave * 32 * CS;
  const bool out_simple = (p.os == 1);
  const int hwg = p.Hg * p.Wg;
  const int ecol = (lane % F4) * 4;
  const int erow = lane / F4;
  const int gcol = n0 + wn * CW + ecol;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if ((EPI & EPI_BIAS) && gcol < p.Nc) bias4 = ldg4(p.bias + gcol);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + tn * 32 + lrow] = acc[tm][tn][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int lr = it * RPI + erow;
      const int row = m0 + (wm * TM + tm) * 32 + lr;
      if (row < p.M && gcol < p.Nc) {
        long long roff;
        if (out_simple) {
          roff = (long long)row * p.Nc;
        } else {
          const int n = row / hwg;
          const int rem = row - n * hwg;
          const int gy = rem / p.Wg;
          const int gx = rem - gy * p.Wg;
          roff = (((long long)n * p.Ho + (gy * p.os + p.ooy)) * p.Wo + (gx * p.os + p.oox)) * p.Nc;
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + lr * CS + ecol);
        float* dst = p.out + roff + gcol;
        if (EPI & EPI_BIAS) v += bias4;
        if (EPI & EPI_ACCUM) v += ldg4(dst);
        if (EPI & EPI_MASKED_ADD) {
          const f32x4 g = ldg4(p.add0 + roff + gcol), z = ldg4(p.add1 + roff + gcol);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (z[e] > 0.f) ? g[e] : 0.f;
        }
        if (EPI & EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (EPI & EPI_MASK_OUT) {
          const f32x4 z = ldg4(p.add1 + roff + gcol);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (z[e] > 0.f) ? v[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
    __syncthreads();
  }
}

static inline bool gg_wide(int Nc) { return (Nc % 128) == 0; }

int gather_gemm_grid_m(int M, int Nc) { return gg_wide(Nc) ? ceil_div(M, 128) : ceil_div(M, 256); }

template <int BM, int BN, int WM, int WN>
static int gg_dispatch_epi(const GatherGemmParams& p, int grid, hipStream_t s) {
  const int f = p.flags;
#define GG_CASE(E)                                                                                                   \
  case E:                                                                                                            \
    hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, WM, WN, E>), dim3(grid), dim3(256), 0, s, p);                \
    return 0;
  switch (f) {
    GG_CASE(0)
    GG_CASE(EPI_STATS)
    GG_CASE(EPI_ACCUM)
    GG_CASE(EPI_MASKED_ADD)
    GG_CASE(EPI_BIAS)
    GG_CASE(EPI_RELU)
    GG_CASE(EPI_BIAS | EPI_RELU)
    GG_CASE(EPI_MASK_OUT)
    default:
      set_last_error("gather_gemm: unsupported epilogue flag combination %d", f);
      return 1;
  }
#undef GG_CASE
}

int launch_gather_gemm(const GatherGemmParams& p, hipStream_t s) {
  R3M_REQUIRE(p.Ci % 32 == 0, "gather_gemm: Ci=%d must be a multiple of 32", p.Ci);
  R3M_REQUIRE(p.Nc % 4 == 0, "gather_gemm: Nc=%d must be a multiple of 4", p.Nc);
  R3M_REQUIRE(p.ntaps >= 0 && p.ntaps <= MAX_TAPS, "gather_gemm: ntaps=%d", p.ntaps);
  R3M_REQUIRE(p.M > 0 && p.Nc > 0, "gather_gemm: empty problem M=%d Nc=%d", p.M, p.Nc);
  R3M_REQUIRE((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(p.out) & 15) == 0,
              "gather_gemm: operands must be 16-byte aligned");
  // algorithmic FLOPs: the stem arrives as 160-wide patch rows of which 147 are real (7*7*3)
  const double kdim = (double)p.ntaps * (p.Ci == 160 ? 147 : p.Ci);
  const double flops = 2.0 * (double)p.M * (double)p.Nc * kdim;
  int rc;
  if (gg_wide(p.Nc)) {
    const int gm = ceil_div(p.M, 128), gn = ceil_div(p.Nc, 128);
    prof_begin(KC_GEMM_WIDE, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    rc = gg_dispatch_epi<128, 128, 2, 2>(p, gm * gn, s);
  } else {
    const int gm = ceil_div(p.M, 256), gn = ceil_div(p.Nc, 64);
    prof_begin(KC_GEMM_NARROW, flops, p.M, p.Nc, p.Ci, p.ntaps, s);
    rc = gg_dispatch_epi<256, 64, 4, 1>(p, gm * gn, s);
  }
  prof_end(s);
  if (rc) return rc;
  return check_launch("gather_gemm");
}

// =====================================================================================================
// wgrad: dW[co, tap, ci] = sum_m dY[m, co] * X[pix(m) + off(tap), ci].  GEMM M' = Co tile, N' = Ci tile,
// K' = rows m (split over blockIdx.y). Both operands arrive row(m)-major with channels contiguous, which is exactly
// the [k][i] LDS image the 32x32x2 MFMA wants for conflict-free ds_read_b32 fragment reads.
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
  const int tap = blockIdx.x % T;  // the taps of one (co, ci) tile are neighbours: they re-read the same dY rows
  const int tile = blockIdx.x / T;
  const int tn_ = tile % p.tilesN, tm_ = tile / p.tilesN;
  const int co0 = tm_ * BMt, ci0 = tn_ * BNt;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int ms = blockIdx.y * p.rows_per_split;
  const int me = min(p.M, ms + p.rows_per_split);

  const int a_c = (tid % A_F4) * 4, a_r = tid / A_F4;
  const int b_c = (tid % B_F4) * 4, b_r = tid / B_F4;
  const bool a_cv = (co0 + a_c) < p.Co;
  const bool b_cv = (ci0 + b_c) < p.Ci;
  const int hw = p.Ho * p.Wo;

  f32x4 ra[AJ], rb[BJ];
  auto load_tile = [&](int mk) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = mk + a_r + j * A_RPP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < me && a_cv) v = ldg4(p.dY + (long long)m * p.Co + co0 + a_c);
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int m = mk + b_r + j * B_RPP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < me && b_cv) {
        if (p.simple_rows) {
          v = ldg4(p.X + (long long)m * p.Ci + ci0 + b_c);
        } else {
          const int n = m / hw;
          const int rem = m - n * hw;
          const int oy = rem / p.Wo;
          const int ox = rem - oy * p.Wo;
          const int iy = oy * p.stride + kh - p.pad, ix = ox * p.stride + kw - p.pad;
          if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
            v = ldg4(p.X + (((long long)n * p.Hi + iy) * p.Wi + ix) * p.Ci + ci0 + b_c);
        }
      }
      rb[j] = v;
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
#pragma unroll
    for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4*>(sA + (a_r + j * A_RPP) * BMt + a_c) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4*>(sB + (b_r + j * B_RPP) * BNt + b_c) = rb[j];
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

  float* out = p.out + (long long)blockIdx.y * p.Co * T * p.Ci;
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

static inline bool wg_wide(int Co, int Ci) { return (Co % 128 == 0) && (Ci % 128 == 0); }

int wgrad_pick_split(int M, int Co, int Ci, int T) {
  const int bt = wg_wide(Co, Ci) ? 128 : 64;
  const long long tiles = (long long)ceil_div(Co, bt) * ceil_div(Ci, bt) * T;
  long long split = (2048 + tiles - 1) / tiles;
  const long long max_split = (M + 255) / 256;  // at least 8 K-steps per block
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
  const int T = p.KH * p.KW;
  const double flops = 2.0 * (double)p.M * (double)p.Co * (double)T * (p.Ci == 160 ? 147 : p.Ci);
  if (wg_wide(p.Co, p.Ci)) {
    p.tilesN = ceil_div(p.Ci, 128);
    const int tiles = ceil_div(p.Co, 128) * p.tilesN * T;
    prof_begin(KC_WGRAD_WIDE, flops, p.M, p.Co, p.Ci, T, s);
    hipLaunchKernelGGL((wgrad_kernel<128, 128>), dim3(tiles, splitK), dim3(256), 0, s, p);
  } else {
    p.tilesN = ceil_div(p.Ci, 64);
    const int tiles = ceil_div(p.Co, 64) * p.tilesN * T;
    prof_begin(KC_WGRAD_NARROW, flops, p.M, p.Co, p.Ci, T, s);
    hipLaunchKernelGGL((wgrad_kernel<64, 64>), dim3(tiles, splitK), dim3(256), 0, s, p);
  }
  prof_end(s);
  return check_launch("wgrad");
}

// dW[i] (+)= sum_s partial[s][i]   — fixed summation order: deterministic gradients
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                            long long n4, long long n, int splitK, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = accumulate ? *reinterpret_cast<const f32x4*>(dW + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int sidx = 0; sidx < splitK; ++sidx) v += ldg4(partial + sidx * n + i * 4);
  *reinterpret_cast<f32x4*>(dW + i * 4) = v;
}

int launch_wgrad_reduce(const float* partial, float* dW, long long n, int splitK, int accumulate, hipStream_t s) {
  R3M_REQUIRE(n % 4 == 0, "wgrad_reduce: n=%lld must be a multiple of 4", n);
  const long long n4 = n / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(n4, 256)), dim3(256), 0, s, partial, dW, n4, n, splitK, accumulate);
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

int launch_transpose_w(const float* W, float* Wt, int Co, int T, int Ci, hipStream_t s) {
  hipLaunchKernelGGL(transpose_w_kernel, dim3(ceil_div(Ci, 32), ceil_div(Co, 32), T), dim3(256), 0, s, W, Wt, Co, T, Ci);
  return check_launch("transpose_w");
}

// =====================================================================================================
// Stem: the reference feeds [F,3,224,224] fp32 frames in 0..255 (NCHW) and does x/255 -> Normalize -> conv1 7x7/2 p3
// (/root/reference/r3m/models/models_r3m.py:96-99). Zero padding happens AFTER normalisation, so the normalisation
// cannot be folded into the weights. This kernel normalises and lays the 7x7x3 patches out as rows of 160 floats
// (147 used, k = (kh*7 + kw)*3 + c, matching the OHWI weight image), which the gather-GEMM then consumes as a 1x1 conv.
// =====================================================================================================
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ col, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k4 = (int)(idx % 40);
  const long long m = idx / 40;
  const int ox = (int)(m % 112);
  const long long t = m / 112;
  const int oy = (int)(t % 112);
  const long long f = t / 112;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float sd[3] = {0.229f, 0.224f, 0.225f};
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = k4 * 4 + e;
    float val = 0.f;
    if (k < 147) {
      const int tap = k / 3, c = k - tap * 3;
      const int kh = tap / 7, kw = tap - kh * 7;
      const int iy = oy * 2 + kh - 3, ix = ox * 2 + kw - 3;
      if ((unsigned)iy < 224u && (unsigned)ix < 224u) {
        const float px = x[((f * 3 + c) * 224 + iy) * 224 + ix];
        val = (px / 255.0f - mean[c]) / sd[c];
      }
    }
    v[e] = val;
  }
  *reinterpret_cast<f32x4*>(col + idx * 4) = v;
}

int launch_stem_im2col(const float* x_nchw, float* col, int F, hipStream_t s) {
  const long long total = (long long)F * 112 * 112 * 40;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, x_nchw, col, total);
  return check_launch("stem_im2col");
}

__global__ void pack_stem_w_kernel(const float* __restrict__ w147, float* __restrict__ w160) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 160) return;
  const int co = i / 160, k = i - co * 160;
  w160[i] = (k < 147) ? w147[co * 147 + k] : 0.f;
}
int launch_pack_stem_w(const float* w147, float* w160, hipStream_t s) {
  hipLaunchKernelGGL(pack_stem_w_kernel, dim3(40), dim3(256), 0, s, w147, w160);
  return check_launch("pack_stem_w");
}

__global__ void unpack_stem_dw_kernel(const float* __restrict__ dw160, float* __restrict__ dw147, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 147) return;
  const int co = i / 147, k = i - co * 147;
  const float v = dw160[co * 160 + k];
  dw147[i] = accumulate ? dw147[i] + v : v;
}
int launch_unpack_stem_dw(const float* dw160, float* dw147, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(unpack_stem_dw_kernel, dim3(ceil_div(64 * 147, 256)), dim3(256), 0, s, dw160, dw147, accumulate);
  return check_launch("unpack_stem_dw");
}

}  // namespace r3m
