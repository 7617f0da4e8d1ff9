// r3m_amd — BatchNorm2d (train / eval) forward + backward fused with ReLU and the residual add, MaxPool 3x3/2 and the
// global average pool, for NHWC fp32 activations on gfx950. All of these are HBM-bound passes: float4 (16 B / lane)
// coalesced streams, per-channel coefficients re-read from L1/L2, fixed-order reductions (bit-reproducible run to run).
//
// Reference semantics: torchvision ResNet BatchNorm2d(eps=1e-5, momentum=0.1) + ReLU + MaxPool2d(3,2,1) +
// AdaptiveAvgPool2d(1) as instantiated at /root/reference/r3m/models/models_r3m.py:44-52,62-63.
//   train: normalise with the biased batch variance, update running_var with the unbiased one;
//   statistics: the conv epilogue (conv.hip EPI_STATS) leaves fp32 per-row-block sum / sum-of-squares, which are
//   combined here in fp64, so E[y^2] - mean^2 is evaluated without fp32 cancellation.
#include "common.h"
#include "conv_dev.h"
#include <cstdlib>

// Traversal order of the streaming BatchNorm passes. The 256 MiB Infinity Cache still holds the TAIL of the tensor the previous
// kernel streamed; a consumer that walks the rows in the opposite direction hits it first. R3M_BN_REV bit 1: forward apply, bit 2:
// backward reduce, bit 4: backward apply walk from the last block down (compile-time; variants built by tools/build_ab.sh).
#ifndef R3M_BN_REV
#define R3M_BN_REV 0
#endif
#define BN_BID(bit) ((R3M_BN_REV & (bit)) ? (gridDim.x - 1 - blockIdx.x) : blockIdx.x)

namespace r3m {

// activations are float or bf16_t (T); per-channel coefficients, statistics and partial sums are always fp32
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Streamed activation tensors (each byte touched once per pass, GBs apart from its next use) use the non-temporal cache
// policy: +3-7 % on every pass (fp32 backward 5.6 -> 6.0 TB/s, forward+residual 6.0 -> 6.35; tools/bn_bench.py against a
// -DR3M_BN_NT=0 build), ≈0.6 % of the whole step.
#ifndef R3M_BN_NT
#define R3M_BN_NT 1
#endif
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 lds4(const float* p) {
#if R3M_BN_NT
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ f32x4 lds4(const bf16_t* p) {
#if R3M_BN_NT
  const u32x2 raw = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
  return __builtin_convertvector(__builtin_bit_cast(bf16x4, raw), f32x4);
#else
  return ld4t(p);
#endif
}
__device__ __forceinline__ void sts4(float* p, f32x4 v) {
#if R3M_BN_NT
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void sts4(bf16_t* p, f32x4 v) {
#if R3M_BN_NT
  __builtin_nontemporal_store(__builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4)), reinterpret_cast<u32x2*>(p));
#else
  st4t(p, v);
#endif
}

// dispatch a templated kernel launch on the activation storage type
#define DT_DISPATCH(dt, NAME, ...)                                        \
  do {                                                                    \
    if ((dt) == DT_BF16) { typedef bf16_t T; __VA_ARGS__; }               \
    else if ((dt) == DT_F32) { typedef float T; __VA_ARGS__; }            \
    else { set_last_error(NAME ": unknown dtype %d", (int)(dt)); return 1; } \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// partials [rows][2][C] (fp32)  ->  acc [S][2][C] (fp64), S = gridDim.y slices, fixed order inside a slice
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_reduce_kernel(const float* __restrict__ partials, int rows, int C,
                                                               double* __restrict__ acc) {
  __shared__ double red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  double s = 0.0, ss = 0.0;
  if (c < C) {
    for (int r = blockIdx.y * 4 + ty; r < rows; r += 4 * gridDim.y) {
      s += (double)partials[((long long)r * 2 + 0) * C + c];
      ss += (double)partials[((long long)r * 2 + 1) * C + c];
    }
  }
  red[0][ty][tx] = s;
  red[1][ty][tx] = ss;
  __syncthreads();
  if (ty == 0 && c < C) {
    s = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
    ss = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
    acc[((long long)blockIdx.y * 2 + 0) * C + c] = s;
    acc[((long long)blockIdx.y * 2 + 1) * C + c] = ss;
  }
}

// Slices of the partial-row reduce: enough blocks for narrow layers with MANY rows (64 channels at 56 x 56 x 1280 frames: 62 720
// EPI_BNRED rows, one column block — 64 slices left 192 CUs idle and a 245-row serial walk per thread), within the accumulator
// buffer of 64 x 2 x 2048 doubles (S * C <= 131072).
static inline int reduce_slices(int rows, int C) {
  int s = (rows + 15) / 16;
  int cap = 131072 / (C > 0 ? C : 1);
  if (cap > 256) cap = 256;
  if (cap < 64) cap = 64;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return s;
}

// bytes of the fp64 slice accumulator for C channels (the largest slice count reduce_slices can pick for that C)
size_t bn_acc_bytes(int C) {
  int cap = 131072 / (C > 0 ? C : 1);
  if (cap > 256) cap = 256;
  if (cap < 64) cap = 64;
  return (size_t)cap * 2 * C * 8;
}

// acc must hold 131072 * 2 doubles (64 slices x 2 x 2048 channels, or more slices of fewer channels); the slice count is a pure function of the partial-row count (reduce_slices), so
// the finalize launchers below take the same `stat_rows` and recompute it.
int launch_bn_stats_reduce(const float* partials, int rows, int C, double* acc, hipStream_t s) {
  const int S = reduce_slices(rows, C);
  hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3(ceil_div(C, 64), S), dim3(256), 0, s, partials, rows, C, acc);
  return check_launch("bn_stats_reduce");
}

// Sum the S fp64 slices of acc[S][2][C] for one channel. Block = 64 channels x SG slice groups (group g adds slices g, g+SG, ...,
// combined in group order -> deterministic); returns true for the thread that owns the channel's totals. These per-layer
// kernels are pure latency: one thread walking 64 dependent slices made them 19 us each (~4 ms per step), 4 groups 15 us
// (S is up to 256); 16 groups with the loads of four slices in flight: see DESIGN.md §5.
constexpr int SG = 16;
__device__ __forceinline__ bool slice_totals(const double* __restrict__ acc, int S, int C, int* c_out, double* s_out, double* ss_out) {
  __shared__ double red[2][SG][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  double s = 0.0, ss = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int i = g; i < S; i += SG) {
      s += acc[((long long)i * 2 + 0) * C + c];
      ss += acc[((long long)i * 2 + 1) * C + c];
    }
  }
  red[0][g][tx] = s;
  red[1][g][tx] = ss;
  __syncthreads();
  *c_out = c;
  s = red[0][0][tx];
  ss = red[1][0][tx];
#pragma unroll
  for (int k = 1; k < SG; ++k) { s += red[0][k][tx]; ss += red[1][k][tx]; }
  *s_out = s;
  *ss_out = ss;
  return g == 0 && c < C;
}

__global__ __launch_bounds__(64 * SG) void bn_finalize_kernel(const double* __restrict__ acc, int S, double inv_count,
                                                           double unbias, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float momentum, float eps,
                                                           float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                           float* __restrict__ scale_o, float* __restrict__ shift_o, int C) {
  int c;
  double s, ss;
  if (!slice_totals(acc, S, C, &c, &s, &ss)) return;
  const double mean = s * inv_count;
  double var = ss * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean;
  const float varf = (float)var;
  const float invstd = 1.0f / sqrtf(varf + eps);
  const float sc = gamma[c] * invstd;
  mean_o[c] = meanf;
  invstd_o[c] = invstd;
  scale_o[c] = sc;
  shift_o[c] = fmaf(-meanf, sc, beta[c]);
  if (running_mean) {
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(var * unbias);
  }
}

// `stat_rows` = number of partial rows that were reduced (fixes the slice count), `count` = elements per channel.
int launch_bn_finalize_rows(const double* acc, int stat_rows, long long count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* mean,
                            float* invstd, float* scale, float* shift, int C, hipStream_t s) {
  const double inv_count = 1.0 / (double)count;
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * SG), 0, s, acc, reduce_slices(stat_rows, C), inv_count,
                     unbias, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, C);
  return check_launch("bn_finalize");
}

__global__ __launch_bounds__(256) void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ rm, const float* __restrict__ rv,
                                                              float eps, float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                              float* __restrict__ scale_o, float* __restrict__ shift_o, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  mean_o[c] = rm[c];
  invstd_o[c] = invstd;
  scale_o[c] = sc;
  shift_o[c] = fmaf(-rm[c], sc, beta[c]);
}

int launch_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          float eps, float* mean, float* invstd, float* scale, float* shift, int C, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, gamma, beta, running_mean,
                     running_var, eps, mean, invstd, scale, shift, C);
  return check_launch("bn_eval_coeffs");
}

// ---------------------------------------------------------------------------------------------------------
// Z = [relu]( scale*Y + shift  [+ R]  [+ scale2*Y2 + shift2] )      (one read per operand, one write)
//   R      : identity branch (already activated block input)
//   Y2,... : downsample branch raw conv output with its own BatchNorm coefficients
// ---------------------------------------------------------------------------------------------------------
template <int MODE, class T>  // 0: plain, 1: + R, 2: + affine(Y2)
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ Y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const T* __restrict__ R,
                                                          const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                          T* __restrict__ Z, long long n4, int c4mask, int relu,
                                                          unsigned* __restrict__ maskbits, int span) {
  // A block owns `span` consecutive items and walks them 256 at a time; the launcher only picks span > 256 when 256 is a
  // multiple of C/4, so the thread's channels and their coefficients are loop-invariant (loaded once, and the block's
  // accesses stay one contiguous range). n4 is a multiple of 8 when maskbits is used: an 8-lane nibble group is all in or out.
  long long i = (long long)BN_BID(1) * span + threadIdx.x;
  if (i >= n4) return;
  const long long end = min((long long)(BN_BID(1) + 1) * span, n4);
  const int c = ((int)(i & c4mask)) * 4;
  const f32x4 sc = ld4(scale + c), sh = ld4(shift + c);
  f32x4 sc2 = sc, sh2 = sh;
  if (MODE == 2) { sc2 = ld4(scale2 + c); sh2 = ld4(shift2 + c); }
  for (; i < end; i += 256) {
  const f32x4 y = lds4(Y + i * 4);
  f32x4 z;
#pragma unroll
  for (int e = 0; e < 4; ++e) z[e] = fmaf(y[e], sc[e], sh[e]);
  if (MODE == 1) {
    const f32x4 r = lds4(R + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] += r[e];
  } else if (MODE == 2) {
    const f32x4 y2 = lds4(R + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] += fmaf(y2[e], sc2[e], sh2[e]);
  }
  if (relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = fmaxf(z[e], 0.f);
  }
  sts4(Z + i * 4, z);
  if (maskbits) {
    // ReLU mask of the stored activation, 1 bit per element: float4 index i owns nibble (i & 7) of word i >> 3. The
    // backward kernels read this (1/32 of the bytes) instead of re-reading the activation just to test z > 0.
    unsigned v = ((z[0] > 0.f) ? 1u : 0u) | ((z[1] > 0.f) ? 2u : 0u) | ((z[2] > 0.f) ? 4u : 0u) | ((z[3] > 0.f) ? 8u : 0u);
    v <<= 4 * (threadIdx.x & 7);
    v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4);
    if ((threadIdx.x & 7) == 0) maskbits[i >> 3] = v;
  }
  }
}

// ---- bf16 variants: 8 elements (16 bytes) per lane, same arithmetic per element as the 4-wide kernels ----
struct f32x8 { f32x4 lo, hi; };
__device__ __forceinline__ f32x8 ld8(const bf16_t* p) {
#if R3M_BN_NT
  const bf16x8 v = __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
#else
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#endif
  f32x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r.lo[e] = (float)v[e]; r.hi[e] = (float)v[4 + e]; }
  return r;
}
__device__ __forceinline__ void st8(bf16_t* p, const f32x8& v) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)v.lo[e]; o[4 + e] = (bf16_t)v.hi[e]; }
#if R3M_BN_NT
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(p));
#else
  *reinterpret_cast<bf16x8*>(p) = o;
#endif
}
__device__ __forceinline__ f32x8 ld8f(const float* p) { f32x8 r; r.lo = ld4(p); r.hi = ld4(p + 4); return r; }
#define FOR8(v, expr_lo, expr_hi) _Pragma("unroll") for (int e = 0; e < 4; ++e) { v.lo[e] = (expr_lo); v.hi[e] = (expr_hi); }

template <int MODE>
__global__ __launch_bounds__(256) void bn_act_fwd16_kernel(const bf16_t* __restrict__ Y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const bf16_t* __restrict__ R,
                                                            const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                            bf16_t* __restrict__ Z, long long n8, int c8mask, int relu,
                                                            unsigned* __restrict__ maskbits, int span) {
  // A block owns `span` consecutive items, 256 at a time: 256 is a multiple of C/8, so a thread keeps ITS 8 channels and loads
  // the per-channel coefficients once — at 16 B of payload per load the coefficient vectors would otherwise be most of the
  // L1 traffic. n8 is a multiple of 4 when maskbits is used: a 4-lane word group is all in or all out.
  long long i = (long long)BN_BID(1) * span + threadIdx.x;
  if (i >= n8) return;
  const long long end = min((long long)(BN_BID(1) + 1) * span, n8);
  const int c = ((int)(i & c8mask)) * 8;
  const f32x8 sc = ld8f(scale + c), sh = ld8f(shift + c);
  f32x8 sc2, sh2;
  if (MODE == 2) { sc2 = ld8f(scale2 + c); sh2 = ld8f(shift2 + c); }
  for (; i < end; i += 256) {
  const f32x8 y = ld8(Y + i * 8);
  f32x8 z;
  FOR8(z, fmaf(y.lo[e], sc.lo[e], sh.lo[e]), fmaf(y.hi[e], sc.hi[e], sh.hi[e]))
  if (MODE == 1) {
    const f32x8 r = ld8(R + i * 8);
    FOR8(z, z.lo[e] + r.lo[e], z.hi[e] + r.hi[e])
  } else if (MODE == 2) {
    const f32x8 y2 = ld8(R + i * 8);
    FOR8(z, z.lo[e] + fmaf(y2.lo[e], sc2.lo[e], sh2.lo[e]), z.hi[e] + fmaf(y2.hi[e], sc2.hi[e], sh2.hi[e]))
  }
  if (relu) { FOR8(z, fmaxf(z.lo[e], 0.f), fmaxf(z.hi[e], 0.f)) }
  st8(Z + i * 8, z);
  if (maskbits) {
    // same bit layout as the 4-wide kernel (element j of the tensor = bit j & 31 of word j >> 5): this lane owns byte i & 3
    unsigned v = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) v |= ((z.lo[e] > 0.f) ? (1u << e) : 0u) | ((z.hi[e] > 0.f) ? (16u << e) : 0u);
    v <<= 8 * (threadIdx.x & 3);
    v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2);
    if ((threadIdx.x & 3) == 0) maskbits[i >> 2] = v;
  }
  }
}

static inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// Items per block of the elementwise passes (span = 256 * items consecutive items per block, walked 256 at a time).
// More than one item per thread needs 256 % cv == 0 (cv = C/4 or C/8 channel vectors per row) so that the thread's channels
// are loop-invariant. Measured (tools/bn_bench.py): the forward passes are fastest with one item per thread (6.0-6.3 TB/s
// against 5.5-5.9 with four); the backward apply pass, which carries six coefficient vectors per thread, gains 15-20 % from
// four items per thread in bf16, and a few % for C >= 512 in fp32. R3M_BN_ITEMS overrides.
static inline int bn_span(int cv, int items) {
  const int force = R3M_ENV_INT("R3M_BN_ITEMS", 0);
  if (force > 0) items = force;
  if (cv > 256 || 256 % cv != 0) items = 1;
  return 256 * items;
}

int launch_bn_act_fwd(const void* Yv, const float* scale, const float* shift, const void* Rv, const float* scale2,
                      const float* shift2, void* Zv, long long rows, int C, int relu, unsigned* maskbits, int dt, hipStream_t s) {
  R3M_REQUIRE(is_pow2(C) && C >= 4, "bn_act_fwd: C=%d must be a power of two >= 4", C);
  R3M_REQUIRE(!maskbits || (rows * C / 4) % 8 == 0, "bn_act_fwd: bit mask needs rows*C to be a multiple of 32");
  const long long n4 = rows * C / 4;
  const int c4mask = C / 4 - 1;
  const int span = bn_span(C / 4, 1);
  const int grid = ceil_div(n4, span);
  if (dt == DT_BF16 && C >= 8) {
    const bf16_t* Y = static_cast<const bf16_t*>(Yv);
    const bf16_t* R = static_cast<const bf16_t*>(Rv);
    bf16_t* Z = static_cast<bf16_t*>(Zv);
    const long long n8 = rows * C / 8;
    const int span8 = bn_span(C / 8, C >= 2048 ? 4 : 1);
    const int g8 = ceil_div(n8, span8), c8mask = C / 8 - 1;
    if (R && scale2)
      hipLaunchKernelGGL((bn_act_fwd16_kernel<2>), dim3(g8), dim3(256), 0, s, Y, scale, shift, R, scale2, shift2, Z, n8, c8mask, relu, maskbits, span8);
    else if (R)
      hipLaunchKernelGGL((bn_act_fwd16_kernel<1>), dim3(g8), dim3(256), 0, s, Y, scale, shift, R, scale2, shift2, Z, n8, c8mask, relu, maskbits, span8);
    else
      hipLaunchKernelGGL((bn_act_fwd16_kernel<0>), dim3(g8), dim3(256), 0, s, Y, scale, shift, R, scale2, shift2, Z, n8, c8mask, relu, maskbits, span8);
    return check_launch("bn_act_fwd16");
  }
  DT_DISPATCH(dt, "bn_act_fwd", {
    const T* Y = static_cast<const T*>(Yv);
    const T* R = static_cast<const T*>(Rv);
    T* Z = static_cast<T*>(Zv);
    if (R && scale2)
      hipLaunchKernelGGL((bn_act_fwd_kernel<2, T>), dim3(grid), dim3(256), 0, s, Y, scale, shift, R, scale2, shift2, Z, n4, c4mask, relu, maskbits, span);
    else if (R)
      hipLaunchKernelGGL((bn_act_fwd_kernel<1, T>), dim3(grid), dim3(256), 0, s, Y, scale, shift, R, scale2, shift2, Z, n4, c4mask, relu, maskbits, span);
    else
      hipLaunchKernelGGL((bn_act_fwd_kernel<0, T>), dim3(grid), dim3(256), 0, s, Y, scale, shift, R, scale2, shift2, Z, n4, c4mask, relu, maskbits, span);
  });
  return check_launch("bn_act_fwd");
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm backward, pass 1:  per-channel  sum(g)  and  sum(g * yhat),  g = dZ * [z > 0],  yhat = (y-mean)*invstd.
// The ReLU mask is either read from the saved activation (Zmask: residual blocks, where z also depends on the
// identity branch) or recomputed from y with the very same fmaf the forward used (no extra read).
// Work split: a block owns RB consecutive rows x up to 1024 channels; each thread keeps 4 channels in registers.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mask_nibble(const unsigned* bits, long long elem_off) {
  const long long i4 = elem_off >> 2;
  return (bits[i4 >> 3] >> (4 * (int)(i4 & 7))) & 15u;
}

template <class T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dZ, const T* __restrict__ Zmask,
                                                             const unsigned* __restrict__ Zbits, const T* __restrict__ Y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ partials,
                                                             long long rows, int C, int cpb4, int rows_per_block) {
  __shared__ f32x4 red[2][256];
  const int tcol = threadIdx.x % cpb4, trow = threadIdx.x / cpb4;
  const int rpp = 256 / cpb4;
  const int c = (blockIdx.y * cpb4 + tcol) * 4;
  const long long r_begin = (long long)BN_BID(2) * rows_per_block;
  long long r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  const f32x4 sc = ld4(scale + c), sh = ld4(shift + c), mu = ld4(mean + c), is = ld4(invstd + c);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  for (long long r = r_begin + trow; r < r_end; r += rpp) {
    const long long off = r * C + c;
    const f32x4 y = lds4(Y + off);
    const f32x4 dz = lds4(dZ + off);
    f32x4 g;
    if (Zbits) {
      const unsigned nb = mask_nibble(Zbits, off);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = ((nb >> e) & 1u) ? dz[e] : 0.f;
    } else if (Zmask) {
      const f32x4 z = ld4t(Zmask + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = z[e] > 0.f ? dz[e] : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = fmaf(y[e], sc[e], sh[e]) > 0.f ? dz[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[e] += g[e];
      s2[e] = fmaf(g[e], (y[e] - mu[e]) * is[e], s2[e]);
    }
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (trow == 0) {
    for (int k = 1; k < rpp; ++k) {
      s1 += red[0][k * cpb4 + tcol];
      s2 += red[1][k * cpb4 + tcol];
    }
    st4(partials + ((long long)BN_BID(2) * 2 + 0) * C + c, s1);
    st4(partials + ((long long)BN_BID(2) * 2 + 1) * C + c, s2);
  }
}

// bf16: a thread keeps 8 channels; block = RB rows x up to 2048 channels
__global__ __launch_bounds__(256) void bn_bwd_reduce16_kernel(const bf16_t* __restrict__ dZ, const unsigned* __restrict__ Zbits,
                                                               const bf16_t* __restrict__ Y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, float* __restrict__ partials,
                                                               long long rows, int C, int cpb8, int rows_per_block) {
  __shared__ f32x4 red[4][256];
  const int tcol = threadIdx.x % cpb8, trow = threadIdx.x / cpb8;
  const int rpp = 256 / cpb8;
  const int c = (blockIdx.y * cpb8 + tcol) * 8;
  const long long r_begin = (long long)BN_BID(2) * rows_per_block;
  long long r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  const f32x8 sc = ld8f(scale + c), sh = ld8f(shift + c), mu = ld8f(mean + c), is = ld8f(invstd + c);
  f32x8 s1, s2;
  FOR8(s1, 0.f, 0.f)
  FOR8(s2, 0.f, 0.f)
  for (long long r = r_begin + trow; r < r_end; r += rpp) {
    const long long off = r * C + c;
    const f32x8 y = ld8(Y + off);
    const f32x8 dz = ld8(dZ + off);
    f32x8 g;
    if (Zbits) {
      const unsigned nb = (Zbits[off >> 5] >> (int)(off & 31)) & 255u;
      FOR8(g, ((nb >> e) & 1u) ? dz.lo[e] : 0.f, ((nb >> (4 + e)) & 1u) ? dz.hi[e] : 0.f)
    } else {
      FOR8(g, fmaf(y.lo[e], sc.lo[e], sh.lo[e]) > 0.f ? dz.lo[e] : 0.f, fmaf(y.hi[e], sc.hi[e], sh.hi[e]) > 0.f ? dz.hi[e] : 0.f)
    }
    FOR8(s1, s1.lo[e] + g.lo[e], s1.hi[e] + g.hi[e])
    FOR8(s2, fmaf(g.lo[e], (y.lo[e] - mu.lo[e]) * is.lo[e], s2.lo[e]), fmaf(g.hi[e], (y.hi[e] - mu.hi[e]) * is.hi[e], s2.hi[e]))
  }
  red[0][threadIdx.x] = s1.lo; red[1][threadIdx.x] = s1.hi;
  red[2][threadIdx.x] = s2.lo; red[3][threadIdx.x] = s2.hi;
  __syncthreads();
  if (trow == 0) {
    for (int k = 1; k < rpp; ++k) {
      s1.lo += red[0][k * cpb8 + tcol]; s1.hi += red[1][k * cpb8 + tcol];
      s2.lo += red[2][k * cpb8 + tcol]; s2.hi += red[3][k * cpb8 + tcol];
    }
    float* p1 = partials + ((long long)BN_BID(2) * 2 + 0) * C + c;
    float* p2 = partials + ((long long)BN_BID(2) * 2 + 1) * C + c;
    st4(p1, s1.lo); st4(p1 + 4, s1.hi);
    st4(p2, s2.lo); st4(p2 + 4, s2.hi);
  }
}

static inline bool use_v8(int dt, int C) { return dt == DT_BF16 && C % 8 == 0; }

static inline void bwd_geometry16(long long rows, int C, int* cpb8, int* rpb, int* nblk) {
  int c8 = C / 8;
  *cpb8 = c8 < 256 ? c8 : 256;
  const int rpp = 256 / *cpb8;
  *rpb = 32 * rpp;
  *nblk = ceil_div(rows, *rpb);
}

static inline void bwd_geometry(long long rows, int C, int* cpb4, int* rpb, int* nblk) {
  int c4 = C / 4;
  *cpb4 = c4 < 256 ? c4 : 256;
  const int rpp = 256 / *cpb4;
  *rpb = 32 * rpp;
  *nblk = ceil_div(rows, *rpb);
}

int bn_bwd_partial_rows(long long rows, int C, int dt) {
  int cpb, rpb, nblk;
  if (use_v8(dt, C)) bwd_geometry16(rows, C, &cpb, &rpb, &nblk);
  else bwd_geometry(rows, C, &cpb, &rpb, &nblk);
  return nblk;
}

int launch_bn_bwd_reduce(const void* dZ, const void* Zmask, const unsigned* Zbits, const void* Y, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float* partials, long long rows, int C,
                         int dt, hipStream_t s) {
  R3M_REQUIRE(is_pow2(C) && C >= 4, "bn_bwd_reduce: C=%d must be a power of two >= 4", C);
  int cpb4, rpb, nblk;
  if (use_v8(dt, C) && !Zmask) {
    bwd_geometry16(rows, C, &cpb4, &rpb, &nblk);
    hipLaunchKernelGGL(bn_bwd_reduce16_kernel, dim3(nblk, ceil_div(C / 8, cpb4)), dim3(256), 0, s, static_cast<const bf16_t*>(dZ), Zbits,
                       static_cast<const bf16_t*>(Y), scale, shift, mean, invstd, partials, rows, C, cpb4, rpb);
    return check_launch("bn_bwd_reduce16");
  }
  R3M_REQUIRE(!use_v8(dt, C), "bn_bwd_reduce(bf16): pass the 1-bit mask (zbits) or no mask; a bf16 zmask tensor is not supported");
  bwd_geometry(rows, C, &cpb4, &rpb, &nblk);
  DT_DISPATCH(dt, "bn_bwd_reduce",
              hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(nblk, ceil_div(C / 4, cpb4)), dim3(256), 0, s,
                                 static_cast<const T*>(dZ), static_cast<const T*>(Zmask), Zbits, static_cast<const T*>(Y), scale,
                                 shift, mean, invstd, partials, rows, C, cpb4, rpb));
  return check_launch("bn_bwd_reduce");
}

__global__ __launch_bounds__(64 * SG) void bn_bwd_finalize_kernel(const double* __restrict__ acc, int S, double inv_count,
                                                               int use_batch_stats, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ c1,
                                                               float* __restrict__ c2, int accumulate, int C,
                                                               const float* __restrict__ second_sum_scale) {
  int c;
  double sg, sgy;
  if (!slice_totals(acc, S, C, &c, &sg, &sgy)) return;
  if (second_sum_scale) sgy *= (double)second_sum_scale[c];    // EPI_BNRED partials carry sum(g (y - mean)): x invstd = sum(g yhat)
  const float db = (float)sg, dg = (float)sgy;
  dbeta[c] = accumulate ? dbeta[c] + db : db;
  dgamma[c] = accumulate ? dgamma[c] + dg : dg;
  c1[c] = use_batch_stats ? (float)(sg * inv_count) : 0.f;
  c2[c] = use_batch_stats ? (float)(sgy * inv_count) : 0.f;
}

int launch_bn_bwd_finalize_rows(const double* acc, int stat_rows, long long count, int use_batch_stats, float* dgamma,
                                float* dbeta, float* c1, float* c2, int accumulate, int C, hipStream_t s,
                                const float* second_sum_scale) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64 * SG), 0, s, acc, reduce_slices(stat_rows, C),
                     1.0 / (double)count, use_batch_stats, dgamma, dbeta, c1, c2, accumulate, C, second_sum_scale);
  return check_launch("bn_bwd_finalize");
}

// pass 2:  dY = scale * (g - c1 - yhat * c2)       (c1 = mean(g), c2 = mean(g*yhat); both 0 in eval mode)
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dZ, const T* __restrict__ Zmask,
                                                            const unsigned* __restrict__ Zbits, const T* __restrict__ Y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ c1,
                                                            const float* __restrict__ c2, T* __restrict__ dY,
                                                            long long n4, int c4mask, int span) {
  long long i = (long long)BN_BID(4) * span + threadIdx.x;   // span consecutive items per block, see bn_act_fwd_kernel
  if (i >= n4) return;
  const long long end = min((long long)(BN_BID(4) + 1) * span, n4);
  const int c = ((int)(i & c4mask)) * 4;
  const f32x4 sc = ld4(scale + c), sh = ld4(shift + c), mu = ld4(mean + c), is = ld4(invstd + c);
  const f32x4 k1 = ld4(c1 + c), k2 = ld4(c2 + c);
  for (; i < end; i += 256) {
  const f32x4 y = lds4(Y + i * 4);
  const f32x4 dz = lds4(dZ + i * 4);
  f32x4 g;
  if (Zbits) {
    const unsigned nb = (Zbits[i >> 3] >> (4 * (int)(i & 7))) & 15u;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = ((nb >> e) & 1u) ? dz[e] : 0.f;
  } else if (Zmask) {
    const f32x4 z = ld4t(Zmask + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = z[e] > 0.f ? dz[e] : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = fmaf(y[e], sc[e], sh[e]) > 0.f ? dz[e] : 0.f;
  }
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float yh = (y[e] - mu[e]) * is[e];
    o[e] = sc[e] * (g[e] - k1[e] - yh * k2[e]);
  }
  sts4(dY + i * 4, o);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply16_kernel(const bf16_t* __restrict__ dZ, const unsigned* __restrict__ Zbits,
                                                              const bf16_t* __restrict__ Y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ c1,
                                                              const float* __restrict__ c2, bf16_t* __restrict__ dY, long long n8,
                                                              int c8mask, int span) {
  long long i = (long long)BN_BID(4) * span + threadIdx.x;   // span consecutive items per block, see bn_act_fwd16_kernel
  if (i >= n8) return;
  const long long end = min((long long)(BN_BID(4) + 1) * span, n8);
  const int c = ((int)(i & c8mask)) * 8;
  const f32x8 sc = ld8f(scale + c), sh = ld8f(shift + c), mu = ld8f(mean + c), is = ld8f(invstd + c);
  const f32x8 k1 = ld8f(c1 + c), k2 = ld8f(c2 + c);
  for (; i < end; i += 256) {
  const f32x8 y = ld8(Y + i * 8);
  const f32x8 dz = ld8(dZ + i * 8);
  f32x8 g;
  if (Zbits) {
    const unsigned nb = (Zbits[i >> 2] >> (8 * (int)(i & 3))) & 255u;
    FOR8(g, ((nb >> e) & 1u) ? dz.lo[e] : 0.f, ((nb >> (4 + e)) & 1u) ? dz.hi[e] : 0.f)
  } else {
    FOR8(g, fmaf(y.lo[e], sc.lo[e], sh.lo[e]) > 0.f ? dz.lo[e] : 0.f, fmaf(y.hi[e], sc.hi[e], sh.hi[e]) > 0.f ? dz.hi[e] : 0.f)
  }
  f32x8 o;
  FOR8(o, sc.lo[e] * (g.lo[e] - k1.lo[e] - ((y.lo[e] - mu.lo[e]) * is.lo[e]) * k2.lo[e]),
       sc.hi[e] * (g.hi[e] - k1.hi[e] - ((y.hi[e] - mu.hi[e]) * is.hi[e]) * k2.hi[e]))
  st8(dY + i * 8, o);
  }
}

int launch_bn_bwd_apply(const void* dZ, const void* Zmask, const unsigned* Zbits, const void* Y, const float* scale,
                        const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, void* dY,
                        long long rows, int C, int dt, hipStream_t s) {
  R3M_REQUIRE(is_pow2(C) && C >= 4, "bn_bwd_apply: C=%d must be a power of two >= 4", C);
  if (use_v8(dt, C) && !Zmask) {
    const long long n8 = rows * C / 8;
    const int span8 = bn_span(C / 8, 4);
    hipLaunchKernelGGL(bn_bwd_apply16_kernel, dim3(ceil_div(n8, span8)), dim3(256), 0, s, static_cast<const bf16_t*>(dZ), Zbits,
                       static_cast<const bf16_t*>(Y), scale, shift, mean, invstd, c1, c2, static_cast<bf16_t*>(dY), n8, C / 8 - 1, span8);
    return check_launch("bn_bwd_apply16");
  }
  const long long n4 = rows * C / 4;
  const int span = bn_span(C / 4, C >= 512 ? 4 : 1);
  DT_DISPATCH(dt, "bn_bwd_apply",
              hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ceil_div(n4, span)), dim3(256), 0, s, static_cast<const T*>(dZ),
                                 static_cast<const T*>(Zmask), Zbits, static_cast<const T*>(Y), scale, shift, mean, invstd, c1, c2,
                                 static_cast<T*>(dY), n4, C / 4 - 1, span));
  return check_launch("bn_bwd_apply");
}

// ---------------------------------------------------------------------------------------------------------
// pass 2 for the TWO BatchNorms of a downsample block's tail in one launch (round 5). out = relu(bn3(y3) + bn_d(yd)): both BatchNorms
// see the same masked gradient g = dOut * [out > 0], so the stand-alone passes read dOut and the mask bits twice. Here they are read
// once: dY3 = scale3 (g - c1_3 - yhat3 c2_3), dYd = scale_d (g - c1_d - yhat_d c2_d). Same arithmetic per element as two launches of
// bn_bwd_apply_kernel (bit-identical results); 20 instead of 24 bytes per element in fp32 (10 / 12 in bf16).
// A / B: {scale, mean, invstd, c1, c2} of the two BatchNorms (the mask comes as bits: block outputs always have them).
// ---------------------------------------------------------------------------------------------------------
struct BnApplyCoef { const float* scale; const float* mean; const float* invstd; const float* c1; const float* c2; };

template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const T* __restrict__ dZ, const unsigned* __restrict__ Zbits,
                                                             const T* __restrict__ YA, BnApplyCoef A, T* __restrict__ dYA,
                                                             const T* __restrict__ YB, BnApplyCoef B, T* __restrict__ dYB,
                                                             long long n4, int c4mask, int span) {
  long long i = (long long)BN_BID(4) * span + threadIdx.x;
  if (i >= n4) return;
  const long long end = min((long long)(BN_BID(4) + 1) * span, n4);
  const int c = ((int)(i & c4mask)) * 4;
  const f32x4 scA = ld4(A.scale + c), muA = ld4(A.mean + c), isA = ld4(A.invstd + c), k1A = ld4(A.c1 + c), k2A = ld4(A.c2 + c);
  const f32x4 scB = ld4(B.scale + c), muB = ld4(B.mean + c), isB = ld4(B.invstd + c), k1B = ld4(B.c1 + c), k2B = ld4(B.c2 + c);
  for (; i < end; i += 256) {
    const f32x4 ya = lds4(YA + i * 4);
    const f32x4 yb = lds4(YB + i * 4);
    const f32x4 dz = lds4(dZ + i * 4);
    const unsigned nb = (Zbits[i >> 3] >> (4 * (int)(i & 7))) & 15u;
    f32x4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = ((nb >> e) & 1u) ? dz[e] : 0.f;
      const float yha = (ya[e] - muA[e]) * isA[e];
      const float yhb = (yb[e] - muB[e]) * isB[e];
      oa[e] = scA[e] * (g - k1A[e] - yha * k2A[e]);
      ob[e] = scB[e] * (g - k1B[e] - yhb * k2B[e]);
    }
    sts4(dYA + i * 4, oa);
    sts4(dYB + i * 4, ob);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply2_16_kernel(const bf16_t* __restrict__ dZ, const unsigned* __restrict__ Zbits,
                                                                const bf16_t* __restrict__ YA, BnApplyCoef A, bf16_t* __restrict__ dYA,
                                                                const bf16_t* __restrict__ YB, BnApplyCoef B, bf16_t* __restrict__ dYB,
                                                                long long n8, int c8mask, int span) {
  long long i = (long long)BN_BID(4) * span + threadIdx.x;
  if (i >= n8) return;
  const long long end = min((long long)(BN_BID(4) + 1) * span, n8);
  const int c = ((int)(i & c8mask)) * 8;
  const f32x8 scA = ld8f(A.scale + c), muA = ld8f(A.mean + c), isA = ld8f(A.invstd + c), k1A = ld8f(A.c1 + c), k2A = ld8f(A.c2 + c);
  const f32x8 scB = ld8f(B.scale + c), muB = ld8f(B.mean + c), isB = ld8f(B.invstd + c), k1B = ld8f(B.c1 + c), k2B = ld8f(B.c2 + c);
  for (; i < end; i += 256) {
    const f32x8 ya = ld8(YA + i * 8);
    const f32x8 yb = ld8(YB + i * 8);
    const f32x8 dz = ld8(dZ + i * 8);
    const unsigned nb = (Zbits[i >> 2] >> (8 * (int)(i & 3))) & 255u;
    f32x8 g, oa, ob;
    FOR8(g, ((nb >> e) & 1u) ? dz.lo[e] : 0.f, ((nb >> (4 + e)) & 1u) ? dz.hi[e] : 0.f)
    FOR8(oa, scA.lo[e] * (g.lo[e] - k1A.lo[e] - ((ya.lo[e] - muA.lo[e]) * isA.lo[e]) * k2A.lo[e]),
         scA.hi[e] * (g.hi[e] - k1A.hi[e] - ((ya.hi[e] - muA.hi[e]) * isA.hi[e]) * k2A.hi[e]))
    FOR8(ob, scB.lo[e] * (g.lo[e] - k1B.lo[e] - ((yb.lo[e] - muB.lo[e]) * isB.lo[e]) * k2B.lo[e]),
         scB.hi[e] * (g.hi[e] - k1B.hi[e] - ((yb.hi[e] - muB.hi[e]) * isB.hi[e]) * k2B.hi[e]))
    st8(dYA + i * 8, oa);
    st8(dYB + i * 8, ob);
  }
}

// pass 1 for the same pair (bf16 plans, where every first pass is stand-alone): sum(g) is common, sum(g yhat) per BatchNorm; dOut and the
// mask bits are read once. Two partial sets [rows][2][C], the second `set_stride` floats behind the first; same block geometry and
// summation order as bn_bwd_reduce16_kernel (bit-identical partials).
__global__ __launch_bounds__(256) void bn_bwd_reduce2_16_kernel(const bf16_t* __restrict__ dZ, const unsigned* __restrict__ Zbits,
                                                                 const bf16_t* __restrict__ YA, const float* __restrict__ meanA,
                                                                 const float* __restrict__ invstdA, const bf16_t* __restrict__ YB,
                                                                 const float* __restrict__ meanB, const float* __restrict__ invstdB,
                                                                 float* __restrict__ partials, long long set_stride, long long rows, int C,
                                                                 int cpb8, int rows_per_block) {
  __shared__ f32x4 red[6][256];
  const int tcol = threadIdx.x % cpb8, trow = threadIdx.x / cpb8;
  const int rpp = 256 / cpb8;
  const int c = (blockIdx.y * cpb8 + tcol) * 8;
  const long long r_begin = (long long)BN_BID(2) * rows_per_block;
  long long r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  const f32x8 muA = ld8f(meanA + c), isA = ld8f(invstdA + c), muB = ld8f(meanB + c), isB = ld8f(invstdB + c);
  f32x8 s1, s2a, s2b;
  FOR8(s1, 0.f, 0.f)
  FOR8(s2a, 0.f, 0.f)
  FOR8(s2b, 0.f, 0.f)
  for (long long r = r_begin + trow; r < r_end; r += rpp) {
    const long long off = r * C + c;
    const f32x8 ya = ld8(YA + off);
    const f32x8 yb = ld8(YB + off);
    const f32x8 dz = ld8(dZ + off);
    const unsigned nb = (Zbits[off >> 5] >> (int)(off & 31)) & 255u;
    f32x8 g;
    FOR8(g, ((nb >> e) & 1u) ? dz.lo[e] : 0.f, ((nb >> (4 + e)) & 1u) ? dz.hi[e] : 0.f)
    FOR8(s1, s1.lo[e] + g.lo[e], s1.hi[e] + g.hi[e])
    FOR8(s2a, fmaf(g.lo[e], (ya.lo[e] - muA.lo[e]) * isA.lo[e], s2a.lo[e]), fmaf(g.hi[e], (ya.hi[e] - muA.hi[e]) * isA.hi[e], s2a.hi[e]))
    FOR8(s2b, fmaf(g.lo[e], (yb.lo[e] - muB.lo[e]) * isB.lo[e], s2b.lo[e]), fmaf(g.hi[e], (yb.hi[e] - muB.hi[e]) * isB.hi[e], s2b.hi[e]))
  }
  red[0][threadIdx.x] = s1.lo; red[1][threadIdx.x] = s1.hi;
  red[2][threadIdx.x] = s2a.lo; red[3][threadIdx.x] = s2a.hi;
  red[4][threadIdx.x] = s2b.lo; red[5][threadIdx.x] = s2b.hi;
  __syncthreads();
  if (trow == 0) {
    for (int k = 1; k < rpp; ++k) {
      s1.lo += red[0][k * cpb8 + tcol]; s1.hi += red[1][k * cpb8 + tcol];
      s2a.lo += red[2][k * cpb8 + tcol]; s2a.hi += red[3][k * cpb8 + tcol];
      s2b.lo += red[4][k * cpb8 + tcol]; s2b.hi += red[5][k * cpb8 + tcol];
    }
    float* p1 = partials + ((long long)BN_BID(2) * 2 + 0) * C + c;
    float* p2 = partials + ((long long)BN_BID(2) * 2 + 1) * C + c;
    st4(p1, s1.lo); st4(p1 + 4, s1.hi);
    st4(p2, s2a.lo); st4(p2 + 4, s2a.hi);
    st4(p1 + set_stride, s1.lo); st4(p1 + set_stride + 4, s1.hi);
    st4(p2 + set_stride, s2b.lo); st4(p2 + set_stride + 4, s2b.hi);
  }
}

// true: launched (bf16 plans with C a multiple of 8); false: the caller runs two stand-alone first passes
bool bn_bwd_reduce2_available(int C, int dt) { return use_v8(dt, C); }
int launch_bn_bwd_reduce2(const void* dZ, const unsigned* Zbits, const void* YA, const float* coefA, const void* YB, const float* coefB,
                          float* partials, long long set_stride, long long rows, int C, int dt, hipStream_t s) {
  R3M_REQUIRE(is_pow2(C) && use_v8(dt, C) && Zbits, "bn_bwd_reduce2: bf16 plans with mask bits only (C=%d dtype=%d)", C, dt);
  int cpb8, rpb, nblk;
  bwd_geometry16(rows, C, &cpb8, &rpb, &nblk);
  R3M_REQUIRE((long long)nblk * 2 * C <= set_stride, "bn_bwd_reduce2: partial sets overlap");
  hipLaunchKernelGGL(bn_bwd_reduce2_16_kernel, dim3(nblk, ceil_div(C / 8, cpb8)), dim3(256), 0, s, static_cast<const bf16_t*>(dZ), Zbits,
                     static_cast<const bf16_t*>(YA), coefA, coefA + C, static_cast<const bf16_t*>(YB), coefB, coefB + C, partials, set_stride,
                     rows, C, cpb8, rpb);
  return check_launch("bn_bwd_reduce2_16");
}

// coefA / coefB: the layer's coefficient block [6][C] = {mean, invstd, scale, shift, c1, c2} (engine.hip coef())
int launch_bn_bwd_apply2(const void* dZ, const unsigned* Zbits, const void* YA, const float* coefA, void* dYA, const void* YB,
                         const float* coefB, void* dYB, long long rows, int C, int dt, hipStream_t s) {
  R3M_REQUIRE(is_pow2(C) && C >= 8 && Zbits, "bn_bwd_apply2: C=%d must be a power of two >= 8 and the mask must come as bits", C);
  const BnApplyCoef A{coefA + 2LL * C, coefA, coefA + C, coefA + 4LL * C, coefA + 5LL * C};
  const BnApplyCoef B{coefB + 2LL * C, coefB, coefB + C, coefB + 4LL * C, coefB + 5LL * C};
  if (use_v8(dt, C)) {
    const long long n8 = rows * C / 8;
    const int span8 = bn_span(C / 8, 4);
    hipLaunchKernelGGL(bn_bwd_apply2_16_kernel, dim3(ceil_div(n8, span8)), dim3(256), 0, s, static_cast<const bf16_t*>(dZ), Zbits,
                       static_cast<const bf16_t*>(YA), A, static_cast<bf16_t*>(dYA), static_cast<const bf16_t*>(YB), B,
                       static_cast<bf16_t*>(dYB), n8, C / 8 - 1, span8);
    return check_launch("bn_bwd_apply2_16");
  }
  R3M_REQUIRE(dt == DT_F32, "bn_bwd_apply2: dtype %d", dt);
  const long long n4 = rows * C / 4;
  const int span = bn_span(C / 4, C >= 512 ? 4 : 1);
  hipLaunchKernelGGL((bn_bwd_apply2_kernel<float>), dim3(ceil_div(n4, span)), dim3(256), 0, s, static_cast<const float*>(dZ), Zbits,
                     static_cast<const float*>(YA), A, static_cast<float*>(dYA), static_cast<const float*>(YB), B,
                     static_cast<float*>(dYB), n4, C / 4 - 1, span);
  return check_launch("bn_bwd_apply2");
}

// ---------------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1), NHWC. Forward keeps the window-local argmax (0..8, first maximum in
// row-major scan order, like ATen) in one byte per output element; backward is a gather over the <= 4 windows that
// contain an input pixel, so it needs neither atomics nor a zero-fill pass.
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ Z, T* __restrict__ P,
                                                           unsigned char* __restrict__ amax, long long total, int Hi, int Wi,
                                                           int Ho, int Wo, int C4) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  long long t = idx / C4;
  const int px = (int)(t % Wo); t /= Wo;
  const int py = (int)(t % Ho);
  const long long n = t / Ho;
  f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0, 0, 0, 0};
  bool first = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int y = py * 2 - 1 + i;
    if ((unsigned)y >= (unsigned)Hi) continue;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int x = px * 2 - 1 + j;
      if ((unsigned)x >= (unsigned)Wi) continue;
      const f32x4 v = ld4t(Z + (((n * Hi + y) * Wi + x) * C4 + c4) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (first || v[e] > best[e]) { best[e] = v[e]; bi[e] = i * 3 + j; }
      first = false;
    }
  }
  st4t(P + idx * 4, best);
  *reinterpret_cast<uchar4*>(amax + idx * 4) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
}

int launch_maxpool_fwd(const void* Z, void* P, unsigned char* amax, int N, int Hi, int Wi, int C, int dt, hipStream_t s) {
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C / 4);
  DT_DISPATCH(dt, "maxpool_fwd",
              hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const T*>(Z),
                                 static_cast<T*>(P), amax, total, Hi, Wi, Ho, Wo, C / 4));
  return check_launch("maxpool_fwd");
}

template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dP, const unsigned char* __restrict__ amax,
                                                           T* __restrict__ dZ, long long total, int Hi, int Wi, int Ho,
                                                           int Wo, int C4) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  long long t = idx / C4;
  const int x = (int)(t % Wi); t /= Wi;
  const int y = (int)(t % Hi);
  const long long n = t / Hi;
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  // windows py with 2*py-1 <= y <= 2*py+1
  const int py0 = y >> 1, py1 = (y + 1) >> 1;
  const int px0 = x >> 1, px1 = (x + 1) >> 1;
  for (int py = py0; py <= py1; ++py) {
    if (py >= Ho) continue;
    const int i = y - (py * 2 - 1);
    for (int px = px0; px <= px1; ++px) {
      if (px >= Wo) continue;
      const int j = x - (px * 2 - 1);
      const int code = i * 3 + j;
      const long long o = (((n * Ho + py) * Wo + px) * C4 + c4) * 4;
      const uchar4 a = *reinterpret_cast<const uchar4*>(amax + o);
      const f32x4 d = ld4t(dP + o);
      if (a.x == code) g[0] += d[0];
      if (a.y == code) g[1] += d[1];
      if (a.z == code) g[2] += d[2];
      if (a.w == code) g[3] += d[3];
    }
  }
  st4t(dZ + idx * 4, g);
}

int launch_maxpool_bwd(const void* dP, const unsigned char* amax, void* dZ, int N, int Hi, int Wi, int C, int dt, hipStream_t s) {
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  const long long total = (long long)N * Hi * Wi * (C / 4);
  DT_DISPATCH(dt, "maxpool_bwd",
              hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const T*>(dP), amax,
                                 static_cast<T*>(dZ), total, Hi, Wi, Ho, Wo, C / 4));
  return check_launch("maxpool_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// Stem tail fused: BatchNorm + ReLU + MaxPool (forward) and MaxPool-backward + ReLU/BatchNorm-backward (both passes).
// The activated stem output Z0 [F,112,112,64] is the largest tensor of the network (4.1 GB fp32 at 1280 frames); the
// unfused sequence wrote it, re-read it for the pooling, and in backward wrote / twice re-read the equally large dZ0.
// Fused, Z0 and dZ0 never exist: forward reads Y0 and writes the pooled tensor + argmax; the backward passes read Y0 and
// gather dZ0 on the fly from the 4x smaller pooled gradient (L2 hits). Arithmetic per element is unchanged: z =
// relu(fmaf(y, scale, shift)) (rounded to the storage type before the comparison, as the stored Z0 was), first maximum in
// scan order, dz = sum of the pooled gradients whose argmax points here (rounded to the storage type as the stored dZ0 was).
// ---------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ f32x4 round_as(f32x4 v);
template <>
__device__ __forceinline__ f32x4 round_as<float>(f32x4 v) { return v; }
template <>
__device__ __forceinline__ f32x4 round_as<bf16_t>(f32x4 v) { return __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4); }

// Channel vectors per thread: 4 channels (one f32x4) for fp32, 8 channels (two f32x4, one 16-byte load) for bf16 — the bf16
// tensors are half the bytes, so the 4-wide kernels were instruction-bound there (measured 4.1 / 2.5 / 3.6 TB/s against
// 7.2 / 5.4 / 5.5 TB/s for fp32).
template <class T>
struct PoolVec { static constexpr int V4 = 1; };
template <>
struct PoolVec<bf16_t> { static constexpr int V4 = 2; };

template <int V4, class T>
__device__ __forceinline__ void ldv(const T* __restrict__ p, f32x4 (&q)[V4]) {   // cached: the gathered, re-read operands
  if constexpr (V4 == 1) {
    q[0] = ld4t(p);
  } else {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) { q[0][e] = (float)v[e]; q[1][e] = (float)v[4 + e]; }
  }
}
template <int V4, class T>
__device__ __forceinline__ void ldv_stream(const T* __restrict__ p, f32x4 (&q)[V4]) {   // read once
  if constexpr (V4 == 1) {
    q[0] = lds4(p);
  } else {
    const f32x8 v = ld8(p);
    q[0] = v.lo; q[1] = v.hi;
  }
}
template <int V4, class T>
__device__ __forceinline__ void stv(T* __restrict__ p, const f32x4 (&q)[V4], bool stream) {
  if constexpr (V4 == 1) {
    if (stream) sts4(p, q[0]); else st4t(p, q[0]);
  } else {
    if (stream) { f32x8 v; v.lo = q[0]; v.hi = q[1]; st8(p, v); }
    else {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)q[0][e]; o[4 + e] = (bf16_t)q[1][e]; }
      *reinterpret_cast<bf16x8*>(p) = o;
    }
  }
}
template <int V4>
__device__ __forceinline__ void ld_codes(const unsigned char* __restrict__ p, unsigned (&a)[V4]) {   // 4 argmax codes per word
  if constexpr (V4 == 1) a[0] = *reinterpret_cast<const unsigned*>(p);
  else { const uint2 v = *reinterpret_cast<const uint2*>(p); a[0] = v.x; a[1] = v.y; }
}

template <class T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ Y, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, T* __restrict__ P,
                                                                   unsigned char* __restrict__ amax, long long total, int Hi,
                                                                   int Wi, int Ho, int Wo, int CV) {
  constexpr int V4 = PoolVec<T>::V4, V = 4 * V4;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % CV);
  long long t = idx / CV;
  const int px = (int)(t % Wo); t /= Wo;
  const int py = (int)(t % Ho);
  const long long n = t / Ho;
  f32x4 sc[V4], sh[V4], best[V4];
  unsigned bi[V4];
#pragma unroll
  for (int k = 0; k < V4; ++k) {
    sc[k] = ld4(scale + cv * V + 4 * k); sh[k] = ld4(shift + cv * V + 4 * k);
    best[k] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    bi[k] = 0u;
  }
  bool first = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int y = py * 2 - 1 + i;
    if ((unsigned)y >= (unsigned)Hi) continue;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int x = px * 2 - 1 + j;
      if ((unsigned)x >= (unsigned)Wi) continue;
      f32x4 yv[V4];
      ldv<V4>(Y + (((n * Hi + y) * Wi + x) * CV + cv) * V, yv);
#pragma unroll
      for (int k = 0; k < V4; ++k) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(yv[k][e], sc[k][e], sh[k][e]), 0.f);
        v = round_as<T>(v);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (first || v[e] > best[k][e]) { best[k][e] = v[e]; bi[k] = (bi[k] & ~(0xffu << (8 * e))) | ((unsigned)(i * 3 + j) << (8 * e)); }
      }
      first = false;
    }
  }
  stv<V4>(P + idx * V, best, false);
  if constexpr (V4 == 1) *reinterpret_cast<unsigned*>(amax + idx * V) = bi[0];
  else *reinterpret_cast<uint2*>(amax + idx * V) = make_uint2(bi[0], bi[1]);
}

// bf16 form of the fused forward. The activated values are >= 0 and rounded to bf16, so as fp32 bit patterns they order like
// integers and their low 16 bits are free: key = bits(z) | (15 - code) turns "first maximum in scan order" into ONE integer max per
// candidate (largest value, then smallest window code), instead of a compare and two selects. A thread owns (px, 8 channels) and walks
// POOL_SEG consecutive output rows downwards: the horizontal 3-tap maxima of input row 2*py+1 are carried into window py+1 (whose
// row 2*(py+1)-1 it is) with the code's row part re-based by plain subtraction (the low bits never borrow: 15 - 3i - j >= 7), so an
// output costs two new input rows, not three. Measured against the one-output-per-thread kernel above: see DESIGN.md §4.2.
constexpr int POOL_SEG = 8;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Keys8 { int k[8]; };

__device__ __forceinline__ void pool_tap_keys(const bf16_t* __restrict__ p, const f32x8& sc, const f32x8& sh, int kc, Keys8& h) {
  const u32x4 raw = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float ylo = __builtin_bit_cast(float, raw[q] << 16), yhi = __builtin_bit_cast(float, raw[q] & 0xffff0000u);
    const float slo = q < 2 ? sc.lo[2 * q] : sc.hi[2 * q - 4], shi = q < 2 ? sc.lo[2 * q + 1] : sc.hi[2 * q - 3];
    const float tlo = q < 2 ? sh.lo[2 * q] : sh.hi[2 * q - 4], thi = q < 2 ? sh.lo[2 * q + 1] : sh.hi[2 * q - 3];
    f32x2 z;
    z[0] = fmaxf(fmaf(ylo, slo, tlo), 0.f);
    z[1] = fmaxf(fmaf(yhi, shi, thi), 0.f);
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(z, bf16x2));   // one v_cvt_pk_bf16_f32
    const int klo = (int)((u << 16) | (unsigned)kc), khi = (int)((u & 0xffff0000u) | (unsigned)kc);
    h.k[2 * q] = h.k[2 * q] > klo ? h.k[2 * q] : klo;
    h.k[2 * q + 1] = h.k[2 * q + 1] > khi ? h.k[2 * q + 1] : khi;
  }
}

// horizontal maxima (keys with the column part of the code) of input row y at columns 2*px-1 .. 2*px+1
__device__ __forceinline__ void pool_row_keys(const bf16_t* __restrict__ Yrow, int px, int Wi, int CV, int cv, const f32x8& sc,
                                              const f32x8& sh, Keys8& h) {
#pragma unroll
  for (int e = 0; e < 8; ++e) h.k[e] = 0;
  const bf16_t* p = Yrow + ((long long)(2 * px) * CV + cv) * 8;
  if (px > 0) pool_tap_keys(p - (long long)CV * 8, sc, sh, 15, h);
  pool_tap_keys(p, sc, sh, 14, h);
  if (2 * px + 1 < Wi) pool_tap_keys(p + (long long)CV * 8, sc, sh, 13, h);
}

__global__ __launch_bounds__(1024) void bn_relu_maxpool_fwd16_kernel(const bf16_t* __restrict__ Y, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, bf16_t* __restrict__ P,
                                                                     unsigned char* __restrict__ amax, int Hi, int Wi, int Ho, int Wo,
                                                                     int CV, int cv_log2, int segs) {
  const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
  const int py0 = seg * POOL_SEG;
  const int py1 = py0 + POOL_SEG < Ho ? py0 + POOL_SEG : Ho;
  const int items = Wo * CV;
  const bf16_t* Yn = Y + (long long)n * Hi * Wi * CV * 8;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int px = it >> cv_log2, cv = it & (CV - 1);
    const f32x8 sc = ld8f(scale + cv * 8), sh = ld8f(shift + cv * 8);
    Keys8 hp, ha, hb;
    if (py0 > 0) pool_row_keys(Yn + (long long)(2 * py0 - 1) * Wi * CV * 8, px, Wi, CV, cv, sc, sh, hp);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) hp.k[e] = 0;
    }
    for (int py = py0; py < py1; ++py) {
      pool_row_keys(Yn + (long long)(2 * py) * Wi * CV * 8, px, Wi, CV, cv, sc, sh, ha);
      if (2 * py + 1 < Hi) pool_row_keys(Yn + (long long)(2 * py + 1) * Wi * CV * 8, px, Wi, CV, cv, sc, sh, hb);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) hb.k[e] = 0;
      }
      u32x4 val;
      unsigned code[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int b[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int e = 2 * q + r;
          int m = ha.k[e] - 3;                       // window row 1; an absent row (key 0) goes negative and loses
          m = hp.k[e] > m ? hp.k[e] : m;             // window row 0
          const int m2 = hb.k[e] - 6;                // window row 2
          b[r] = m > m2 ? m : m2;
          hp.k[e] = hb.k[e];
        }
        val[q] = ((unsigned)b[0] >> 16) | ((unsigned)b[1] & 0xffff0000u);
        const unsigned c2 = ((unsigned)b[0] & 15u) | (((unsigned)b[1] & 15u) << 8);
        if ((q & 1) == 0) code[q >> 1] = c2; else code[q >> 1] |= c2 << 16;
      }
      const long long o = ((((long long)n * Ho + py) * Wo + px) * CV + cv) * 8;
      *reinterpret_cast<u32x4*>(P + o) = val;
      *reinterpret_cast<uint2*>(amax + o) = make_uint2(0x0f0f0f0fu - code[0], 0x0f0f0f0fu - code[1]);
    }
  }
}

static inline int ilog2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
// threads of a block that owns Wo x CV (pixel, channel-vector) items: all of them when they fit, else 1024 (a multiple of CV <= 256)
static inline int pool_row_threads(int items) { return items >= 1024 ? 1024 : (items + 63) / 64 * 64; }

int launch_bn_relu_maxpool_fwd(const void* Y, const float* scale, const float* shift, void* P, unsigned char* amax, int N, int Hi,
                               int Wi, int C, int dt, hipStream_t s) {
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  R3M_REQUIRE(C % 8 == 0, "bn_relu_maxpool_fwd: C=%d must be a multiple of 8", C);
  if (dt == DT_BF16 && is_pow2(C) && C <= 2048) {
    const int CV = C / 8, segs = ceil_div(Ho, POOL_SEG);
    hipLaunchKernelGGL(bn_relu_maxpool_fwd16_kernel, dim3(N * segs), dim3(pool_row_threads(Wo * CV)), 0, s,
                       static_cast<const bf16_t*>(Y), scale, shift, static_cast<bf16_t*>(P), amax, Hi, Wi, Ho, Wo, CV, ilog2_exact(CV), segs);
    return check_launch("bn_relu_maxpool_fwd16");
  }
  DT_DISPATCH(dt, "bn_relu_maxpool_fwd", {
    constexpr int V = 4 * PoolVec<T>::V4;
    const long long total = (long long)N * Ho * Wo * (C / V);
    hipLaunchKernelGGL((bn_relu_maxpool_fwd_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const T*>(Y), scale,
                       shift, static_cast<T*>(P), amax, total, Hi, Wi, Ho, Wo, C / V);
  });
  return check_launch("bn_relu_maxpool_fwd");
}

// Backward through the (never stored) pre-pool activation, quad form. Windows of 3x3 / stride 2 / pad 1 tile the image into 2x2
// quads: quad (qy, qx) = pixels (2qy + a, 2qx + b), a, b in {0, 1}, lies inside its HOME window (qy, qx) (codes 4, 5, 7, 8); its right
// column is also column 0 of window (qy, qx+1) (codes 3, 6), its lower row is row 0 of window (qy+1, qx) (codes 1, 2), and pixel
// (1, 1) is code 0 of window (qy+1, qx+1). One thread = one quad x V channels: four window reads (pooled gradient + argmax bytes,
// the home one coalesced with the pooled tensor's own layout) serve four pixels, all twelve loads are issued before the first use,
// and there is no data-dependent loop. (The per-pixel gather it replaces read 2.25 windows per pixel in a divergent loop with a
// wait per window: 2.2-3.0 TB/s; rounds of dependent L2 hits, not bytes, were the bound.) Per pixel the contributions are added
// in window scan order, as the stand-alone maxpool_bwd_kernel does, and rounded to the storage type as the stored dZ0 was.
template <class T>
struct PoolQuad {
  static constexpr int V4 = PoolVec<T>::V4;
  f32x4 y[4][V4];      // pixel (a, b) at [2a + b]
  f32x4 dz[4][V4];
  bool va, vb;         // row 2qy+1 / column 2qx+1 exist
};

template <class T>
__device__ __forceinline__ void pool_quad(const T* __restrict__ dP, const unsigned char* __restrict__ amax, const T* __restrict__ Y,
                                          long long n, int qy, int qx, int cv, int Hi, int Wi, int Ho, int Wo, int CV, PoolQuad<T>& q) {
  constexpr int V4 = PoolVec<T>::V4, V = 4 * V4;
  const bool hr = qx + 1 < Wo, hd = qy + 1 < Ho;
  q.va = 2 * qy + 1 < Hi;
  q.vb = 2 * qx + 1 < Wi;
  const long long o = (((n * Ho + qy) * Wo + qx) * CV + cv) * V;
  const long long sW = (long long)CV * V, sH = (long long)Wo * CV * V;
  const long long oo[4] = {o, hr ? o + sW : o, hd ? o + sH : o, (hr && hd) ? o + sH + sW : o};
  const long long p = (((n * Hi + 2 * qy) * Wi + 2 * qx) * CV + cv) * V;
  const long long pH = (long long)Wi * CV * V;
  const long long pp[4] = {p, q.vb ? p + sW : p, q.va ? p + pH : p, (q.va && q.vb) ? p + pH + sW : p};
  unsigned a[4][V4];
  f32x4 d[4][V4];
#pragma unroll
  for (int w = 0; w < 4; ++w) ldv_stream<V4>(Y + pp[w], q.y[w]);
#pragma unroll
  for (int w = 0; w < 4; ++w) { ld_codes<V4>(amax + oo[w], a[w]); ldv<V4>(dP + oo[w], d[w]); }
#pragma unroll
  for (int k = 0; k < V4; ++k) {            // a window that does not exist matches no code
    if (!hr) { a[1][k] = 0xffffffffu; a[3][k] = 0xffffffffu; }
    if (!hd) { a[2][k] = 0xffffffffu; a[3][k] = 0xffffffffu; }
  }
#pragma unroll
  for (int k = 0; k < V4; ++k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned cH = (a[0][k] >> (8 * e)) & 0xffu, cR = (a[1][k] >> (8 * e)) & 0xffu;
      const unsigned cD = (a[2][k] >> (8 * e)) & 0xffu, cX = (a[3][k] >> (8 * e)) & 0xffu;
      const float dH = d[0][k][e], dR = d[1][k][e], dD = d[2][k][e], dX = d[3][k][e];
      q.dz[0][k][e] = cH == 4u ? dH : 0.f;
      q.dz[1][k][e] = (cH == 5u ? dH : 0.f) + (cR == 3u ? dR : 0.f);
      q.dz[2][k][e] = (cH == 7u ? dH : 0.f) + (cD == 1u ? dD : 0.f);
      q.dz[3][k][e] = (((cH == 8u ? dH : 0.f) + (cR == 6u ? dR : 0.f)) + (cD == 2u ? dD : 0.f)) + (cX == 0u ? dX : 0.f);
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) q.dz[w][k] = round_as<T>(q.dz[w][k]);
  }
}

// pass 1 of BatchNorm backward with dZ gathered through the max-pool. A block owns `rq` consecutive quad rows (n, qy) and all
// Wo x CV (quad, channel-vector) items of a row; a thread's channel vector is loop-invariant, its sums stay in registers, and
// the block's threads of one channel vector are combined in a fixed order (two LDS stages) into one partial row.
template <class T>
__global__ __launch_bounds__(1024) void bn_bwd_reduce_pool_kernel(const T* __restrict__ dP, const unsigned char* __restrict__ amax,
                                                                   const T* __restrict__ Y, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, float* __restrict__ partials,
                                                                   int quad_rows, int rq, int C, int CV, int cv_log2, int Hi, int Wi,
                                                                   int Ho, int Wo) {
  constexpr int V4 = PoolVec<T>::V4, V = 4 * V4;
  extern __shared__ f32x4 pool_red[];            // [2][V4][blockDim.x]
  const int nt = blockDim.x;
  const int cv = threadIdx.x & (CV - 1);
  const int c = cv * V;
  const int items = Wo * CV;
  f32x4 sc[V4], sh[V4], mu[V4], is[V4], s1[V4], s2[V4];
#pragma unroll
  for (int k = 0; k < V4; ++k) {
    sc[k] = ld4(scale + c + 4 * k); sh[k] = ld4(shift + c + 4 * k); mu[k] = ld4(mean + c + 4 * k); is[k] = ld4(invstd + c + 4 * k);
    s1[k] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int qr0 = blockIdx.x * rq;
  const int qr1 = qr0 + rq < quad_rows ? qr0 + rq : quad_rows;
  for (int qr = qr0; qr < qr1; ++qr) {
    const int n = qr / Ho, qy = qr - n * Ho;
    for (int it = threadIdx.x; it < items; it += nt) {
      PoolQuad<T> q;
      pool_quad<T>(dP, amax, Y, n, qy, it >> cv_log2, cv, Hi, Wi, Ho, Wo, CV, q);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const bool valid = ((w & 1) == 0 || q.vb) && ((w & 2) == 0 || q.va);
#pragma unroll
        for (int k = 0; k < V4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float yv = q.y[w][k][e];
            const float g = (valid && fmaf(yv, sc[k][e], sh[k][e]) > 0.f) ? q.dz[w][k][e] : 0.f;
            s1[k][e] += g;
            s2[k][e] = fmaf(g, (yv - mu[k][e]) * is[k][e], s2[k][e]);
          }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < V4; ++k) { pool_red[(0 * V4 + k) * nt + threadIdx.x] = s1[k]; pool_red[(1 * V4 + k) * nt + threadIdx.x] = s2[k]; }
  __syncthreads();
  // stage 2: G groups per channel vector, group g adds the threads g, g+G, ... of its vector; stage 3: one thread adds the groups
  const int per_cv = nt >> cv_log2;
  const int G = per_cv < 8 ? per_cv : 8;
  if ((int)threadIdx.x < G * CV) {
    const int g = threadIdx.x >> cv_log2;
#pragma unroll
    for (int k = 0; k < V4; ++k) { s1[k] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int t = g; t < per_cv; t += G)
#pragma unroll
      for (int k = 0; k < V4; ++k) { s1[k] += pool_red[(0 * V4 + k) * nt + t * CV + cv]; s2[k] += pool_red[(1 * V4 + k) * nt + t * CV + cv]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < G * CV) {
#pragma unroll
    for (int k = 0; k < V4; ++k) { pool_red[(0 * V4 + k) * nt + threadIdx.x] = s1[k]; pool_red[(1 * V4 + k) * nt + threadIdx.x] = s2[k]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < CV) {
    for (int g = 1; g < G; ++g)
#pragma unroll
      for (int k = 0; k < V4; ++k) { s1[k] += pool_red[(0 * V4 + k) * nt + g * CV + cv]; s2[k] += pool_red[(1 * V4 + k) * nt + g * CV + cv]; }
#pragma unroll
    for (int k = 0; k < V4; ++k) {
      st4(partials + ((long long)blockIdx.x * 2 + 0) * C + c + 4 * k, s1[k]);
      st4(partials + ((long long)blockIdx.x * 2 + 1) * C + c + 4 * k, s2[k]);
    }
  }
}

// pass 2: dY = scale * (g - c1 - yhat * c2) with g gathered through the max-pool; one quad row (n, qy) per block.
template <class T>
__global__ __launch_bounds__(1024) void bn_bwd_apply_pool_kernel(const T* __restrict__ dP, const unsigned char* __restrict__ amax,
                                                                  const T* __restrict__ Y, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ c1,
                                                                  const float* __restrict__ c2, T* __restrict__ dY, int CV, int cv_log2,
                                                                  int Hi, int Wi, int Ho, int Wo) {
  constexpr int V4 = PoolVec<T>::V4, V = 4 * V4;
  const int n = blockIdx.x / Ho, qy = blockIdx.x - n * Ho;
  const int items = Wo * CV;
  const int cv = threadIdx.x & (CV - 1);
  const int c = cv * V;
  f32x4 sc[V4], sh[V4], mu[V4], is[V4], k1[V4], k2[V4];
#pragma unroll
  for (int k = 0; k < V4; ++k) {
    sc[k] = ld4(scale + c + 4 * k); sh[k] = ld4(shift + c + 4 * k); mu[k] = ld4(mean + c + 4 * k); is[k] = ld4(invstd + c + 4 * k);
    k1[k] = ld4(c1 + c + 4 * k); k2[k] = ld4(c2 + c + 4 * k);
  }
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const int qx = it >> cv_log2;
    PoolQuad<T> q;
    pool_quad<T>(dP, amax, Y, n, qy, qx, cv, Hi, Wi, Ho, Wo, CV, q);
    const long long p = ((((long long)n * Hi + 2 * qy) * Wi + 2 * qx) * CV + cv) * V;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      f32x4 o[V4];
#pragma unroll
      for (int k = 0; k < V4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float yv = q.y[w][k][e];
          const float g = fmaf(yv, sc[k][e], sh[k][e]) > 0.f ? q.dz[w][k][e] : 0.f;
          const float yh = (yv - mu[k][e]) * is[k][e];
          o[k][e] = sc[k][e] * (g - k1[k][e] - yh * k2[k][e]);
        }
      const bool valid = ((w & 1) == 0 || q.vb) && ((w & 2) == 0 || q.va);
      if (valid) stv<V4>(dY + p + (long long)(w & 1) * CV * V + (long long)(w >> 1) * Wi * CV * V, o, true);
    }
  }
}

// Quad rows per block of the pooled reduce: 8, or more when that would give more partial rows than the plain fp32 reduce of the
// same tensor writes (the size every BatchNorm workspace is laid out for).
static inline void pool_reduce_geometry(int N, int Hi, int Wi, int C, int* rq, int* nblk) {
  const int Ho = (Hi + 2 - 3) / 2 + 1;
  const int quad_rows = N * Ho;
  const int cap = bn_bwd_partial_rows((long long)N * Hi * Wi, C, DT_F32);
  int r = ceil_div(quad_rows, cap);
  if (r < 8) r = 8;
  *rq = r;
  *nblk = ceil_div(quad_rows, r);
}

// rows of the partial buffer the pooled reduce writes
int bn_bwd_pool_partial_rows(int N, int Hi, int Wi, int C) {
  int rq, nblk;
  pool_reduce_geometry(N, Hi, Wi, C, &rq, &nblk);
  return nblk;
}

int launch_bn_bwd_reduce_pool(const void* dP, const unsigned char* amax, const void* Y, const float* scale, const float* shift,
                              const float* mean, const float* invstd, float* partials, int N, int Hi, int Wi, int C, int dt,
                              hipStream_t s) {
  R3M_REQUIRE(is_pow2(C) && C >= 8 && C <= 1024, "bn_bwd_reduce_pool: C=%d must be a power of two in [8, 1024]", C);
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  int rq, nblk;
  pool_reduce_geometry(N, Hi, Wi, C, &rq, &nblk);
  DT_DISPATCH(dt, "bn_bwd_reduce_pool", {
    constexpr int V4 = PoolVec<T>::V4, V = 4 * V4;
    const int CV = C / V, nt = pool_row_threads(Wo * CV);
    hipLaunchKernelGGL((bn_bwd_reduce_pool_kernel<T>), dim3(nblk), dim3(nt), (size_t)2 * V4 * nt * sizeof(f32x4), s,
                       static_cast<const T*>(dP), amax, static_cast<const T*>(Y), scale, shift, mean, invstd, partials, N * Ho, rq, C,
                       CV, ilog2_exact(CV), Hi, Wi, Ho, Wo);
  });
  return check_launch("bn_bwd_reduce_pool");
}

int launch_bn_bwd_apply_pool(const void* dP, const unsigned char* amax, const void* Y, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* c1, const float* c2, void* dY, int N, int Hi, int Wi,
                             int C, int dt, hipStream_t s) {
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  R3M_REQUIRE(is_pow2(C) && C >= 8 && C <= 1024, "bn_bwd_apply_pool: C=%d must be a power of two in [8, 1024]", C);
  DT_DISPATCH(dt, "bn_bwd_apply_pool", {
    constexpr int V = 4 * PoolVec<T>::V4;
    const int CV = C / V;
    hipLaunchKernelGGL((bn_bwd_apply_pool_kernel<T>), dim3(N * Ho), dim3(pool_row_threads(Wo * CV)), 0, s, static_cast<const T*>(dP),
                       amax, static_cast<const T*>(Y), scale, shift, mean, invstd, c1, c2, static_cast<T*>(dY), CV, ilog2_exact(CV), Hi, Wi,
                       Ho, Wo);
  });
  return check_launch("bn_bwd_apply_pool");
}

// ---------------------------------------------------------------------------------------------------------
// AdaptiveAvgPool2d(1) + flatten: [N, HW, C] -> [N, C]  and its backward (broadcast of dH / HW)
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ X, float* __restrict__ H, long long total,
                                                           int HW, int C4) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  const long long n = idx / C4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < HW; ++p) s += ld4t(X + ((n * HW + p) * C4 + c4) * 4);
  const float d = (float)HW;
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = s[e] / d;
  st4(H + idx * 4, s);
}

int launch_avgpool_fwd(const void* X, float* H, int N, int HW, int C, int dt, hipStream_t s) {
  const long long total = (long long)N * (C / 4);
  DT_DISPATCH(dt, "avgpool_fwd",
              hipLaunchKernelGGL((avgpool_fwd_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const T*>(X), H, total, HW, C / 4));
  return check_launch("avgpool_fwd");
}

template <class T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dH, T* __restrict__ dX, long long total,
                                                           int HW, int C4) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  const long long n = idx / ((long long)HW * C4);
  f32x4 g = ld4(dH + (n * C4 + c4) * 4);
  const float d = (float)HW;
#pragma unroll
  for (int e = 0; e < 4; ++e) g[e] = g[e] / d;
  st4t(dX + idx * 4, g);
}

int launch_avgpool_bwd(const float* dH, void* dX, int N, int HW, int C, int dt, hipStream_t s) {
  const long long total = (long long)N * HW * (C / 4);
  DT_DISPATCH(dt, "avgpool_bwd",
              hipLaunchKernelGGL((avgpool_bwd_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0, s, dH, static_cast<T*>(dX), total, HW, C / 4));
  return check_launch("avgpool_bwd");
}

}  // namespace r3m
