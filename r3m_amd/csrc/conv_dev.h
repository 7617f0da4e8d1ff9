// r3m_amd — device helpers shared by the convolution translation units (conv.hip: fp32, conv_bf16.hip: bf16 activations).
#pragma once
#include "common.h"
#include <utility>

namespace r3m {

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 4 consecutive activations as fp32, whatever the storage type (bf16 <-> fp32 conversions are exact / round-to-nearest-even)
__device__ __forceinline__ f32x4 ld4t(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4t(const bf16_t* p) { return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4); }
__device__ __forceinline__ void st4t(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4t(bf16_t* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4); }

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// conv outputs are written once and next read a whole tensor later: optional non-temporal stores (R3M_EPI_NT)
#ifndef R3M_EPI_NT
#define R3M_EPI_NT 0
#endif
typedef unsigned int epi_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int epi_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_out(float* p, f32x4 v) {
#if R3M_EPI_NT
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void st4_out(bf16_t* p, f32x4 v) {
#if R3M_EPI_NT
  __builtin_nontemporal_store(__builtin_bit_cast(epi_u32x2, __builtin_convertvector(v, bf16x4)), reinterpret_cast<epi_u32x2*>(p));
#else
  *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4);
#endif
}
constexpr unsigned BUF_OOB = 0x7FFFF000u;   // >= any descriptor byte count used here: such a lane's 16 bytes are dropped / arrive as zeros
// 16-byte store through a buffer descriptor [base, base + bytes): wave-uniform base, 32-bit per-lane byte offset, lanes whose
// offset is past the end store nothing. (Target builtins stay inside device-compile guards: see buf_dma16.)
__device__ __forceinline__ void buf_store16(void* base, int bytes, unsigned voff, epi_u32x4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_buffer_store_b128(v, __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000), voff, 0, R3M_EPI_NT ? 2 : 0);
#endif
}
// 16-byte load through a buffer descriptor: lanes whose offset is past the end read zeros
__device__ __forceinline__ epi_u32x4 buf_load16(const void* base, int bytes, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000), voff, 0, 0);
#else
  return epi_u32x4{0u, 0u, 0u, 0u};
#endif
}
__device__ __forceinline__ void st8_out(bf16_t* p, bf16x8 v) {
#if R3M_EPI_NT
  __builtin_nontemporal_store(__builtin_bit_cast(epi_u32x4, v), reinterpret_cast<epi_u32x4*>(p));
#else
  *reinterpret_cast<bf16x8*>(p) = v;
#endif
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>) — indices usable as array subscripts without scratch
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed; speed only). Remap so that
// each XCD walks a contiguous range of logical tiles: tiles that share an operand panel then share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  const int xcd = bid % NX, idx = bid / NX;
  const int q = nwg / NX, r = nwg % NX;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// BatchNorm partials of one block tile: per-column sum / sum of squares over the block's rows -> stats[mt][2][Nc].
// PER_WM: every wave row writes its own partial row (stats[mt * WM + wm]) — a 256-row tile then produces the same partial-row
// geometry as two 128-row tiles, and the engine's row count does not depend on the tile the launcher picks.
template <int BM, int BN, int WM, int WN, bool PER_WM = false>
__device__ __forceinline__ void gg_stats(const GatherGemmParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem, int n0,
                                         int mt) {
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  {
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
      if (PER_WM) {
        const int col = n0 + (wn * TN + tn) * 32 + lane;
        const long long prow = (long long)mt * WM + wm;
        if (lane < 32 && col < p.Nc && prow * (BM / WM) < p.M) {
          p.stats[(prow * 2 + 0) * p.Nc + col] = s;
          p.stats[(prow * 2 + 1) * p.Nc + col] = ss;
        }
      } else if (lane < 32) {
        const int c = (wn * TN + tn) * 32 + lane;
        red[(wm * 2 + 0) * BN + c] = s;
        red[(wm * 2 + 1) * BN + c] = ss;
      }
    }
    if (PER_WM) return;
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
}

// ---- EPI_BNRED: BatchNorm-backward partials accumulated in the store phase of a dgrad epilogue -----------------------------
// In the store phase a lane holds V consecutive columns of one result row per store instruction and walks 64 rows of its wave in
// steps of RPI = 64 / LPR rows (LPR = lanes per row). Sums live in 2 V registers; after the 64 rows the lanes that share a column
// vector (lane, lane ^ LPR, lane ^ 2 LPR, ...) are combined with shuffles and lanes < LPR write the partial row. Every (64-row
// group, column range) belongs to exactly one wave: no atomics, no block barrier, fixed summation order.
template <int V>
struct BnRedAcc {
  float s1[V], s2[V];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int e = 0; e < V; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  }
};

// coefficient vectors of the lane's V columns (loop-invariant)
template <int V>
struct BnRedCoef {
  float sc[V], sh[V], mu[V];
  __device__ __forceinline__ void load(const GatherGemmParams& p, int gcol) {
    const bool ok = gcol < p.Nc;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      mu[e] = ok ? p.bn_mean[gcol + e] : 0.f;
      sc[e] = (ok && !p.bn_bits) ? p.bn_scale[gcol + e] : 0.f;
      sh[e] = (ok && !p.bn_bits) ? p.bn_shift[gcol + e] : 0.f;
    }
  }
};

// dz: the V values just stored at element offset eo (already rounded to the storage type), y: the BatchNorm input there
template <int V>
__device__ __forceinline__ void bnred_add(BnRedAcc<V>& a, const BnRedCoef<V>& k, const GatherGemmParams& p, const float (&dz)[V],
                                          const float (&y)[V], long long eo) {
  unsigned bits = ~0u;
  if (p.bn_bits) bits = p.bn_bits[eo >> 5] >> (int)(eo & 31);     // eo is a multiple of V (4 or 8): the V bits sit in one word
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const bool on = p.bn_bits ? ((bits >> e) & 1u) != 0u : fmaf(y[e], k.sc[e], k.sh[e]) > 0.f;
    const float g = on ? dz[e] : 0.f;
    a.s1[e] += g;
    a.s2[e] = fmaf(g, y[e] - k.mu[e], a.s2[e]);
  }
}

template <int V, int LPR>
__device__ __forceinline__ void bnred_flush(BnRedAcc<V>& a, const GatherGemmParams& p, int lane, long long prow, int gcol, bool group_valid) {
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      a.s1[e] += __shfl_xor(a.s1[e], o);
      a.s2[e] += __shfl_xor(a.s2[e], o);
    }
  if (lane < LPR && gcol < p.Nc && group_valid) {
    float* d1 = p.stats + (prow * 2 + 0) * p.Nc + gcol;
    float* d2 = p.stats + (prow * 2 + 1) * p.Nc + gcol;
#pragma unroll
    for (int e = 0; e < V; e += 4) {
      *reinterpret_cast<f32x4*>(d1 + e) = f32x4{a.s1[e], a.s1[e + 1], a.s1[e + 2], a.s1[e + 3]};
      *reinterpret_cast<f32x4*>(d2 + e) = f32x4{a.s2[e], a.s2[e + 1], a.s2[e + 2], a.s2[e + 3]};
    }
  }
  a.clear();
}

// ---- shared epilogue: BatchNorm partials from the accumulators + LDS-transposed, 16-byte-per-lane output stores ----
// Each wave transposes its 32 x (TN*32) accumulator slabs through a private LDS slab so that every store instruction
// writes whole 128/256-byte row segments (dwordx4 per lane) instead of single dwords; no block barrier is involved, so the
// waves of a block drain independently (the caller's barrier after the K loop already retired the operand tiles); flag-dependent operand reads are
// compile-time (EPI) so the plain-store path carries no loads (and no vmcnt waits between stores).
template <int BM, int BN, int WM, int WN, int EPI, int SMEM_FLOATS, class OT = float>
__device__ __forceinline__ void gg_epilogue(const GatherGemmParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem,
                                            int m0, int n0, int mt) {
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane & 31;

  if (EPI & EPI_STATS) gg_stats<BM, BN, WM, WN>(p, acc, smem, n0, mt);

  constexpr int CW = TN * 32;          // columns owned by the wave
  constexpr int CS = CW + 4;           // padded slab row stride (floats)
  constexpr int F4 = CW / 4;           // float4 per slab row
  constexpr int RPI = 64 / F4;         // rows covered per store instruction
  static_assert(WM * WN * 32 * CS <= SMEM_FLOATS, "epilogue slab must fit in the operand tiles' LDS");
  float* slab = smem + wave * 32 * CS;
  const bool out_simple = (p.os == 1);
  const int hwg = p.Hg * p.Wg;
  const int ecol = (lane % F4) * 4;
  const int erow = lane / F4;
  const int gcol = n0 + wn * CW + ecol;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if ((EPI & EPI_BIAS) && gcol < p.Nc) bias4 = ldg4(p.bias + gcol);
  f32x4 aff_sc = {1.f, 1.f, 1.f, 1.f}, aff_sh = {0.f, 0.f, 0.f, 0.f};
  if ((EPI & EPI_AFFINE) && gcol < p.Nc) { aff_sc = ldg4(p.bn_scale + gcol); aff_sh = ldg4(p.bn_shift + gcol); }
  BnRedAcc<4> bnacc;
  BnRedCoef<4> bncoef;
  if constexpr ((EPI & EPI_BNRED) != 0) {
    static_assert(TM % 2 == 0, "EPI_BNRED: a wave covers whole 64-row groups");
    bnacc.clear();
    bncoef.load(p, gcol);
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + tn * 32 + lrow] = acc[tm][tn][r];
    __builtin_amdgcn_wave_barrier();   // the slab is private to the wave: its own LDS accesses execute in order
    auto roff_of = [&](int row) -> long long {
      if (out_simple) return (long long)row * p.Nc;
      const int n = row / hwg;
      const int rem = row - n * hwg;
      const int gy = rem / p.Wg;
      const int gx = rem - gy * p.Wg;
      return (((long long)n * p.Ho + (gy * p.os + p.ooy)) * p.Wo + (gx * p.os + p.oox)) * p.Nc;
    };
    // Read-modify-write epilogues: the operands of the next YPRE rows (y of EPI_BNRED, the residual gradient of EPI_MASKED_ADD,
    // the old result of EPI_ACCUM) are requested BEFORE those rows' stores are issued — the compiler cannot move a load above an
    // earlier store through un-restricted pointers, so without this every row would pay a full load latency.
    constexpr int NIT = 32 / RPI;
    constexpr int YPRE = NIT < 4 ? NIT : 4;
    constexpr bool PRELOAD = (EPI & (EPI_BNRED | EPI_MASKED_ADD | EPI_ACCUM)) != 0;
    f32x4 ypre[YPRE], gpre[YPRE], opre[YPRE];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if constexpr (PRELOAD) {
        if (it % YPRE == 0) {
#pragma unroll
          for (int q = 0; q < YPRE; ++q) {
            const int rowq = m0 + (wm * TM + tm) * 32 + (it + q) * RPI + erow;
            ypre[q] = gpre[q] = opre[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (it + q < NIT && rowq < p.M && gcol < p.Nc) {
              const long long eq = roff_of(rowq) + gcol;
              if constexpr ((EPI & EPI_BNRED) != 0) ypre[q] = ld4t(reinterpret_cast<const OT*>(p.bn_y) + eq);
              if constexpr ((EPI & EPI_MASKED_ADD) != 0) gpre[q] = ld4t(reinterpret_cast<const OT*>(p.add0) + eq);
              if constexpr ((EPI & EPI_ACCUM) != 0) opre[q] = ld4t(reinterpret_cast<const OT*>(p.out) + eq);
            }
          }
        }
      }
      const int lr = it * RPI + erow;
      const int row = m0 + (wm * TM + tm) * 32 + lr;
      if (row < p.M && gcol < p.Nc) {
        const long long roff = roff_of(row);
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + lr * CS + ecol);
        OT* dst = reinterpret_cast<OT*>(p.out) + roff + gcol;     // activation pointers are typed float in the params struct;
        if (EPI & EPI_AFFINE) {                                   // eval-mode BatchNorm: the same fmaf(y, scale, shift) bn_act_fwd computes
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], aff_sc[e], aff_sh[e]);
        }
        if (EPI & EPI_BIAS) v += bias4;                           // with OT = bf16_t they address bf16 tensors
        if (EPI & EPI_ACCUM) v += opre[it % YPRE];
        if (EPI & EPI_MASKED_ADD) {
          const f32x4 g = gpre[it % YPRE];
          if (p.addbits) {
            const long long i4 = (roff + gcol) >> 2;
            const unsigned nb = (p.addbits[i4 >> 3] >> (4 * (int)(i4 & 7))) & 15u;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += ((nb >> e) & 1u) ? g[e] : 0.f;
          } else {
            const f32x4 z = ld4t(reinterpret_cast<const OT*>(p.add1) + roff + gcol);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (z[e] > 0.f) ? g[e] : 0.f;
          }
        }
        if (EPI & EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (EPI & EPI_MASK_OUT) {
          const f32x4 z = ld4t(reinterpret_cast<const OT*>(p.add1) + roff + gcol);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (z[e] > 0.f) ? v[e] : 0.f;
        }
        st4_out(dst, v);
        if constexpr ((EPI & EPI_BNRED) != 0) {
          f32x4 vr = v;                                                    // as stored: rounded once for bf16 tensors
          if constexpr (sizeof(OT) == 2) vr = __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4);
          const f32x4 yv = ypre[it % YPRE];
          const float dz[4] = {vr[0], vr[1], vr[2], vr[3]};
          const float yy[4] = {yv[0], yv[1], yv[2], yv[3]};
          bnred_add<4>(bnacc, bncoef, p, dz, yy, roff + gcol);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // the slab is private to the wave: its own LDS accesses execute in order
    if constexpr ((EPI & EPI_BNRED) != 0) {
      if (tm & 1) {                                                        // two 32-row tiles = one 64-row group
        const int g0 = m0 + (wm * TM + tm - 1) * 32;
        bnred_flush<4, F4>(bnacc, p, lane, (long long)(g0 >> 6), gcol, g0 < p.M);
      }
    }
  }
}

// bf16 output variant of the store part: 8 columns (16 bytes) per lane. EPI: 0 / EPI_ACCUM / EPI_MASKED_ADD (with the 1-bit
// mask, or a bf16 activation as the mask source). Each wave transposes its 64 x 64 accumulators through a PRIVATE LDS slab,
// so no block barrier is needed (a wave's own LDS accesses execute in order): waves drain independently, which matters
// because for the 1x1 convolutions with few K tiles the epilogue is more than half of a block's lifetime.
// Plain stores round to bf16 BEFORE the transposition (slab of bf16, one pass over all rows); the read-modify-write
// variants keep an fp32 slab (two passes of 32 rows) so that the sum is rounded once.
template <int BM, int BN, int WM, int WN, int EPI, int SMEM_FLOATS>
__device__ __forceinline__ void gg_store_bf16(const GatherGemmParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem,
                                              int m0, int n0) {
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int CW = TN * 32;          // columns owned by the wave
  constexpr int F8 = CW / 8;           // 16-byte stores per slab row
  constexpr int RPI = 64 / F8;         // rows covered per store instruction
  constexpr bool RMW = (EPI & (EPI_ACCUM | EPI_MASKED_ADD)) != 0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane & 31;
  const bool out_simple = (p.os == 1);
  const int hwg = p.Hg * p.Wg;
  const int ecol = (lane % F8) * 8;
  const int erow = lane / F8;
  const int gcol = n0 + wn * CW + ecol;
  bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
  BnRedAcc<8> bnacc;
  BnRedCoef<8> bncoef;
  if constexpr ((EPI & EPI_BNRED) != 0) {
    bnacc.clear();
    bncoef.load(p, gcol);
  }
  // y of the BatchNorm is requested for all rows of a pass BEFORE the pass's stores (a load cannot move above an earlier store,
  // so loading it next to its use made every store instruction wait a full memory latency — round 4)
  auto bnred8 = [&](const bf16x8 dzv, const bf16x8 yv, long long eo) __attribute__((always_inline)) {
    float dz[8], yy[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dz[e] = (float)dzv[e]; yy[e] = (float)yv[e]; }
    bnred_add<8>(bnacc, bncoef, p, dz, yy, eo);
  };
  auto row_off = [&](int row) -> long long {
    if (out_simple) return (long long)row * p.Nc;
    const int n = row / hwg;
    const int rem = row - n * hwg;
    const int gy = rem / p.Wg;
    const int gx = rem - gy * p.Wg;
    return (((long long)n * p.Ho + (gy * p.os + p.ooy)) * p.Wo + (gx * p.os + p.oox)) * p.Nc;
  };
  // EPI_AFFINE (inference forward): per-column coefficients — in the accumulator layout a lane owns column tn * 32 + lrow of the wave's
  // columns (plain stores), in the store layout 8 consecutive columns from gcol (read-modify-write stores)
  float aff_sc[TN], aff_sh[TN];
  float aff_sc8[8], aff_sh8[8];
  if constexpr ((EPI & EPI_AFFINE) != 0) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int c = min(n0 + wn * CW + tn * 32 + lrow, p.Nc - 1);
      aff_sc[tn] = p.bn_scale[c];
      aff_sh[tn] = p.bn_shift[c];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = min(gcol + e, p.Nc - 1);
      aff_sc8[e] = RMW ? p.bn_scale[c] : 1.f;
      aff_sh8[e] = RMW ? p.bn_shift[c] : 0.f;
    }
  }
  if constexpr (!RMW) {
    constexpr int CSH = CW + 8;          // slab row stride in bf16 (16-byte aligned rows, 4-bank skew)
    constexpr int TMP = TM > 2 ? 2 : TM; // row tiles per pass (64 slab rows per wave at most)
    static_assert(TM % TMP == 0, "whole passes");
    static_assert(WM * WN * TMP * 32 * CSH * 2 <= SMEM_FLOATS * 4, "epilogue slab must fit in the operand tiles' LDS");
    bf16_t* slab = reinterpret_cast<bf16_t*>(smem) + wave * TMP * 32 * CSH;
#pragma unroll
    for (int ps = 0; ps < TM / TMP; ++ps) {
      if (ps) __builtin_amdgcn_wave_barrier();   // the previous pass's slab reads are done (in-order LDS per wave)
#pragma unroll
      for (int tm = 0; tm < TMP; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[ps * TMP + tm][tn][r];
            if constexpr ((EPI & EPI_AFFINE) != 0) v = fmaf(v, aff_sc[tn], aff_sh[tn]);     // eval-mode BatchNorm on the fp32 accumulator, one rounding
            if constexpr ((EPI & EPI_RELU) != 0) v = fmaxf(v, 0.f);
            slab[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CSH + tn * 32 + lrow] = (bf16_t)v;
          }
      __builtin_amdgcn_wave_barrier();
      bool stored = false;
      {
        // Contiguous output rows (every forward, stride-1 dgrad): the wave's 64 rows go out through a buffer descriptor that starts
        // at its first row and ends with the tensor — a lane's address is a 32-bit offset advanced by one add per store, rows past
        // M fall off the descriptor's end. (The generic path below costs a 64-bit multiply-add, a compare and an exec mask per
        // store: ~12 instructions x 16 stores per lane; this epilogue is as long as the MFMAs of a K = 256 tile.)
        const int rowbase = m0 + (__builtin_amdgcn_readfirstlane(wm) * TM + ps * TMP) * 32;
        if (out_simple && rowbase < p.M) {
          const long long left = (long long)(p.M - rowbase) * p.Nc * 2;
          bf16_t* rbase = outp + (long long)rowbase * p.Nc;
          const int rbytes = left < (long long)BUF_OOB ? (int)left : (int)BUF_OOB;
          unsigned voff = gcol < p.Nc ? (unsigned)((erow * p.Nc + gcol) * 2) : BUF_OOB;
          const unsigned vstep = (unsigned)(RPI * p.Nc * 2);
          constexpr int NITP = TMP * 32 / RPI;
          bf16x8 ypre[(EPI & EPI_BNRED) ? NITP : 1];
          if constexpr ((EPI & EPI_BNRED) != 0) {        // EPI_BNRED (round 4): y of the whole pass through the same offsets, before the stores
            const bf16_t* ybase = reinterpret_cast<const bf16_t*>(p.bn_y) + (long long)rowbase * p.Nc;
#pragma unroll
            for (int it = 0; it < NITP; ++it) ypre[it] = __builtin_bit_cast(bf16x8, buf_load16(ybase, rbytes, voff + it * vstep));
          }
#pragma unroll
          for (int it = 0; it < NITP; ++it) {
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(slab + (it * RPI + erow) * CSH + ecol);
            buf_store16(rbase, rbytes, voff, __builtin_bit_cast(epi_u32x4, ov));
            if constexpr ((EPI & EPI_BNRED) != 0) {
              if (voff < (unsigned)rbytes) bnred8(ov, ypre[it], (long long)rowbase * p.Nc + (voff >> 1));   // rows past M hold no mask bits
            }
            voff += vstep;
          }
          stored = true;
        }
        if (out_simple) stored = true;       // rowbase >= M: nothing of this pass is inside the tensor
      }
      if (!stored) {
        constexpr int NITP = TMP * 32 / RPI;
        bf16x8 ypre[(EPI & EPI_BNRED) ? NITP : 1];
        if constexpr ((EPI & EPI_BNRED) != 0) {
#pragma unroll
          for (int it = 0; it < NITP; ++it) {
            const int row = m0 + (wm * TM + ps * TMP) * 32 + it * RPI + erow;
            if (row < p.M && gcol < p.Nc)
              ypre[it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.bn_y) + row_off(row) + gcol);
          }
        }
#pragma unroll
        for (int it = 0; it < NITP; ++it) {
          const int lr = it * RPI + erow;
          const int row = m0 + (wm * TM + ps * TMP) * 32 + lr;
          if (row < p.M && gcol < p.Nc) {
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(slab + lr * CSH + ecol);
            const long long eo = row_off(row) + gcol;
            st8_out(outp + eo, ov);
            if constexpr ((EPI & EPI_BNRED) != 0) bnred8(ov, ypre[it], eo);
          }
        }
      }
      if constexpr ((EPI & EPI_BNRED) != 0) {
        static_assert(TMP == 2, "EPI_BNRED: one 64-row group per pass");
        const int g0 = m0 + (wm * TM + ps * TMP) * 32;
        bnred_flush<8, F8>(bnacc, p, lane, (long long)(g0 >> 6), gcol, g0 < p.M);
      }
    }
  } else {
    constexpr int CS = CW + 4;           // padded slab row stride (floats)
    static_assert(WM * WN * 32 * CS <= SMEM_FLOATS, "epilogue slab must fit in the operand tiles' LDS");
    float* slab = smem + wave * 32 * CS;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + tn * 32 + lrow] = acc[tm][tn][r];
      __builtin_amdgcn_wave_barrier();
      // operands of the tile's rows requested before any of its stores (see gg_epilogue: loads cannot move above earlier stores)
      constexpr int NIT = 32 / RPI;
      bf16x8 opre[NIT], gpre[NIT], ypre[NIT];
#pragma unroll
      for (int q = 0; q < NIT; ++q) {
        const int rowq = m0 + (wm * TM + tm) * 32 + q * RPI + erow;
        if (rowq < p.M && gcol < p.Nc) {
          const long long eq = row_off(rowq) + gcol;
          if constexpr ((EPI & EPI_ACCUM) != 0) opre[q] = *reinterpret_cast<const bf16x8*>(outp + eq);
          if constexpr ((EPI & EPI_MASKED_ADD) != 0) gpre[q] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.add0) + eq);
          if constexpr ((EPI & EPI_BNRED) != 0) ypre[q] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.bn_y) + eq);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int lr = it * RPI + erow;
        const int row = m0 + (wm * TM + tm) * 32 + lr;
        if (row < p.M && gcol < p.Nc) {
          const long long eo = row_off(row) + gcol;      // element offset, multiple of 8
          float v[8];
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + lr * CS + ecol);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + lr * CS + ecol + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
          if constexpr ((EPI & EPI_AFFINE) != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], aff_sc8[e], aff_sh8[e]);
          }
          if (EPI & EPI_ACCUM) {
            const bf16x8 o = opre[it];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)o[e];
          }
          if constexpr ((EPI & EPI_RELU) != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (EPI & EPI_MASKED_ADD) {
            const bf16x8 g = gpre[it];
            if (p.addbits) {
              const unsigned nb = (p.addbits[eo >> 5] >> (int)(eo & 31)) & 255u;
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += ((nb >> e) & 1u) ? (float)g[e] : 0.f;
            } else {
              const bf16x8 z = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.add1) + eo);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += ((float)z[e] > 0.f) ? (float)g[e] : 0.f;
            }
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
          st8_out(outp + eo, o);
          if constexpr ((EPI & EPI_BNRED) != 0) bnred8(o, ypre[it], eo);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if constexpr ((EPI & EPI_BNRED) != 0) {
        if (tm & 1) {                                                      // two 32-row tiles = one 64-row group
          const int g0 = m0 + (wm * TM + tm - 1) * 32;
          bnred_flush<8, F8>(bnacc, p, lane, (long long)(g0 >> 6), gcol, g0 < p.M);
        }
      }
    }
    static_assert(!(EPI & EPI_BNRED) || TM % 2 == 0, "EPI_BNRED: waves cover whole 64-row groups");
  }
}

// ---- LDS DMA through a buffer descriptor (round 3; see the wgrad kernels) ----

// One `buffer_load_dwordx4 ... lds`: 16 bytes per lane from base + voff (zeros when voff >= bytes) to lds + 16 * lane. The
// descriptor (base, bytes) must be wave-uniform. The body exists in the device pass only (the host pass of hipcc has no such
// builtin and would silently drop the kernels that call it).
// buf_dma16 with the descriptor words forced into scalar registers. Under register pressure (the 192-accumulator bf16 weight
// gradient) the compiler parked the loop-carried base / size in VECTOR registers and wrapped every DMA instruction in a waterfall
// loop (4 readfirstlanes, two 64-bit compares, an exec save/restore and a branch: 12 instructions, 36 loops in that kernel).
__device__ __forceinline__ void buf_dma16_uniform(const void* base, int bytes, void* lds, unsigned voff, int soff = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long v = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  void* b = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(b, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000),
                                           (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}
__device__ __forceinline__ void buf_dma16(const void* base, int bytes, void* lds, unsigned voff, int soff = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
  // soff: wave-uniform byte offset added to the address (an SGPR operand of the instruction). Callers only put offsets there that
  // keep a valid lane inside the tensor and an out-of-range lane out of range, so the result does not depend on whether the
  // hardware includes it in the range check (tools/micro/bufload.hip: gfx950 does not)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000 /* raw, 32-bit */),
                                           (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

// per-thread descriptor of one staged A row: image base offset + top-left input pixel of the GEMM row
struct RowDesc {
  long long base;
  int iy, ix;
};

__device__ __forceinline__ RowDesc decode_row(const GatherGemmParams& p, int m) {
  RowDesc d;
  d.base = 0; d.iy = 0; d.ix = 0;
  if (m < p.M) {
    if (p.simple_rows) {
      d.base = (long long)m * p.Ci;
    } else {
      const int hw = p.Hg * p.Wg;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int gy = rem / p.Wg;
      const int gx = rem - gy * p.Wg;
      d.base = (long long)n * p.Hi * p.Wi * p.Ci;
      d.iy = gy * p.is;
      d.ix = gx * p.is;
    }
  }
  return d;
}

}  // namespace r3m
