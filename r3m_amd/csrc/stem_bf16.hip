// r3m_amd — the stem (x/255 -> Normalize -> conv 7x7 stride 2 pad 3, 3 -> 64; /root/reference/r3m/models/models_r3m.py:97-99 into
// torchvision's conv1) on the bf16 MFMA, used by bf16 plans: with the fp32 kernels of conv.hip the stem was 8 % of the bf16
// step at 35-60 % of the (16x slower) fp32 matrix rate.
//
// Same geometry trick as the fp32 stem: for a fixed kernel row kh the 7 x 3 (kw, c) taps of an output pixel are 21 CONSECUTIVE
// elements of a channel-interleaved image row, starting at 6*ox. Here the normalised frames are kept as a PADDED bf16 image
//     xn16[f][r][e],  r = iy + 3 in [0, 232),  e = 9 + 3*ix + c in [0, 704)       (zeros outside the picture)
// so that the 13 input rows an output tile needs are ONE contiguous 18 304-byte block: the forward stages it with 18 direct-to-LDS
// DMA instructions and no VALU, and the A fragment of MFMA step (kh, s) is patch[2*oy + kh][6*ox + 16*s + 8*h .. +7] — a per-lane
// base plus an immediate. K is walked as 7 x 32 (taps 21..31 multiply zero weights).
//   forward : out[pixel, n]  = sum_{kh, j} patch[..] * w[n][kh][j]         fp32 accumulation, BatchNorm partials from the accumulators
//   wgrad   : dW[n][kh][j]   = sum_pixels dY[pixel][n] * patch[2*oy+kh][6*ox+j]
//             contraction over pixels: dY arrives [pixel][n], i.e. K-strided -> ds_read_b64_tr_b16 (as wgrad_bf16_kernel); the
//             patch operand is 8 pixels at a 12-byte stride -> eight ds_read_u16 per operand, packed in registers.
#include "common.h"
#include "conv_dev.h"
#include "augment_dev.h"
#include <cstring>

namespace r3m {

constexpr int XN_ROW = 704;            // elements per padded row (1408 bytes)
constexpr int XN_ROWS = 232;           // padded rows per frame (3 + 224 + 3, + 2 so that a 13-row block never leaves the frame)
constexpr int XN_ROWB = XN_ROW * 2;
constexpr int W_ROWB = 464;            // LDS weight row of one output channel: 7 x 32 bf16 = 448 B, padded: conflict-free b128 reads

size_t stem_xn16_bytes(int F) { return (size_t)F * XN_ROWS * XN_ROWB; }

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// frames NCHW fp32 0..255 -> padded, normalised, channel-interleaved bf16 rows (the reference's (x/255 - mean)/std, then rounded)
__global__ __launch_bounds__(256) void stem_prep16_kernel(const float* __restrict__ x, bf16_t* __restrict__ xn16, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread per 8-element chunk
  if (i >= total) return;
  const int ch = (int)(i % (XN_ROW / 8));
  long long t = i / (XN_ROW / 8);
  const int r = (int)(t % XN_ROWS);
  const long long f = t / XN_ROWS;
  bf16x8 o;
  const int iy = r - 3;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int idx = ch * 8 + e - 9;
    float v = 0.f;
    if ((unsigned)iy < 224u && (unsigned)idx < 672u) {
      const int ix = idx / 3, c = idx - ix * 3;
      v = stem_normalize(x[((f * 3 + c) * 224 + iy) * 224 + ix], c);
    }
    o[e] = (bf16_t)v;
  }
  *reinterpret_cast<bf16x8*>(xn16 + i * 8) = o;
}

// the same image built from the RAW clips through their crop boxes (see stem_prep_crop_kernel in conv.hip)
template <typename T>
__global__ __launch_bounds__(256) void stem_prep16_crop_kernel(const T* __restrict__ raw, const int* __restrict__ boxes,
                                                                bf16_t* __restrict__ xn16, long long total, int Hi, int Wi, int fpb) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one thread per 8-element chunk
  if (i >= total) return;
  const int ch = (int)(i % (XN_ROW / 8));
  long long t = i / (XN_ROW / 8);
  const int r = (int)(t % XN_ROWS);
  const long long f = t / XN_ROWS;
  const int* b = boxes + (f / fpb) * 4;
  const int top = b[0], left = b[1], bh = b[2], bw = b[3];
  bf16x8 o;
  const int iy = r - 3;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int idx = ch * 8 + e - 9;
    float v = 0.f;
    if ((unsigned)iy < 224u && (unsigned)idx < 672u) {
      const int ix = idx / 3, c = idx - ix * 3;
      v = stem_normalize(bilinear_sample(raw + (f * 3 + c) * (long long)Hi * Wi, Wi, top, left, bh, bw, iy, ix, 0, 0, 224, 224), c);
    }
    o[e] = (bf16_t)v;
  }
  *reinterpret_cast<bf16x8*>(xn16 + i * 8) = o;
}

// x / d for a CONSTANT divisor, bit-identical to the IEEE quotient for d in {255, 0.229, 0.224, 0.225} and every float of
// magnitude 2^-40 .. 512 (and 0): q0 = x * RN(1/d), one exact-residual correction. Checked exhaustively on the CPU for these
// four divisors (tests/test_const_division.py); three operations instead of the ~12 of the generic expansion.
__device__ __forceinline__ float div_const(float x, float d, float rcp) {
#pragma clang fp contract(off)
  const float q = x * rcp;
  const float r = fmaf(-d, q, x);
  return fmaf(r, rcp, q);
}

// uint8 clips (what the loader ships): the same values as the kernel above, cheaper to produce. A thread owns 8 whole pixels (24
// elements, three 16-byte stores) so the horizontal sample position and weights are computed once per pixel instead of once
// per element; p / 255 comes from a 256-entry table (p is a byte) and the two remaining divisions by constants use div_const.
// Operation order per element is bilinear_sample + stem_normalize's (augment_dev.h); the GPU tests compare the two bit for bit.
// Round 6: the source bytes arrive as unaligned DWORD loads shared by the two taps of a pixel and — whenever the four taps of a
// pixel PAIR lie within four consecutive bytes (down-sampling factors up to 1.5: every box of 256-wide clips) — by both pixels:
// 24-48 loads per thread instead of 96 byte loads. The kernel was bound by exactly those (each byte load of a wave touches the
// same 5-9 cache lines its neighbours do: 1.0 ms for 2560 frames at 0.16 of the HBM roof). A load never leaves its source row:
// its address is clamped to the row's last four bytes and the byte positions shift accordingly.
constexpr int XN_GROUPS = (XN_ROW + 23) / 24;      // 30 groups of 24 elements; the last one holds 8 (padding) elements
__device__ __forceinline__ unsigned ld_u32_unaligned(const unsigned char* p) {
  unsigned w;
  __builtin_memcpy(&w, p, 4);
  return w;
}
__global__ __launch_bounds__(256) void stem_prep16_crop_u8_kernel(const unsigned char* __restrict__ raw, const int* __restrict__ boxes,
                                                                   bf16_t* __restrict__ xn16, long long total, int Hi, int Wi, int fpb) {
#pragma clang fp contract(off)
  __shared__ float lut[256];
  lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int g = (int)(i % XN_GROUPS);
  long long t = i / XN_GROUPS;
  const int r = (int)(t % XN_ROWS);
  const long long f = t / XN_ROWS;
  bf16_t* dst = xn16 + (f * XN_ROWS + r) * XN_ROW + g * 24;
  const int nst = g == XN_GROUPS - 1 ? (XN_ROW - 24 * (XN_GROUPS - 1)) / 8 : 3;
  const int iy = r - 3;
  bf16x8 o[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[q][e] = (bf16_t)0.f;
  if ((unsigned)iy < 224u && g * 8 - 3 < 224) {
    const int* b = boxes + (f / fpb) * 4;
    const int top = b[0], left = b[1], bh = b[2], bw = b[3];
    const float sy = fmaxf(fmaf((float)iy + 0.5f, (float)bh / 224.0f, -0.5f), 0.f);
    const int y0 = (int)sy;
    const int y1 = y0 + (y0 < bh - 1 ? 1 : 0);
    const float ly = sy - (float)y0, hy = 1.f - ly;
    const float xscale = (float)bw / 224.0f;
    const long long plane = (long long)Hi * Wi;
    const unsigned char* r0 = raw + f * 3 * plane + (long long)(top + y0) * Wi;      // source rows (column 0 of the clip, not of the box)
    const unsigned char* r1 = raw + f * 3 * plane + (long long)(top + y1) * Wi;
    const int amax = Wi - 4;                                  // last column a dword load may start at inside a row (launcher: Wi >= 4)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      // the two pixels of the pair: sample columns, weights, validity
      int x0[2], x1[2];
      float lx[2], hx[2];
      bool ok[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ix = g * 8 - 3 + 2 * kk + h;
        ok[h] = (unsigned)ix < 224u;
        const float sx = fmaxf(fmaf((float)(ok[h] ? ix : 0) + 0.5f, xscale, -0.5f), 0.f);
        x0[h] = (int)sx;
        x1[h] = x0[h] + (x0[h] < bw - 1 ? 1 : 0);
        lx[h] = sx - (float)x0[h];
        hx[h] = 1.f - lx[h];
      }
      if (!ok[0] && !ok[1]) continue;
      // dword A starts at pixel 0's left tap (clamped into the row); pixel 1 shares it when its right tap still lies inside
      const int ca = left + x0[0], cb = left + x0[1];
      const int aa = min(ca, amax);
      const bool share = (left + x1[1] - aa <= 3) && (cb >= aa);
      const int ab = share ? aa : min(cb, amax);
      const int s00 = 8 * (ca - aa), s01 = 8 * (left + x1[0] - aa);          // bit positions of pixel 0's taps in dword A
      const int s10 = 8 * (cb - ab), s11 = 8 * (left + x1[1] - ab);          // of pixel 1's taps in dword B (= A when shared)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned wa0 = ld_u32_unaligned(r0 + c * plane + aa);
        const unsigned wa1 = ld_u32_unaligned(r1 + c * plane + aa);
        const unsigned wb0 = share ? wa0 : ld_u32_unaligned(r0 + c * plane + ab);
        const unsigned wb1 = share ? wa1 : ld_u32_unaligned(r1 + c * plane + ab);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (!ok[h]) continue;
          const unsigned w0 = h ? wb0 : wa0, w1 = h ? wb1 : wa1;
          const int sl = h ? s10 : s00, sr = h ? s11 : s01;
          const float v00 = lut[(w0 >> sl) & 0xffu], v01 = lut[(w0 >> sr) & 0xffu];
          const float v10 = lut[(w1 >> sl) & 0xffu], v11 = lut[(w1 >> sr) & 0xffu];
          const float t0 = fmaf(hx[h], v00, lx[h] * v01);
          const float t1 = fmaf(hx[h], v10, lx[h] * v11);
          const float v = fmaf(hy, t0, ly * t1) * 255.0f;
          const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
          const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
          const float rsd = c == 0 ? 1.0f / 0.229f : (c == 1 ? 1.0f / 0.224f : 1.0f / 0.225f);
          const float n = div_const(div_const(v, 255.0f, 1.0f / 255.0f) - mean, sd, rsd);
          const int k = 2 * kk + h;
          o[(k * 3 + c) >> 3][(k * 3 + c) & 7] = (bf16_t)n;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (q < nst) *reinterpret_cast<bf16x8*>(dst + q * 8) = o[q];
}

int launch_stem_prep16_crop(const FrameSource& src, void* xn16, int F, hipStream_t s) {
  const long long total = (long long)F * XN_ROWS * (XN_ROW / 8);
  if (src.is_u8 && src.Wi >= 4) {     // (narrower clips — never a frame — take the generic kernel below: same values)
    const long long tg = (long long)F * XN_ROWS * XN_GROUPS;
    hipLaunchKernelGGL(stem_prep16_crop_u8_kernel, dim3(ceil_div(tg, 256)), dim3(256), 0, s, static_cast<const unsigned char*>(src.frames),
                       src.boxes, reinterpret_cast<bf16_t*>(xn16), tg, src.Hi, src.Wi, src.frames_per_box);
  } else if (src.is_u8)
    hipLaunchKernelGGL((stem_prep16_crop_kernel<unsigned char>), dim3(ceil_div(total, 256)), dim3(256), 0, s,
                       static_cast<const unsigned char*>(src.frames), src.boxes, reinterpret_cast<bf16_t*>(xn16), total, src.Hi, src.Wi,
                       src.frames_per_box);
  else
    hipLaunchKernelGGL((stem_prep16_crop_kernel<float>), dim3(ceil_div(total, 256)), dim3(256), 0, s,
                       static_cast<const float*>(src.frames), src.boxes, reinterpret_cast<bf16_t*>(xn16), total, src.Hi, src.Wi,
                       src.frames_per_box);
  return check_launch("stem_prep16_crop");
}

int launch_stem_prep16(const float* x_nchw, void* xn16, int F, hipStream_t s) {
  const long long total = (long long)F * XN_ROWS * (XN_ROW / 8);
  hipLaunchKernelGGL(stem_prep16_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, x_nchw, reinterpret_cast<bf16_t*>(xn16), total);
  return check_launch("stem_prep16");
}

// ---------------------------------------------------------------------------------------------------------
// forward: persistent blocks, tile = 256 output pixels x 64 channels, 4 waves x (64 pixels x 64 channels)
// ---------------------------------------------------------------------------------------------------------
constexpr int SF_REGION_A = 4 * 64 * 72 * 2;          // epilogue slabs (bf16, 4 waves x 64 rows x 72); the 13-row patch (18 304 B) aliases it
constexpr int SF_LDS = SF_REGION_A + 64 * W_ROWB;     // + weight image

template <int EPI>
__global__ __launch_bounds__(256) void stem_fwd16_kernel(const bf16_t* __restrict__ xn16, const float* __restrict__ w,
                                                          const GatherGemmParams p, int ntiles) {
  extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
  unsigned char* patch = smem;
  unsigned char* wl = smem + SF_REGION_A;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // weight image once per (persistent) block: wl[n][kh][j] = bf16(w[n][kh][j]) for j < 21, else 0
  for (int i = tid; i < 64 * 7 * 32; i += 256) {
    const int n = i / 224, rem = i - n * 224;
    const int kh = rem >> 5, j = rem & 31;
    const float v = (j < 21) ? w[n * 147 + kh * 21 + j] : 0.f;
    *reinterpret_cast<bf16_t*>(wl + n * W_ROWB + (kh * 32 + j) * 2) = (bf16_t)v;
  }
  const int lrow = lane & 31, lh = lane >> 5;
  int b_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) b_off[t] = (t * 32 + lrow) * W_ROWB + lh * 16;

  for (int blk = blockIdx.x; blk < ntiles; blk += gridDim.x) {
    const long long f = blk / 49;
    const int lm0 = (blk - (int)f * 49) * 256;    // first output pixel of this tile inside its frame (12544 = 49 * 256)
    const int oy0 = lm0 / 112;
    {   // the 13 padded rows 2*oy0 .. 2*oy0+12 are contiguous: 18 x 1 KiB direct-to-LDS copies, waves round-robin
      const char* src = reinterpret_cast<const char*>(xn16) + ((long long)f * XN_ROWS + 2 * oy0) * XN_ROWB + lane * 16;
      for (int q = wave; q < 18; q += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                         (__attribute__((address_space(3))) void*)(patch + q * 1024), 16, 0, 0);
    }
    int a_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int lm = lm0 + wave * 64 + t * 32 + lrow;
      const int oy = lm / 112, ox = lm - oy * 112;
      a_off[t] = 2 * (oy - oy0) * XN_ROWB + 12 * ox + 16 * lh;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned* ap = reinterpret_cast<const unsigned*>(patch + a_off[t] + kh * XN_ROWB + sidx * 32);   // 4-byte aligned only
          const u32x4v raw = {ap[0], ap[1], ap[2], ap[3]};
          a[t] = __builtin_bit_cast(bf16x8, raw);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const bf16x8*>(wl + b_off[t] + kh * 64 + sidx * 32);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
    __syncthreads();   // every wave is done with the patch before the slabs overwrite it
    if (EPI & EPI_STATS) gg_stats<256, 64, 4, 1>(p, acc, reinterpret_cast<float*>(smem), 0, blk);
    gg_store_bf16<256, 64, 4, 1, EPI & ~EPI_STATS, SF_REGION_A / 4>(p, acc, reinterpret_cast<float*>(smem), blk * 256, 0);
    __syncthreads();   // slabs drained before the next tile's patch lands
  }
}

int launch_stem_fwd16(const void* xn16, const float* w147, void* y, float* stats, int F, hipStream_t s) {
  GatherGemmParams p;
  memset(&p, 0, sizeof p);
  p.out = static_cast<float*>(y); p.stats = stats; p.dtype = DT_BF16;
  p.M = F * 12544; p.Nc = 64; p.os = 1;
  p.Hg = 112; p.Wg = 112; p.Ho = 112; p.Wo = 112;
  const double flops = 2.0 * (double)p.M * 64.0 * 147.0;
  prof_begin(KC_GEMM_NARROW, flops, p.M, 64, 147, 1, s);
  prof_bytes((double)stem_xn16_bytes(F) + (double)p.M * 64 * 2);
  const int ntiles = F * 49;
  const int grid = ntiles < 512 ? ntiles : 512;   // persistent blocks (2 per CU): the weight image is converted once per block
  static DynLdsOptIn optin_stats, optin_plain;
  if (int e = ensure_dyn_lds(optin_stats, reinterpret_cast<const void*>(stem_fwd16_kernel<EPI_STATS>), SF_LDS, "stem_fwd16")) return e;
  if (int e = ensure_dyn_lds(optin_plain, reinterpret_cast<const void*>(stem_fwd16_kernel<0>), SF_LDS, "stem_fwd16")) return e;
  const bf16_t* xn = static_cast<const bf16_t*>(xn16);
  if (stats) hipLaunchKernelGGL((stem_fwd16_kernel<EPI_STATS>), dim3(grid), dim3(256), SF_LDS, s, xn, w147, p, ntiles);
  else hipLaunchKernelGGL((stem_fwd16_kernel<0>), dim3(grid), dim3(256), SF_LDS, s, xn, w147, p, ntiles);
  prof_end(s);
  return check_launch("stem_fwd16");
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient: persistent blocks, one output image row (112 pixels = 7 K steps of 16) per iteration, double-buffered.
// GEMM per row: M = 64 channels (dY^T), N = 7 x 32 (kh, tap j; 21 of 32 real), K = 112 pixels.
// Waves 2 x 2: wm = channel half, wn = kernel rows {0..3} / {4..6}; accumulators live in registers across all rows of the block.
// ---------------------------------------------------------------------------------------------------------
constexpr int SW_DY = 112 * 128;                 // dY row image: [pixel][64 channels] bf16
constexpr int SW_PATCH = 10 * 1024;              // 7 padded rows = 9856 B, staged as 10 x 1 KiB
constexpr int SW_STAGE = SW_DY + SW_PATCH;       // 24 576 B
constexpr int STEM_WG16_BLOCKS = 768;

size_t stem_wgrad16_ws_floats() { return (size_t)STEM_WG16_BLOCKS * 64 * 224 + 64 * 224; }

typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void stem_wgrad16_kernel(const bf16_t* __restrict__ xn16, const bf16_t* __restrict__ dY,
                                                            float* __restrict__ partial, int total_rows) {
  __shared__ __attribute__((aligned(256))) unsigned char smem[2 * SW_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kh0 = wn * 4, ntile = wn ? 3 : 4;
  const int lrow = lane & 31, lh = lane >> 5;

  // DMA of one output row into ring slot `st`: 14 instructions of dY (8 pixels x 128 B each, 64-byte channel groups swizzled by
  // the pixel row so that the transpose reads are conflict-free) + 10 of the contiguous 7-row patch; 6 per wave
  auto issue_row = [&](int row, int st) __attribute__((always_inline)) {
    const long long f = row / 112;
    const int oy = row - (int)f * 112;
    unsigned char* dst = smem + st * SW_STAGE;
    const int key = ((lane >> 3) >> 1) & 1;
    const char* dsrc = reinterpret_cast<const char*>(dY) + (long long)row * SW_DY + (lane >> 3) * 128 + (((lane & 7) ^ (4 * key)) * 16);
    const char* psrc = reinterpret_cast<const char*>(xn16) + ((long long)f * XN_ROWS + 2 * oy) * XN_ROWB + lane * 16;
    for (int q = wave; q < 24; q += 4) {
      if (q < 14)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + q * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(psrc + (q - 14) * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + SW_DY + (q - 14) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // A (dY^T) transpose-read addressing, as wgrad_bf16_kernel<64, 64>: pixel row 8*(l>>5) + (q>>2) (+16s + 4r), channel group wm ^ key
  const int q16 = lane & 15;
  const int frow = 8 * (lane >> 5) + (q16 >> 2);
  const int fkey = (frow >> 1) & 1;
  const int a_off = frow * 128 + ((wm ^ fkey) * 64) + (16 * ((lane >> 4) & 1) + 4 * (q16 & 3)) * 2;
  // B (patch) addressing: element (kh, pixel k, tap j = lrow) at byte kh*1408 + 12*k + 2*j ; k = 16*s + 8*lh + e
  const int b_off = SW_DY + 96 * lh + 2 * lrow;

  int st = 0;
  int row = blockIdx.x;
  if (row < total_rows) issue_row(row, 0);
  for (; row < total_rows; row += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // this row has landed for everyone; the other slot is free
    const int nxt = row + gridDim.x;
    if (nxt < total_rows) issue_row(nxt, st ^ 1);
    const unsigned char* base = smem + st * SW_STAGE;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const s16x4v alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(base + a_off + (16 * s) * 128));
      const s16x4v ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(base + a_off + (16 * s + 4) * 128));
      const bf16x8 a = __builtin_bit_cast(bf16x8, (s16x8v)__builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < ntile) {
          const unsigned char* bp = base + b_off + (kh0 + t) * XN_ROWB + 192 * s;
          u32x4v raw;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned lo = *reinterpret_cast<const unsigned short*>(bp + 24 * e);
            const unsigned hi = *reinterpret_cast<const unsigned short*>(bp + 24 * e + 12);
            raw[e] = lo | (hi << 16);
          }
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, raw), acc[t], 0, 0, 0);
        }
      }
    }
    st ^= 1;
  }
  float* out = partial + (long long)blockIdx.x * 64 * 224;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < ntile) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[(n * 7 + kh0 + t) * 32 + lrow] = acc[t][r];
      }
    }
  }
}

// dw147[n][kh*21 + j] (+)= dw224[n][kh*32 + j]
__global__ void stem_unpack_dw32_kernel(const float* __restrict__ dw224, float* __restrict__ dw147, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 147) return;
  const int n = i / 147, k = i - n * 147;
  const int kh = k / 21, j = k - kh * 21;
  const float v = dw224[(n * 7 + kh) * 32 + j];
  dw147[i] = accumulate ? dw147[i] + v : v;
}

int launch_stem_wgrad16(const void* xn16, const void* dY, float* dw147, float* ws, int F, int accumulate, hipStream_t s) {
  const int total_rows = F * 112;
  const int nb = total_rows < STEM_WG16_BLOCKS ? total_rows : STEM_WG16_BLOCKS;
  const double flops = 2.0 * (double)F * 12544.0 * 64.0 * 147.0;
  prof_begin(KC_WGRAD_NARROW, flops, F * 12544, 64, 147, 1, s);
  prof_bytes((double)stem_xn16_bytes(F) + (double)F * 12544 * 64 * 2);
  hipLaunchKernelGGL(stem_wgrad16_kernel, dim3(nb), dim3(256), 0, s, static_cast<const bf16_t*>(xn16), static_cast<const bf16_t*>(dY), ws,
                     total_rows);
  prof_end(s);
  if (int e = check_launch("stem_wgrad16")) return e;
  float* dw224 = ws + (size_t)STEM_WG16_BLOCKS * 64 * 224;
  if (int e = launch_wgrad_reduce(ws, dw224, 64 * 224, nb, 0, s)) return e;
  hipLaunchKernelGGL(stem_unpack_dw32_kernel, dim3(ceil_div(64 * 147, 256)), dim3(256), 0, s, dw224, dw147, accumulate);
  return check_launch("stem_unpack_dw32");
}

}  // namespace r3m
