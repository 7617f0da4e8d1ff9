// r3m_amd — language-reward head G(e_a, e_b, l) of R3M, batched: the reference evaluates LanguageReward 15 times per step
// on [B, 2D+768] inputs (/root/reference/r3m/trainer.py:72-92 -> models_r3m.py:78-81 -> models_language.py:43-55); here
// the 15 calls are ONE [15B, 2D+768] MLP pass (4 x Linear+ReLU on the MFMA gather-GEMM, final Linear(1024->1) as a GEMV),
// forward and backward, with the permuted image pairs gathered on the fly and their gradients returned to alle through
// the inverse permutations (gather form: no atomics, fixed summation order).
//
// Call order (row block q of the batched input; text features are NOT permuted, trainer.py:88-92):
//   q0 (e0,eg) q1 (e0,es1) q2 (e0,es2) | q3 (e0,e0) q4 (e0,es0) q5 (e0,es1) | q6+3k+j: (e0[pi], other_j[pi]), pi = perm[3k+j],
//   other_j = (eg, es1, es2)[j]
// Parameter layout (flat, = state-dict order pred.{0,2,4,6,8}.{weight,bias}): W1[H][K1] b1[H] W2[H][H] b2 W3 b3 W4 b4 w5[H] b5[1]
#include "common.h"
#include "conv_dev.h"

namespace r3m {

int conv_forward_launch(const float* X, const float* W, float* Y, float* stats, const float* bias, int N, int Hi, int Wi, int Ci,
                        int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s);
int conv_wgrad_launch(const float* X, const float* dY, float* dW, float* partial_ws, int N, int Hi, int Wi, int Ci, int Co, int k,
                      int stride, int pad, int accumulate, int dt, hipStream_t s);
size_t conv_wgrad_ws_floats(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int dt);
int launch_convert_bf16(const float* src, void* dst, long long n, hipStream_t s);
int launch_transpose_w_bf16(const float* W, void* Wt, int Co, int T, int Ci, hipStream_t s);

__device__ __forceinline__ f32x4 ld4g(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ int lang_bframe(int q) {
  // frame role of the second image of call q: eg=1, es0=2, es1=3, es2=4, e0=0
  if (q < 3) return q == 0 ? 1 : (q == 1 ? 3 : 4);
  if (q < 6) return q == 3 ? 0 : (q == 4 ? 2 : 3);
  const int j = (q - 6) % 3;
  return j == 0 ? 1 : (j == 1 ? 3 : 4);
}

// X[(q*B + i), :] = [ alle[src,0,:], alle[src,bf(q),:], feats[i,:] ],  src = q < 6 ? i : perm[q-6][i]
template <class T>   // T = storage type of the MLP's tensors (float, or bf16_t for the mixed-precision head)
__global__ __launch_bounds__(256) void lang_gather_kernel(const float* __restrict__ alle, const float* __restrict__ feats,
                                                           const int* __restrict__ perm, T* __restrict__ X, int B, int D,
                                                           int LD) {
  const int row = blockIdx.x;
  const int q = row / B, i = row - q * B;
  const int src = q < 6 ? i : perm[(q - 6) * B + i];
  const int K1 = 2 * D + LD;
  T* x = X + (long long)row * K1;
  const float* a = alle + ((long long)src * 5 + 0) * D;
  const float* b = alle + ((long long)src * 5 + lang_bframe(q)) * D;
  const float* l = feats + (long long)i * LD;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    st4t(x + d, ld4g(a + d));
    st4t(x + D + d, ld4g(b + d));
  }
  for (int d = threadIdx.x * 4; d < LD; d += 1024) st4t(x + 2 * D + d, ld4g(l + d));
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// score[r] = H[r,:] . w + b     (one wave per row)
template <class T>
__global__ __launch_bounds__(256) void gemv_fwd_kernel(const T* __restrict__ H, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ score, int rows, int K) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const f32x4 h = ld4t(H + (long long)r * K + k), ww = ld4g(w + k);
    s += h[0] * ww[0] + h[1] * ww[1] + h[2] * ww[2] + h[3] * ww[3];
  }
  s = wsum(s);
  if (lane == 0) score[r] = s + b[0];
}

// dZ[r,k] = H[r,k] > 0 ? ds[r] * w[k] : 0    (gradient entering the last ReLU)
template <class T>
__global__ __launch_bounds__(256) void gemv_bwd_input_kernel(const float* __restrict__ ds, const T* __restrict__ H,
                                                              const float* __restrict__ w, T* __restrict__ dZ,
                                                              long long n4, int K4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long long r = i / K4;
  const int k = (int)(i - r * K4) * 4;
  const float g = ds[r];
  const f32x4 h = ld4t(H + i * 4), ww = ld4g(w + k);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = h[e] > 0.f ? g * ww[e] : 0.f;
  st4t(dZ + i * 4, o);
}

// column reductions over rows, fixed order: out[c] (+)= sum_r coef(r) * A[r,c]   (coef = ds[r] when ds != null, else 1)
// Two passes: grid (column groups of 64) x (row slices); 256 threads = 64 columns x 4 row lanes sum one slice into
// partial[slice][c]; the second pass adds the slices in order. (One block per column group walking every row was 290 us per
// call at 3840 rows x 1024 columns — 16 blocks on a 256-CU chip, 1.5 ms of the step for five bias gradients.)
constexpr int CS_MAX_SLICES = 64;
static inline int colsum_slices(int rows) {
  const int s = ceil_div(rows, 64);
  return s < 1 ? 1 : (s > CS_MAX_SLICES ? CS_MAX_SLICES : s);
}
template <class T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ A, const float* __restrict__ ds,
                                                              float* __restrict__ partial, int rows, int C, int rows_per_slice) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_slice;
  const int r1 = r0 + rows_per_slice < rows ? r0 + rows_per_slice : rows;
  float s = 0.f;
  if (c < C)
    for (int r = r0 + ty; r < r1; r += 4) s = fmaf(ds ? ds[r] : 1.f, (float)A[(long long)r * C + c], s);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) partial[(long long)blockIdx.y * C + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int slices, int C,
                                                            int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < slices; ++k) s += partial[(long long)k * C + c];
  out[c] = accumulate ? out[c] + s : s;
}
// scratch: CS_MAX_SLICES * C floats
template <class T>
static int launch_colsum(const T* A, const float* ds, float* out, float* scratch, int rows, int C, int accumulate, hipStream_t s) {
  const int S = colsum_slices(rows);
  hipLaunchKernelGGL((colsum_partial_kernel<T>), dim3(ceil_div(C, 64), S), dim3(256), 0, s, A, ds, scratch, rows, C, ceil_div(rows, S));
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, scratch, out, S, C, accumulate);
  return check_launch("colsum");
}

__global__ void sum_kernel(const float* __restrict__ v, float* __restrict__ out, int n, int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (red[0] + red[1]) + (red[2] + red[3]);
    out[0] = accumulate ? out[0] + s : s;
  }
}

// dalle[i,f,:] += sum over every row of dX that was gathered from alle[i,f,:]   (fixed order q = 0..14)
template <class T>
__global__ __launch_bounds__(256) void lang_scatter_kernel(const T* __restrict__ dX, const int* __restrict__ iperm,
                                                            float* __restrict__ dalle, int B, int D, int LD) {
  const int i = blockIdx.x, f = blockIdx.y;
  const long long K1 = 2LL * D + LD;
  __shared__ int rows[15];
  if (threadIdx.x < 15) {
    const int q = threadIdx.x;
    rows[q] = q < 6 ? i : iperm[(q - 6) * B + i];
  }
  __syncthreads();
  float* out = dalle + ((long long)i * 5 + f) * D;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    f32x4 g = ld4g(out + d);
#pragma unroll
    for (int q = 0; q < 15; ++q) {
      const T* x = dX + ((long long)q * B + rows[q]) * K1;
      if (f == 0) g += ld4t(x + d);                        // first image of every call is e0
      if (lang_bframe(q) == f) g += ld4t(x + D + d);       // second image
    }
    *reinterpret_cast<f32x4*>(out + d) = g;
  }
}

struct LangDims {
  int B, D, H, LD, K1, R;
  long long w[5], b[5];      // parameter offsets (floats)
  long long n_params;
  // workspace offsets (floats)
  long long X, Hh[4], dA, dB, Wt, wgp, cs, w16, total;
};

// R = rows of the MLP input: 15 B for the batched step (B clips), or the row count of ONE get_reward call (B = 0)
static LangDims lang_dims_rows(int R, int B, int D, int H, int LD) {
  LangDims d;
  d.B = B; d.D = D; d.H = H; d.LD = LD; d.K1 = 2 * D + LD; d.R = R;
  long long o = 0;
  for (int l = 0; l < 5; ++l) {
    const long long in = l == 0 ? d.K1 : H, out = l == 4 ? 1 : H;
    d.w[l] = o; o += in * out;
    d.b[l] = o; o += out;
  }
  d.n_params = o;
  long long ws = 0;
  auto take = [&](long long n) { long long r = ws; ws = (ws + n + 63) / 64 * 64; return r; };
  d.X = take((long long)d.R * d.K1);
  for (int l = 0; l < 4; ++l) d.Hh[l] = take((long long)d.R * H);
  d.dA = take((long long)d.R * (d.K1 > H ? d.K1 : H));
  d.dB = take((long long)d.R * H);
  d.Wt = take((long long)H * d.K1);
  long long wg = 0;
  for (int dt : {DT_F32, DT_BF16})
    for (int K : {d.K1, H}) {
      const long long v = (long long)conv_wgrad_ws_floats(d.R, 1, 1, K, H, 1, 1, 0, dt);
      if (v > wg) wg = v;
    }
  d.wgp = take(wg);
  d.cs = take((long long)CS_MAX_SLICES * H);
  d.w16 = take((d.n_params + 1) / 2);       // bf16 image of the weights (mixed-precision head)
  d.total = ws;
  return d;
}

static LangDims lang_dims(int B, int D, int H, int LD) { return lang_dims_rows(15 * B, B, D, H, LD); }

long long langrew_num_params(int D, int H, int LD) { return lang_dims(1, D, H, LD).n_params; }
size_t langrew_call_ws_floats(int R, int D, int H, int LD) { return (size_t)lang_dims_rows(R, 0, D, H, LD).total; }
size_t langrew_ws_floats(int B, int D, int H, int LD) { return (size_t)lang_dims(B, D, H, LD).total; }

static int check_dims(const LangDims& d, int dt = DT_F32) {
  R3M_REQUIRE(d.D % 32 == 0 && d.LD % 32 == 0 && d.H % 64 == 0, "langrew: D=%d, lang_dim=%d must be multiples of 32, hidden=%d of 64",
              d.D, d.LD, d.H);
  R3M_REQUIRE(dt == DT_F32 || dt == DT_BF16, "langrew: unknown dtype %d", dt);
  R3M_REQUIRE(dt == DT_F32 || d.LD % 64 == 0, "langrew(bf16): lang_dim=%d must be a multiple of 64", d.LD);
  return 0;
}

// ---- mixed-precision head (dt == DT_BF16; R3M(precision="bf16"): what torch.autocast(bfloat16) around the reference's
// get_reward calls would do) ----------------------------------------------------------------------------------------------
// X, the hidden activations and their gradients are stored bf16 (same workspace slots, half used); every Linear runs on the
// bf16 GEMM kernels with fp32 accumulation (forward: bf16 image of the fp32 master weights made per call; dgrad: bf16
// transposed image; weight gradients: bf16 operands -> fp32 gradients); biases, the final Linear(H -> 1), the scores, all
// parameter gradients and dalle stay fp32. The bf16 GEMM epilogues in use are the plain stores, so bias + ReLU and the ReLU
// mask of the backward are two small in-place passes over the [R, H] tensor (3840 x 1024: ~4 us each).
// Where this is NOT autocast: a hidden activation is rounded TWICE — the GEMM result to bf16, then bf16 + fp32 bias -> ReLU -> bf16
// — whereas autocast's Linear adds the bias in the fp32 accumulator before its single rounding (one extra half-ulp of bf16 on
// the pre-activation; the rounding model tests/test_gpu_lang.py checks against has exactly these two roundings). And only the
// BATCHED training pass runs in bf16: the single-call get_reward / eval path stays fp32, so scores of a precision="bf16" model
// differ between training and evaluation at bf16 level.
__global__ __launch_bounds__(256) void bias_relu16_kernel(bf16_t* __restrict__ Z, const float* __restrict__ bias, long long n8, int C8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int c = (int)(i % C8) * 8;
  const bf16x8 z = *reinterpret_cast<const bf16x8*>(Z + i * 8);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaxf((float)z[e] + bias[c + e], 0.f);
  *reinterpret_cast<bf16x8*>(Z + i * 8) = o;
}
// dA[r, k] = H[r, k] > 0 ? dA[r, k] : 0
__global__ __launch_bounds__(256) void relu_mask16_kernel(bf16_t* __restrict__ dA, const bf16_t* __restrict__ Hh, long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(dA + i * 8), h = *reinterpret_cast<const bf16x8*>(Hh + i * 8);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)h[e] > 0.f ? g[e] : (bf16_t)0.f;
  *reinterpret_cast<bf16x8*>(dA + i * 8) = o;
}

static inline bf16_t* as16(float* p) { return reinterpret_cast<bf16_t*>(p); }

// the MLP proper on the rows staged at ws + d.X: 4 x (Linear + ReLU) on the gather-GEMM, Linear(H -> 1) as a GEMV
static int mlp_forward(const LangDims& d, const float* params, float* scores, float* ws, int dt, hipStream_t s) {
  const int H = d.H;
  const float* in = ws + d.X;
  int K = d.K1;
  for (int l = 0; l < 4; ++l) {
    if (dt == DT_BF16) {
      bf16_t* w16 = as16(ws + d.w16) + d.w[l];
      if (int e = launch_convert_bf16(params + d.w[l], w16, (long long)K * H, s)) return e;
      if (int e = conv_forward_launch(in, reinterpret_cast<const float*>(w16), ws + d.Hh[l], nullptr, nullptr, d.R, 1, 1, K, H, 1, 1, 0, 0,
                                      DT_BF16, s))
        return e;
      const long long n8 = (long long)d.R * H / 8;
      hipLaunchKernelGGL(bias_relu16_kernel, dim3(ceil_div(n8, 256)), dim3(256), 0, s, as16(ws + d.Hh[l]), params + d.b[l], n8, H / 8);
      if (int e = check_launch("lang_bias_relu16")) return e;
    } else if (int e = conv_forward_launch(in, params + d.w[l], ws + d.Hh[l], nullptr, params + d.b[l], d.R, 1, 1, K, H, 1, 1, 0,
                                           EPI_BIAS | EPI_RELU, DT_F32, s))
      return e;
    in = ws + d.Hh[l];
    K = H;
  }
  if (dt == DT_BF16)
    hipLaunchKernelGGL((gemv_fwd_kernel<bf16_t>), dim3(ceil_div(d.R, 4)), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(in),
                       params + d.w[4], params + d.b[4], scores, d.R, H);
  else
    hipLaunchKernelGGL((gemv_fwd_kernel<float>), dim3(ceil_div(d.R, 4)), dim3(256), 0, s, in, params + d.w[4], params + d.b[4], scores, d.R, H);
  return check_launch("gemv_fwd");
}

int langrew_forward(const float* alle, const float* feats, const int* perm, const float* params, float* scores, float* ws, int B,
                    int D, int H, int LD, int dt, hipStream_t s) {
  const LangDims d = lang_dims(B, D, H, LD);
  if (int e = check_dims(d, dt)) return e;
  if (dt == DT_BF16)
    hipLaunchKernelGGL((lang_gather_kernel<bf16_t>), dim3(d.R), dim3(256), 0, s, alle, feats, perm, as16(ws + d.X), B, D, LD);
  else
    hipLaunchKernelGGL((lang_gather_kernel<float>), dim3(d.R), dim3(256), 0, s, alle, feats, perm, ws + d.X, B, D, LD);
  if (int e = check_launch("lang_gather")) return e;
  return mlp_forward(d, params, scores, ws, dt, s);
}

int launch_transpose_w(const float* W, float* Wt, int Co, int T, int Ci, hipStream_t s);
int conv_dgrad_launch(const float* dY, const float* Wt, float* dX, const float* add0, const float* add1, const unsigned* addbits,
                      int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s);

// Needs the workspace left by the forward (X and the four hidden activations).
// dscore [R] -> parameter gradients (= or +=); returns the buffer holding dX [R, K1] (inside ws) through *dx_out
static int mlp_backward(const LangDims& d, const float* dscore, const float* params, float* grads, float* ws, int accumulate,
                        float** dx_out, int dt, hipStream_t s) {
  const int H = d.H;
  const bool h16 = dt == DT_BF16;
  float* dA = ws + d.dA;
  float* dB = ws + d.dB;
  float* Wt = ws + d.Wt;
  float* wgp = ws + d.wgp;
  const float* H4 = ws + d.Hh[3];
  // last Linear(H -> 1)
  if (int e = h16 ? launch_colsum(reinterpret_cast<const bf16_t*>(H4), dscore, grads + d.w[4], ws + d.cs, d.R, H, accumulate, s)
                  : launch_colsum(H4, dscore, grads + d.w[4], ws + d.cs, d.R, H, accumulate, s))
    return e;
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, dscore, grads + d.b[4], d.R, accumulate);
  const long long n4 = (long long)d.R * H / 4;
  if (h16)
    hipLaunchKernelGGL((gemv_bwd_input_kernel<bf16_t>), dim3(ceil_div(n4, 256)), dim3(256), 0, s, dscore, reinterpret_cast<const bf16_t*>(H4),
                       params + d.w[4], as16(dA), n4, H / 4);
  else
    hipLaunchKernelGGL((gemv_bwd_input_kernel<float>), dim3(ceil_div(n4, 256)), dim3(256), 0, s, dscore, H4, params + d.w[4], dA, n4, H / 4);
  if (int e = check_launch("lang_head_bwd")) return e;
  // hidden Linear layers 4..1:  dZ_l -> dW_l, db_l, dZ_{l-1} (ReLU mask of H_{l-1}: fused into the fp32 dgrad epilogue, a separate
  // in-place pass for bf16). Ping-pong dA -> dB -> dA -> dB -> dA: the last product, dX [15B, K1], lands in dA (sized for K1).
  float* dz = dA;
  float* nxt = dB;
  for (int l = 3; l >= 0; --l) {
    const float* in = l == 0 ? ws + d.X : ws + d.Hh[l - 1];
    const int K = l == 0 ? d.K1 : H;
    if (int e = conv_wgrad_launch(in, dz, grads + d.w[l], wgp, d.R, 1, 1, K, H, 1, 1, 0, accumulate, dt, s)) return e;
    if (int e = h16 ? launch_colsum(reinterpret_cast<const bf16_t*>(dz), nullptr, grads + d.b[l], ws + d.cs, d.R, H, accumulate, s)
                    : launch_colsum(dz, nullptr, grads + d.b[l], ws + d.cs, d.R, H, accumulate, s))
      return e;
    if (h16) {
      if (int e = launch_transpose_w_bf16(params + d.w[l], Wt, H, 1, K, s)) return e;
      if (int e = conv_dgrad_launch(dz, Wt, nxt, nullptr, nullptr, nullptr, d.R, 1, 1, K, H, 1, 1, 0, 0, DT_BF16, s)) return e;
      if (l > 0) {
        const long long n8 = (long long)d.R * H / 8;
        hipLaunchKernelGGL(relu_mask16_kernel, dim3(ceil_div(n8, 256)), dim3(256), 0, s, as16(nxt), reinterpret_cast<const bf16_t*>(in), n8);
        if (int e = check_launch("lang_relu_mask16")) return e;
      }
    } else {
      if (int e = launch_transpose_w(params + d.w[l], Wt, H, 1, K, s)) return e;
      if (int e = conv_dgrad_launch(dz, Wt, nxt, nullptr, l == 0 ? nullptr : in, nullptr, d.R, 1, 1, K, H, 1, 1, 0, l == 0 ? 0 : EPI_MASK_OUT, DT_F32, s))
        return e;
    }
    float* t = dz; dz = nxt; nxt = t;
  }
  *dx_out = dz;   // dX [R, K1]
  return 0;
}

int langrew_backward(const float* dscore, const int* iperm, const float* params, float* grads, float* dalle, float* ws, int B, int D,
                     int H, int LD, int accumulate, int dt, hipStream_t s) {
  const LangDims d = lang_dims(B, D, H, LD);
  if (int e = check_dims(d, dt)) return e;
  float* dz = nullptr;
  if (int e = mlp_backward(d, dscore, params, grads, ws, accumulate, &dz, dt, s)) return e;
  if (dalle) {
    if (dt == DT_BF16)
      hipLaunchKernelGGL((lang_scatter_kernel<bf16_t>), dim3(B, 5), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(dz), iperm, dalle, B, D, LD);
    else
      hipLaunchKernelGGL((lang_scatter_kernel<float>), dim3(B, 5), dim3(256), 0, s, dz, iperm, dalle, B, D, LD);
    if (int e = check_launch("lang_scatter")) return e;
  }
  return 0;
}

// ---- ONE evaluation G(e0, eg, le), differentiable: the reference's own calling form (R3M.get_reward, models_r3m.py:78-81 ->
// LanguageReward.forward, models_language.py:53-55), 15 of them per step in trainer.py:72-92. Same MLP core as the batched
// pass; the input rows are the concatenation [e0 | eg | le] staged by one copy kernel, the input gradient is split back.
__global__ __launch_bounds__(256) void lang_concat_kernel(const float* __restrict__ e0, const float* __restrict__ eg,
                                                           const float* __restrict__ le, float* __restrict__ X, int D, int LD) {
  const long long row = blockIdx.x;
  float* x = X + row * (2LL * D + LD);
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    *reinterpret_cast<f32x4*>(x + d) = ld4g(e0 + row * D + d);
    *reinterpret_cast<f32x4*>(x + D + d) = ld4g(eg + row * D + d);
  }
  for (int d = threadIdx.x * 4; d < LD; d += 1024) *reinterpret_cast<f32x4*>(x + 2 * D + d) = ld4g(le + row * LD + d);
}

__global__ __launch_bounds__(256) void lang_split_kernel(const float* __restrict__ dX, float* __restrict__ de0, float* __restrict__ deg,
                                                          float* __restrict__ dle, int D, int LD) {
  const long long row = blockIdx.x;
  const float* x = dX + row * (2LL * D + LD);
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    if (de0) *reinterpret_cast<f32x4*>(de0 + row * D + d) = ld4g(x + d);
    if (deg) *reinterpret_cast<f32x4*>(deg + row * D + d) = ld4g(x + D + d);
  }
  if (dle)
    for (int d = threadIdx.x * 4; d < LD; d += 1024) *reinterpret_cast<f32x4*>(dle + row * LD + d) = ld4g(x + 2 * D + d);
}

int langrew_call_forward(const float* e0, const float* eg, const float* le, const float* params, float* score, float* ws, int R,
                         int D, int H, int LD, hipStream_t s) {
  const LangDims d = lang_dims_rows(R, 0, D, H, LD);
  if (int e = check_dims(d)) return e;
  hipLaunchKernelGGL(lang_concat_kernel, dim3(R), dim3(256), 0, s, e0, eg, le, ws + d.X, D, LD);
  if (int e = check_launch("lang_concat")) return e;
  return mlp_forward(d, params, score, ws, DT_F32, s);
}

int langrew_call_backward(const float* dscore, const float* params, float* grads, float* de0, float* deg, float* dle, float* ws,
                          int R, int D, int H, int LD, int accumulate, hipStream_t s) {
  const LangDims d = lang_dims_rows(R, 0, D, H, LD);
  if (int e = check_dims(d)) return e;
  float* dz = nullptr;
  if (int e = mlp_backward(d, dscore, params, grads, ws, accumulate, &dz, DT_F32, s)) return e;
  if (de0 || deg || dle) {
    hipLaunchKernelGGL(lang_split_kernel, dim3(R), dim3(256), 0, s, dz, de0, deg, dle, D, LD);
    if (int e = check_launch("lang_split")) return e;
  }
  return 0;
}

}  // namespace r3m
