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

namespace r3m {

int conv_forward_launch(const float* X, const float* W, float* Y, float* stats, const float* bias, int N, int Hi, int Wi, int Ci,
                        int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s);
int conv_wgrad_launch(const float* X, const float* dY, float* dW, float* partial_ws, int N, int Hi, int Wi, int Ci, int Co, int k,
                      int stride, int pad, int accumulate, int dt, hipStream_t s);
size_t conv_wgrad_ws_floats(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int dt);

__device__ __forceinline__ f32x4 ld4g(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__device__ __forceinline__ int lang_bframe(int q) {
  // frame role of the second image of call q: eg=1, es0=2, es1=3, es2=4, e0=0
  if (q < 3) return q == 0 ? 1 : (q == 1 ? 3 : 4);
  if (q < 6) return q == 3 ? 0 : (q == 4 ? 2 : 3);
  const int j = (q - 6) % 3;
  return j == 0 ? 1 : (j == 1 ? 3 : 4);
}

// X[(q*B + i), :] = [ alle[src,0,:], alle[src,bf(q),:], feats[i,:] ],  src = q < 6 ? i : perm[q-6][i]
__global__ __launch_bounds__(256) void lang_gather_kernel(const float* __restrict__ alle, const float* __restrict__ feats,
                                                           const int* __restrict__ perm, float* __restrict__ X, int B, int D,
                                                           int LD) {
  const int row = blockIdx.x;
  const int q = row / B, i = row - q * B;
  const int src = q < 6 ? i : perm[(q - 6) * B + i];
  const int K1 = 2 * D + LD;
  float* x = X + (long long)row * K1;
  const float* a = alle + ((long long)src * 5 + 0) * D;
  const float* b = alle + ((long long)src * 5 + lang_bframe(q)) * D;
  const float* l = feats + (long long)i * LD;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    *reinterpret_cast<f32x4*>(x + d) = ld4g(a + d);
    *reinterpret_cast<f32x4*>(x + D + d) = ld4g(b + d);
  }
  for (int d = threadIdx.x * 4; d < LD; d += 1024) *reinterpret_cast<f32x4*>(x + 2 * D + d) = ld4g(l + d);
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// score[r] = H[r,:] . w + b     (one wave per row)
__global__ __launch_bounds__(256) void gemv_fwd_kernel(const float* __restrict__ H, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ score, int rows, int K) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const f32x4 h = ld4g(H + (long long)r * K + k), ww = ld4g(w + k);
    s += h[0] * ww[0] + h[1] * ww[1] + h[2] * ww[2] + h[3] * ww[3];
  }
  s = wsum(s);
  if (lane == 0) score[r] = s + b[0];
}

// dZ[r,k] = H[r,k] > 0 ? ds[r] * w[k] : 0    (gradient entering the last ReLU)
__global__ __launch_bounds__(256) void gemv_bwd_input_kernel(const float* __restrict__ ds, const float* __restrict__ H,
                                                              const float* __restrict__ w, float* __restrict__ dZ,
                                                              long long n4, int K4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const long long r = i / K4;
  const int k = (int)(i - r * K4) * 4;
  const float g = ds[r];
  const f32x4 h = ld4g(H + i * 4), ww = ld4g(w + k);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = h[e] > 0.f ? g * ww[e] : 0.f;
  *reinterpret_cast<f32x4*>(dZ + i * 4) = o;
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
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ A, const float* __restrict__ ds,
                                                              float* __restrict__ partial, int rows, int C, int rows_per_slice) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_slice;
  const int r1 = r0 + rows_per_slice < rows ? r0 + rows_per_slice : rows;
  float s = 0.f;
  if (c < C)
    for (int r = r0 + ty; r < r1; r += 4) s = fmaf(ds ? ds[r] : 1.f, A[(long long)r * C + c], s);
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
static int launch_colsum(const float* A, const float* ds, float* out, float* scratch, int rows, int C, int accumulate, hipStream_t s) {
  const int S = colsum_slices(rows);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(C, 64), S), dim3(256), 0, s, A, ds, scratch, rows, C, ceil_div(rows, S));
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
__global__ __launch_bounds__(256) void lang_scatter_kernel(const float* __restrict__ dX, const int* __restrict__ iperm,
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
      const float* x = dX + ((long long)q * B + rows[q]) * K1;
      if (f == 0) g += ld4g(x + d);                        // first image of every call is e0
      if (lang_bframe(q) == f) g += ld4g(x + D + d);       // second image
    }
    *reinterpret_cast<f32x4*>(out + d) = g;
  }
}

struct LangDims {
  int B, D, H, LD, K1, R;
  long long w[5], b[5];      // parameter offsets (floats)
  long long n_params;
  // workspace offsets (floats)
  long long X, Hh[4], dA, dB, Wt, wgp, cs, total;
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
  long long wg = conv_wgrad_ws_floats(d.R, 1, 1, d.K1, H, 1, 1, 0, DT_F32);
  const long long wg2 = conv_wgrad_ws_floats(d.R, 1, 1, H, H, 1, 1, 0, DT_F32);
  if (wg2 > wg) wg = wg2;
  d.wgp = take(wg);
  d.cs = take((long long)CS_MAX_SLICES * H);
  d.total = ws;
  return d;
}

static LangDims lang_dims(int B, int D, int H, int LD) { return lang_dims_rows(15 * B, B, D, H, LD); }

long long langrew_num_params(int D, int H, int LD) { return lang_dims(1, D, H, LD).n_params; }
size_t langrew_call_ws_floats(int R, int D, int H, int LD) { return (size_t)lang_dims_rows(R, 0, D, H, LD).total; }
size_t langrew_ws_floats(int B, int D, int H, int LD) { return (size_t)lang_dims(B, D, H, LD).total; }

static int check_dims(const LangDims& d) {
  R3M_REQUIRE(d.D % 32 == 0 && d.LD % 32 == 0 && d.H % 64 == 0, "langrew: D=%d, lang_dim=%d must be multiples of 32, hidden=%d of 64",
              d.D, d.LD, d.H);
  return 0;
}

// the MLP proper on the rows staged at ws + d.X: 4 x (Linear + ReLU) on the gather-GEMM, Linear(H -> 1) as a GEMV
static int mlp_forward(const LangDims& d, const float* params, float* scores, float* ws, hipStream_t s) {
  const int H = d.H;
  const float* in = ws + d.X;
  int K = d.K1;
  for (int l = 0; l < 4; ++l) {
    if (int e = conv_forward_launch(in, params + d.w[l], ws + d.Hh[l], nullptr, params + d.b[l], d.R, 1, 1, K, H, 1, 1, 0,
                                    EPI_BIAS | EPI_RELU, DT_F32, s))
      return e;
    in = ws + d.Hh[l];
    K = H;
  }
  hipLaunchKernelGGL(gemv_fwd_kernel, dim3(ceil_div(d.R, 4)), dim3(256), 0, s, in, params + d.w[4], params + d.b[4], scores, d.R, H);
  return check_launch("gemv_fwd");
}

int langrew_forward(const float* alle, const float* feats, const int* perm, const float* params, float* scores, float* ws, int B,
                    int D, int H, int LD, hipStream_t s) {
  const LangDims d = lang_dims(B, D, H, LD);
  if (int e = check_dims(d)) return e;
  hipLaunchKernelGGL(lang_gather_kernel, dim3(d.R), dim3(256), 0, s, alle, feats, perm, ws + d.X, B, D, LD);
  if (int e = check_launch("lang_gather")) return e;
  return mlp_forward(d, params, scores, ws, s);
}

int launch_transpose_w(const float* W, float* Wt, int Co, int T, int Ci, hipStream_t s);
int conv_dgrad_launch(const float* dY, const float* Wt, float* dX, const float* add0, const float* add1, const unsigned* addbits,
                      int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s);

// Needs the workspace left by the forward (X and the four hidden activations).
// dscore [R] -> parameter gradients (= or +=); returns the buffer holding dX [R, K1] (inside ws) through *dx_out
static int mlp_backward(const LangDims& d, const float* dscore, const float* params, float* grads, float* ws, int accumulate,
                        float** dx_out, hipStream_t s) {
  const int H = d.H;
  float* dA = ws + d.dA;
  float* dB = ws + d.dB;
  float* Wt = ws + d.Wt;
  float* wgp = ws + d.wgp;
  const float* H4 = ws + d.Hh[3];
  // last Linear(H -> 1)
  if (int e = launch_colsum(H4, dscore, grads + d.w[4], ws + d.cs, d.R, H, accumulate, s)) return e;
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, dscore, grads + d.b[4], d.R, accumulate);
  const long long n4 = (long long)d.R * H / 4;
  hipLaunchKernelGGL(gemv_bwd_input_kernel, dim3(ceil_div(n4, 256)), dim3(256), 0, s, dscore, H4, params + d.w[4], dA, n4, H / 4);
  if (int e = check_launch("lang_head_bwd")) return e;
  // hidden Linear layers 4..1:  dZ_l -> dW_l, db_l, dZ_{l-1} (ReLU mask of H_{l-1} fused into the dgrad epilogue).
  // Ping-pong dA -> dB -> dA -> dB -> dA: the last product, dX [15B, K1], lands in dA (the buffer sized for K1).
  float* dz = dA;
  float* nxt = dB;
  for (int l = 3; l >= 0; --l) {
    const float* in = l == 0 ? ws + d.X : ws + d.Hh[l - 1];
    const int K = l == 0 ? d.K1 : H;
    if (int e = conv_wgrad_launch(in, dz, grads + d.w[l], wgp, d.R, 1, 1, K, H, 1, 1, 0, accumulate, DT_F32, s)) return e;
    if (int e = launch_colsum(dz, nullptr, grads + d.b[l], ws + d.cs, d.R, H, accumulate, s)) return e;
    if (int e = launch_transpose_w(params + d.w[l], Wt, H, 1, K, s)) return e;
    if (int e = conv_dgrad_launch(dz, Wt, nxt, nullptr, l == 0 ? nullptr : in, nullptr, d.R, 1, 1, K, H, 1, 1, 0, l == 0 ? 0 : EPI_MASK_OUT, DT_F32, s))
      return e;
    float* t = dz; dz = nxt; nxt = t;
  }
  *dx_out = dz;   // dX [R, K1]
  return 0;
}

int langrew_backward(const float* dscore, const int* iperm, const float* params, float* grads, float* dalle, float* ws, int B, int D,
                     int H, int LD, int accumulate, hipStream_t s) {
  const LangDims d = lang_dims(B, D, H, LD);
  if (int e = check_dims(d)) return e;
  float* dz = nullptr;
  if (int e = mlp_backward(d, dscore, params, grads, ws, accumulate, &dz, s)) return e;
  if (dalle) {
    hipLaunchKernelGGL(lang_scatter_kernel, dim3(B, 5), dim3(256), 0, s, dz, iperm, dalle, B, D, LD);
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
  return mlp_forward(d, params, score, ws, s);
}

int langrew_call_backward(const float* dscore, const float* params, float* grads, float* de0, float* deg, float* dle, float* ws,
                          int R, int D, int H, int LD, int accumulate, hipStream_t s) {
  const LangDims d = lang_dims_rows(R, 0, D, H, LD);
  if (int e = check_dims(d)) return e;
  float* dz = nullptr;
  if (int e = mlp_backward(d, dscore, params, grads, ws, accumulate, &dz, s)) return e;
  if (de0 || deg || dle) {
    hipLaunchKernelGGL(lang_split_kernel, dim3(R), dim3(256), 0, s, dz, de0, deg, dle, D, LD);
    if (int e = check_launch("lang_split")) return e;
  }
  return 0;
}

}  // namespace r3m
