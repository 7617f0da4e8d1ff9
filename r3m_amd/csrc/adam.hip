// r3m_amd — fused Adam step over ONE flat fp32 parameter buffer (all encoder + language-head tensors are views into it),
// replacing the per-tensor python loop of torch.optim.Adam that the reference builds at
// /root/reference/r3m/models/models_r3m.py:76 and steps at /root/reference/r3m/trainer.py:156-158
// (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad). One HBM pass: read p,g,m,v - write p,m,v.
#include "common.h"

namespace r3m {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n4, float beta1, float beta2,
                                                    float one_minus_beta1, float one_minus_beta2, float neg_step_size,
                                                    float bc2_sqrt, float eps, float grad_scale) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n4; i += stride) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mm = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * grad_scale;
      // exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
      mm[e] = mm[e] + (gr - mm[e]) * one_minus_beta1;
      vv[e] = vv[e] * beta2 + (one_minus_beta2 * gr) * gr;
      // denom = sqrt(v)/sqrt(bias_correction2) + eps ; p.addcdiv_(m, denom, value=-lr/bias_correction1)
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pp[e] = pp[e] + (neg_step_size * mm[e]) / denom;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pp;
    *reinterpret_cast<f32x4*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
  (void)beta1;
}

int launch_adam(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                long long step, float grad_scale, hipStream_t s) {
  R3M_REQUIRE(n % 4 == 0, "adam: n=%lld must be a multiple of 4", n);
  R3M_REQUIRE(step >= 1, "adam: step=%lld must be >= 1", step);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float neg_step = (float)(-(lr / bc1));
  const float bc2s = (float)sqrt(bc2);
  const long long n4 = n / 4;
  int grid = ceil_div(n4, 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n4, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                     (float)(1.0 - beta2), neg_step, bc2s, (float)eps, grad_scale);
  return check_launch("adam");
}

// Fused SGD over a flat buffer, torch.optim.SGD semantics (momentum, dampening, weight decay, Nesterov):
//   g' = g*grad_scale + wd*p ; buf = (first step) g' : momentum*buf + (1-dampening)*g' ; d = nesterov ? g' + momentum*buf : buf ; p -= lr*d
// The reference trains with Adam only (models_r3m.py:76); BASELINE.json's north_star names "the SGD/Adam step", so the plain
// optimizer is provided on the same flat-buffer layout (one HBM pass: read p, g, buf - write p, buf).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n4,
                                                   float lr, float momentum, float one_minus_damp, float wd, int nesterov, int first,
                                                   float grad_scale) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n4; i += stride) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i * 4);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 bb = {0.f, 0.f, 0.f, 0.f};
    if (buf && !first) bb = *reinterpret_cast<f32x4*>(buf + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gg[e] * grad_scale;
      if (wd != 0.f) gr = gr + wd * pp[e];
      float d = gr;
      if (buf) {
        bb[e] = first ? gr : bb[e] * momentum + one_minus_damp * gr;
        d = nesterov ? gr + momentum * bb[e] : bb[e];
      }
      pp[e] = pp[e] - lr * d;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pp;
    if (buf) *reinterpret_cast<f32x4*>(buf + i * 4) = bb;
  }
}

int launch_sgd(float* p, const float* g, float* momentum_buf, long long n, double lr, double momentum, double dampening,
               double weight_decay, int nesterov, long long step, float grad_scale, hipStream_t s) {
  R3M_REQUIRE(n % 4 == 0, "sgd: n=%lld must be a multiple of 4", n);
  R3M_REQUIRE(step >= 1, "sgd: step=%lld must be >= 1", step);
  R3M_REQUIRE(momentum == 0.0 || momentum_buf, "sgd: momentum needs a momentum buffer");
  R3M_REQUIRE(!nesterov || (momentum > 0.0 && dampening == 0.0), "sgd: nesterov needs momentum > 0 and dampening = 0");
  const long long n4 = n / 4;
  int grid = ceil_div(n4, 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, s, p, g, momentum != 0.0 ? momentum_buf : nullptr, n4, (float)lr, (float)momentum,
                     (float)(1.0 - dampening), (float)weight_decay, nesterov, step == 1 ? 1 : 0, grad_scale);
  return check_launch("sgd");
}

}  // namespace r3m
