// r3m_amd — the R3M objective on embeddings alle[B,5,D] (frame roles e0, eg, es0, es1, es2), forward + backward fused:
// LP norms, time-contrastive InfoNCE with permuted cross-clip negatives, and the language-alignment InfoNCE on the
// reward-head scores. Restates /root/reference/r3m/trainer.py:39-152 and R3M.sim (/root/reference/r3m/models/models_r3m.py:102-107):
//   * permutations are INPUTS (the reference draws them with torch.randperm on the CPU generator, trainer.py:87-91,136-137);
//     negatives are gathered through the permutation, their gradient returns through the inverse permutation (a gather,
//     so no atomics and a fixed summation order);
//   * InfoNCE keeps the reference's literal form -log(eps + e^{s+} / (eps + sum e^{s})), eps = 1e-8, no max-subtraction;
//   * ||a-b||_2 has sub-gradient 0 at a == b (as torch.linalg.norm does).
// The reference runs ~60 tiny latency-bound kernels plus ~10 .item() syncs here; this is 3 launches and one metrics array.
#include "common.h"

namespace r3m {

constexpr float LOSS_EPS = 1e-8f;
constexpr int NPAIR = 9;     // s02, s12, s01, neg0_{0,1,2}, neg2_{0,1,2}
constexpr int NCLIPM = 8;    // per-clip metric partials: sum||.||2, sum||.||1, sum||.||0, tcn_i, aligned_i, (3 spare)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem /* >= 4*NV floats */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) smem[wave * NV + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = smem[k] + smem[NV + k] + smem[2 * NV + k] + smem[3 * NV + k];
  __syncthreads();
}

struct TcnArgs {
  const float* alle;   // [B,5,D]
  const int* perm;     // [6][B]: order of trainer.py:136-137 -> (es0-perm, es2-perm) for k = 0,1,2
  const int* iperm;    // [6][B]: inverse permutations
  float* pair_stat;    // [B][9][4]: l2dist: {dist,-,-,-}; cosine: {dot, na, nb, -}
  float* coef;         // [B][9]: d full_loss / d sim
  float* rownorm;      // [B][5]
  float* clipm;        // [B][NCLIPM]
  float* dalle;        // [B,5,D]
  int B, D;
  int l2dist;
  float l2weight, l1weight, tcnweight;
};

// ---- pass A: one block per clip: all row norms, the 9 pair similarities, per-clip loss terms and dL/dsim ----
__global__ __launch_bounds__(256) void tcn_pairs_kernel(const TcnArgs a) {
  __shared__ float smem[4 * 27];
  __shared__ float sims_s[NPAIR];
  const int i = blockIdx.x;
  const int D = a.D;
  const float* base = a.alle + (long long)i * 5 * D;
  const float* r[5] = {base, base + D, base + 2 * D, base + 3 * D, base + 4 * D};
  const float* n0p[3];
  const float* n2p[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    n0p[k] = a.alle + ((long long)a.perm[(2 * k + 0) * a.B + i] * 5 + 2) * D;
    n2p[k] = a.alle + ((long long)a.perm[(2 * k + 1) * a.B + i] * 5 + 4) * D;
  }
  // --- row norms ---
  float nv[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) nv[k] = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
#pragma unroll
    for (int f = 0; f < 5; ++f) {
      const float e = r[f][d];
      nv[f * 3 + 0] = fmaf(e, e, nv[f * 3 + 0]);
      nv[f * 3 + 1] += fabsf(e);
      nv[f * 3 + 2] += (e != 0.f) ? 1.f : 0.f;
    }
  }
  block_sum<15>(nv, smem);
  // --- pair sums ---
  float pv[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) pv[k] = 0.f;
  if (a.tcnweight > 0.f) {
    for (int d = threadIdx.x; d < D; d += 256) {
      const float x0 = r[2][d], x1 = r[3][d], x2 = r[4][d];
      float pa[NPAIR], pb[NPAIR];
      pa[0] = x2; pb[0] = x0;
      pa[1] = x2; pb[1] = x1;
      pa[2] = x1; pb[2] = x0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        pa[3 + k] = x0; pb[3 + k] = n0p[k][d];
        pa[6 + k] = x2; pb[6 + k] = n2p[k][d];
      }
#pragma unroll
      for (int q = 0; q < NPAIR; ++q) {
        if (a.l2dist) {
          const float df = pa[q] - pb[q];
          pv[q * 3] = fmaf(df, df, pv[q * 3]);
        } else {
          pv[q * 3 + 0] = fmaf(pa[q], pb[q], pv[q * 3 + 0]);
          pv[q * 3 + 1] = fmaf(pa[q], pa[q], pv[q * 3 + 1]);
          pv[q * 3 + 2] = fmaf(pb[q], pb[q], pv[q * 3 + 2]);
        }
      }
    }
    block_sum<27>(pv, smem);
  }
  if (threadIdx.x == 0) {
    float s2 = 0.f, s1 = 0.f, s0 = 0.f;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
      const float n2 = sqrtf(nv[f * 3]);
      a.rownorm[i * 5 + f] = n2;
      s2 += n2; s1 += nv[f * 3 + 1]; s0 += nv[f * 3 + 2];
    }
    float* cm = a.clipm + (long long)i * NCLIPM;
    cm[0] = s2; cm[1] = s1; cm[2] = s0; cm[3] = 0.f; cm[4] = 0.f; cm[5] = 0.f; cm[6] = 0.f; cm[7] = 0.f;
    if (a.tcnweight > 0.f) {
      float sim[NPAIR];
#pragma unroll
      for (int q = 0; q < NPAIR; ++q) {
        float* ps = a.pair_stat + ((long long)i * NPAIR + q) * 4;
        if (a.l2dist) {
          const float dist = sqrtf(pv[q * 3]);
          ps[0] = dist; ps[1] = 0.f; ps[2] = 0.f; ps[3] = 0.f;
          sim[q] = -dist;
        } else {
          const float na = sqrtf(pv[q * 3 + 1]), nb = sqrtf(pv[q * 3 + 2]);
          ps[0] = pv[q * 3]; ps[1] = na; ps[2] = nb; ps[3] = 0.f;
          sim[q] = pv[q * 3] / (fmaxf(na, LOSS_EPS) * fmaxf(nb, LOSS_EPS));
        }
      }
      const float e02 = expf(sim[0]), e12 = expf(sim[1]), e01 = expf(sim[2]);
      const float en0[3] = {expf(sim[3]), expf(sim[4]), expf(sim[5])};
      const float en2[3] = {expf(sim[6]), expf(sim[7]), expf(sim[8])};
      const float En0 = (en0[0] + en0[1]) + en0[2];
      const float En2 = (en2[0] + en2[1]) + en2[2];
      const float den1 = ((LOSS_EPS + e02) + e12) + En2;
      const float den2 = ((LOSS_EPS + e01) + e02) + En0;
      const float p1 = e12 / den1, p2 = e01 / den2;
      const float L1 = -logf(LOSS_EPS + p1), L2 = -logf(LOSS_EPS + p2);
      cm[3] = (L1 + L2) / 2.0f;
      cm[4] = ((sim[0] < sim[1]) && (sim[2] > sim[0])) ? 1.f : 0.f;
      const float w = a.tcnweight / (2.0f * (float)a.B);
      const float g1 = -w / (LOSS_EPS + p1), g2 = -w / (LOSS_EPS + p2);
      float* cf = a.coef + (long long)i * NPAIR;
      cf[0] = g1 * (-p1 * e02 / den1) + g2 * (-p2 * e02 / den2);
      cf[1] = g1 * (p1 - p1 * e12 / den1);
      cf[2] = g2 * (p2 - p2 * e01 / den2);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        cf[3 + k] = g2 * (-p2 * en0[k] / den2);
        cf[6 + k] = g1 * (-p1 * en2[k] / den1);
      }
    }
  }
  (void)sims_s;
}

// d sim(a,b) / d(row) for the row playing `role` (0: a, 1: b); returns the two scalars (alpha, beta) such that
//   grad_row += c * (alpha * row + beta * partner)
__device__ __forceinline__ void sim_grad_coeffs(const float* ps, int l2dist, int role, float* alpha, float* beta) {
  if (l2dist) {
    const float dist = ps[0];
    const float k = dist > 0.f ? -1.0f / dist : 0.f;  // s = -dist ; ds/da = -(a-b)/dist ; ds/db = +(a-b)/dist
    *alpha = k;
    *beta = -k;
  } else {
    const float dot = ps[0];
    const float nr = role == 0 ? ps[1] : ps[2];   // this row's norm
    const float np = role == 0 ? ps[2] : ps[1];   // partner's norm
    const float nrc = fmaxf(nr, LOSS_EPS), npc = fmaxf(np, LOSS_EPS);
    *beta = 1.0f / (nrc * npc);
    *alpha = nr > LOSS_EPS ? -dot / (nrc * npc * nr * nr) : 0.f;
  }
}

// ---- pass B: one block per (clip, frame role): LP gradient + every similarity term that touches this row ----
__global__ __launch_bounds__(256) void tcn_grad_kernel(const TcnArgs a) {
  const int i = blockIdx.x, f = blockIdx.y;
  const int D = a.D, B = a.B;
  const float* row = a.alle + ((long long)i * 5 + f) * D;
  float* out = a.dalle + ((long long)i * 5 + f) * D;
  const float invF = 1.0f / (float)(5 * B);
  const float nrm = a.rownorm[i * 5 + f];
  const float k2 = nrm > 0.f ? a.l2weight * invF / nrm : 0.f;
  const float k1 = a.l1weight * invF;

  // term list: (owner clip of the pair, pair id, role of this row, partner row pointer)
  const float* partner[8];
  float calpha[8], cbeta[8];
  int nt = 0;
  auto add_term = [&](int owner, int q, int role, const float* prow) {
    float al, be;
    sim_grad_coeffs(a.pair_stat + ((long long)owner * NPAIR + q) * 4, a.l2dist, role, &al, &be);
    const float c = a.coef[(long long)owner * NPAIR + q];
    partner[nt] = prow; calpha[nt] = c * al; cbeta[nt] = c * be; ++nt;
  };
  if (a.tcnweight > 0.f) {
    const float* es0 = a.alle + ((long long)i * 5 + 2) * D;
    const float* es1 = a.alle + ((long long)i * 5 + 3) * D;
    const float* es2 = a.alle + ((long long)i * 5 + 4) * D;
    if (f == 2) {
      add_term(i, 0, 1, es2);
      add_term(i, 2, 1, es1);
      for (int k = 0; k < 3; ++k) {
        const int pj = a.perm[(2 * k) * B + i];
        add_term(i, 3 + k, 0, a.alle + ((long long)pj * 5 + 2) * D);
        const int ij = a.iperm[(2 * k) * B + i];
        add_term(ij, 3 + k, 1, a.alle + ((long long)ij * 5 + 2) * D);
      }
    } else if (f == 3) {
      add_term(i, 1, 1, es2);
      add_term(i, 2, 0, es0);
    } else if (f == 4) {
      add_term(i, 0, 0, es0);
      add_term(i, 1, 0, es1);
      for (int k = 0; k < 3; ++k) {
        const int pj = a.perm[(2 * k + 1) * B + i];
        add_term(i, 6 + k, 0, a.alle + ((long long)pj * 5 + 4) * D);
        const int ij = a.iperm[(2 * k + 1) * B + i];
        add_term(ij, 6 + k, 1, a.alle + ((long long)ij * 5 + 4) * D);
      }
    }
  }
  for (int d = threadIdx.x; d < D; d += 256) {
    const float e = row[d];
    float g = k2 * e + k1 * ((e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f));
    for (int t = 0; t < nt; ++t) g += calpha[t] * e + cbeta[t] * partner[t][d];
    out[d] = g;
  }
}

// ---- language InfoNCE on the 15 score vectors (trainer.py:72-117) ----
// scores [15][B]: q0..2 = pos1..3 ; q3..5 = in-clip negatives (e0e0, e0es0, e0es1) ; q6+3k+j = k-th permuted negative of
// head j (call order of the reference loop). Writes dscore [15][B] and per-clip partials {rew_i, acc1, acc2, acc3}.
__global__ __launch_bounds__(256) void lang_infonce_kernel(const float* __restrict__ scores, const float* __restrict__ mask,
                                                            float* __restrict__ dscore, float* __restrict__ clipl, int B,
                                                            float langweight) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  const float mk = mask[i];
  const float w = langweight * mk / (3.0f * (float)B);
  float rew = 0.f;
  for (int j = 0; j < 3; ++j) {
    const float pos = scores[(long long)j * B + i];
    const int qn[4] = {3 + j, 6 + j, 9 + j, 12 + j};
    float en[4], mx = -INFINITY, sum = 0.f;
    for (int k = 0; k < 4; ++k) {
      const float sn = scores[(long long)qn[k] * B + i];
      en[k] = expf(sn);
      mx = fmaxf(mx, sn);
    }
    sum = ((en[0] + en[1]) + en[2]) + en[3];
    const float ep = expf(pos);
    const float den = (LOSS_EPS + ep) + sum;
    const float p = ep / den;
    rew += -logf(LOSS_EPS + p);
    const float g = -w / (LOSS_EPS + p);
    dscore[(long long)j * B + i] = g * (p - p * ep / den);
    for (int k = 0; k < 4; ++k) dscore[(long long)qn[k] * B + i] = g * (-p * en[k] / den);
    clipl[(long long)i * 4 + 1 + j] = (mx < pos) ? 1.f : 0.f;
  }
  clipl[(long long)i * 4 + 0] = mk * (rew / 3.0f);
}

// ---- finalize: fixed-order sums over clips -> metrics[16] ----
// metrics: 0 l2loss, 1 l1loss, 2 l0loss, 3 tcnloss, 4 aligned, 5 rewloss, 6 rewacc1, 7 rewacc2, 8 rewacc3, 9 full_loss
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ clipm, const float* __restrict__ clipl,
                                                             float* __restrict__ metrics, int B, float l2w, float l1w,
                                                             float tcnw, float langw) {
  __shared__ float smem[4 * 9];
  float v[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) v[k] = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float* cm = clipm + (long long)i * NCLIPM;
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] += cm[k];
    if (clipl) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[5 + k] += clipl[(long long)i * 4 + k];
    }
  }
  block_sum<9>(v, smem);
  if (threadIdx.x == 0) {
    const float invF = 1.0f / (float)(5 * B), invB = 1.0f / (float)B;
    const float l2 = v[0] * invF, l1 = v[1] * invF, l0 = v[2] * invF;
    const float tcn = v[3] * invB, al = v[4] * invB;
    const float rew = v[5] * invB;
    metrics[0] = l2; metrics[1] = l1; metrics[2] = l0; metrics[3] = tcn; metrics[4] = al;
    metrics[5] = rew; metrics[6] = v[6] * invB; metrics[7] = v[7] * invB; metrics[8] = v[8] * invB;
    float full = l2w * l2;
    full += l1w * l1;
    if (langw > 0.f) full += langw * rew;
    if (tcnw > 0.f) full += tcnw * tcn;
    metrics[9] = full;
  }
}

size_t loss_workspace_floats(int B) { return (size_t)B * (NPAIR * 4 + NPAIR + 5 + NCLIPM + 4) + 64; }

int launch_tcn_lp_loss(const float* alle, const int* perm, const int* iperm, float* dalle, float* ws, int B, int D, int l2dist,
                       float l2w, float l1w, float tcnw, hipStream_t s) {
  R3M_REQUIRE(B >= 1 && D >= 1, "loss: B=%d D=%d", B, D);
  TcnArgs a;
  a.alle = alle; a.perm = perm; a.iperm = iperm; a.dalle = dalle;
  a.pair_stat = ws;
  a.coef = a.pair_stat + (size_t)B * NPAIR * 4;
  a.rownorm = a.coef + (size_t)B * NPAIR;
  a.clipm = a.rownorm + (size_t)B * 5;
  a.B = B; a.D = D; a.l2dist = l2dist; a.l2weight = l2w; a.l1weight = l1w; a.tcnweight = tcnw;
  hipLaunchKernelGGL(tcn_pairs_kernel, dim3(B), dim3(256), 0, s, a);
  if (int e = check_launch("tcn_pairs")) return e;
  if (dalle) {
    hipLaunchKernelGGL(tcn_grad_kernel, dim3(B, 5), dim3(256), 0, s, a);
    if (int e = check_launch("tcn_grad")) return e;
  }
  return 0;
}

float* loss_ws_clipm(float* ws, int B) { return ws + (size_t)B * (NPAIR * 4 + NPAIR + 5); }
float* loss_ws_clipl(float* ws, int B) { return ws + (size_t)B * (NPAIR * 4 + NPAIR + 5 + NCLIPM); }

int launch_lang_infonce(const float* scores, const float* mask, float* dscore, float* ws, int B, float langw, hipStream_t s) {
  hipLaunchKernelGGL(lang_infonce_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, s, scores, mask, dscore, loss_ws_clipl(ws, B), B, langw);
  return check_launch("lang_infonce");
}

int launch_loss_finalize(float* ws, int B, int have_lang, float* metrics, float l2w, float l1w, float tcnw, float langw,
                         hipStream_t s) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, loss_ws_clipm(ws, B),
                     have_lang ? loss_ws_clipl(ws, B) : (const float*)nullptr, metrics, B, l2w, l1w, tcnw, langw);
  return check_launch("loss_finalize");
}

}  // namespace r3m
