// r3m_amd — the ResNet-18/34/50 encoder engine: a native plan (layer table + HBM arena layout) and the forward /
// backward launch sequences, all on one caller-supplied HIP stream. This is the from-scratch counterpart of what the
// reference obtains from torchvision.models.resnet{18,34,50}(pretrained=False) with fc = Identity, driven by
// R3M.forward (/root/reference/r3m/models/models_r3m.py:44-52,62,84-100) and autograd's backward
// (/root/reference/r3m/trainer.py:157). Architecture restated from SURVEY.md Appendix A (torchvision 0.8.2 is not
// vendored): stem 7x7/2 + BN + ReLU + maxpool 3x3/2, BasicBlock [2,2,2,2] / [3,4,6,3] or Bottleneck v1.5 [3,4,6,3],
// global average pool, flatten.
//
// HBM layout: activations NHWC fp32 in ONE arena owned by the caller (offsets fixed at plan creation, sized for the
// frame count F); parameters and gradients are two flat fp32 buffers in torchvision parameter order with conv weights
// stored OHWI (= logical OIHW tensors with channels_last strides, so state-dict interchange needs no copy kernels).
#include "common.h"
#include "augment_dev.h"
#ifndef R3M_BN_PAIR_DEFAULT
#define R3M_BN_PAIR_DEFAULT 1     // A/B builds: tools/build_ab.sh none variant nopair -DR3M_BN_PAIR_DEFAULT=0
#endif
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace r3m {

struct ConvSpec {
  std::string name;      // e.g. "layer1.0.conv1" ; BatchNorm is name_bn
  std::string bn_name;
  int Ci, Co, k, stride, pad;
  int Hi, Wi, Ho, Wo;
  long long w_off, gamma_off, beta_off;   // flat parameter / gradient buffer (floats)
  long long rm_off, rv_off;               // flat running-statistics buffer (floats)
  long long Y_off;                        // arena: raw conv output [F,Ho,Wo,Co]
  long long Z_off;                        // arena: activated output, -1 when the block epilogue produces it
  long long coef_off;                     // arena: mean, invstd, scale, shift, c1, c2  (6*Co floats)
  int stats_rows;                         // row blocks of the forward GEMM (BatchNorm partials)
  long long wt_off = 0;                   // dgrad weight image [Ci][k*k][Co] inside the plan's Wt region (elements of the activation type)
};

struct BlockSpec {
  int conv[3];
  int nconv;
  int ds;                 // downsample conv index or -1
  long long in_off;       // arena offset of the block input
  long long out_off;      // arena offset of the block output
  long long mask_off;     // arena offset of the block output's 1-bit ReLU mask (rows*C/32 words)
  int Ho, Wo, Co;
  int stage;              // 0..3 = layer1..layer4
};

struct TensorInfo {
  std::string name;
  int kind;               // 0 conv weight, 1 bn weight, 2 bn bias, 3 running_mean, 4 running_var
  long long offset;       // floats, in the params buffer (kind 0-2) or the buffers buffer (kind 3-4)
  int shape[4];           // logical shape (conv: O, I, kh, kw)
  int ndim;
};

struct Plan {
  int size, F, D;
  int dtype = DT_F32;     // activation storage: fp32, or bf16 (bf16 conv operands, fp32 accumulation / statistics / gradients of weights)
  long long w16_off = 0;  // arena: bf16 image of the flat parameter buffer (DT_BF16 only)
  std::vector<ConvSpec> convs;
  std::vector<BlockSpec> blocks;
  std::vector<TensorInfo> tensors;
  long long n_params = 0, n_buffers = 0;
  long long stage_param_begin[5];  // params of stem+layer1 | layer2 | layer3 | layer4 boundaries (see stage_range)
  // arena offsets (floats)
  long long col_off, P0_off, amax_off, partial_off, acc_off, wt_off, wgp_off;
  long long ctr_off = 0;    // [convs][8] + [convs][4][8] unsigned: per-XCD tile queues of the persistent kernel — the forward launch of each conv, and the (up to four: stride-2 parity classes) backward launches
  long long G_off[5];     // gradient ping-pong buffers: D (block output grad), A0/A1 (dY, alternating), B, C
  long long E_off = -1;   // dY of a downsample block's downsample BatchNorm: written with the block's last BatchNorm backward (one pass
                          // for both, bn_backward_pair), read by the downsample dgrad / wgrad at the END of the block (-1: no such block)
  long long arena_floats = 0;
  long long gmax = 0;
  int last_training = 1;
  int gd = 0;             // which G buffer holds the running output-gradient between backward stages
  // side stream: wgrad(L) runs concurrently with dgrad(L) (both only need dY_L), filling each other's tile-quantisation tails
  hipStream_t side = nullptr;
  hipEvent_t ev_dy = nullptr, ev_wg[2] = {nullptr, nullptr}, ev_join = nullptr;
  bool wg_pending[2] = {false, false};
  int a_next = 0;         // which of A0/A1 the next dY goes to
  int roles[5] = {0, 1, 2, 3, 4};
  bool bnred_init = false;
  int fuse_bnred = 1;       // BatchNorm-backward partials from the producing dgrad's epilogue (EPI_BNRED); 0: stand-alone reduce pass
  int dout_fused_rows = 0;  // > 0: the dgrad that wrote the running output gradient also wrote the BatchNorm-backward partials of the
                            // block that consumes it next (EPI_BNRED): that many partial rows wait in the partial buffer
  int use_side = -1;
  int bn_pair = R3M_BN_PAIR_DEFAULT;   // the two tail BatchNorms of a downsample block share their backward passes (bn_backward_pair); 0: separate passes
  // Backward stages carry state from one r3m_resnet_backward call to the next (buffer roles, the running output gradient and — with
  // EPI_BNRED — BatchNorm partials waiting in the shared partial buffer for the NEXT block). They are only valid in the order
  // 0,1,2,3 after ONE forward: next_stage is what the following call must begin with (0 = a backward may (re)start, -1 = no
  // forward has run yet).
  int next_stage = -1;
  // dgrad weight images of all layers, rebuilt by ONE launch at the start of each backward (launch_transpose_w_all)
  std::vector<WtEntry> wt_tab;
  std::vector<int> wt_tile0;
  long long wt_elems = 0;
  WtEntry* d_wt_tab = nullptr;             // device copies, made at the first backward (plan creation needs no GPU)
  int* d_wt_tile0 = nullptr;
};

static long long align64(long long x) { return (x + 63) / 64 * 64; }

static int add_conv(Plan& P, const std::string& name, const std::string& bn, int Ci, int Co, int k, int stride, int pad,
                    int Hi, int Wi) {
  ConvSpec c;
  c.name = name; c.bn_name = bn;
  c.Ci = Ci; c.Co = Co; c.k = k; c.stride = stride; c.pad = pad; c.Hi = Hi; c.Wi = Wi;
  c.Ho = (Hi + 2 * pad - k) / stride + 1;
  c.Wo = (Wi + 2 * pad - k) / stride + 1;
  c.w_off = P.n_params;
  P.n_params += (long long)Co * Ci * k * k;
  c.gamma_off = P.n_params; P.n_params += Co;
  c.beta_off = P.n_params; P.n_params += Co;
  c.rm_off = P.n_buffers; P.n_buffers += Co;
  c.rv_off = P.n_buffers; P.n_buffers += Co;
  TensorInfo t;
  t.name = name + ".weight"; t.kind = 0; t.offset = c.w_off; t.ndim = 4;
  t.shape[0] = Co; t.shape[1] = Ci; t.shape[2] = k; t.shape[3] = k;
  P.tensors.push_back(t);
  t.ndim = 1; t.shape[0] = Co; t.shape[1] = t.shape[2] = t.shape[3] = 1;
  t.name = bn + ".weight"; t.kind = 1; t.offset = c.gamma_off; P.tensors.push_back(t);
  t.name = bn + ".bias"; t.kind = 2; t.offset = c.beta_off; P.tensors.push_back(t);
  t.name = bn + ".running_mean"; t.kind = 3; t.offset = c.rm_off; P.tensors.push_back(t);
  t.name = bn + ".running_var"; t.kind = 4; t.offset = c.rv_off; P.tensors.push_back(t);
  c.Y_off = c.Z_off = c.coef_off = -1;
  c.stats_rows = 0;
  P.convs.push_back(c);
  return (int)P.convs.size() - 1;
}

Plan* plan_create(int size, int F, int dtype) {
  if (size != 18 && size != 34 && size != 50) { set_last_error("resnet: unsupported size %d (18, 34, 50)", size); return nullptr; }
  if (dtype != DT_F32 && dtype != DT_BF16) { set_last_error("resnet: unsupported dtype %d (0 fp32, 1 bf16)", dtype); return nullptr; }
  if (F < 1) { set_last_error("resnet: F=%d must be >= 1", F); return nullptr; }
  Plan* Pp = new Plan();
  Plan& P = *Pp;
  P.size = size; P.F = F; P.dtype = dtype;
  P.fuse_bnred = dtype == DT_F32 ? 1 : 0;   // see side_init(): measured gain for fp32 plans, measured loss for bf16 plans
  const bool bottleneck = (size == 50);
  const int expansion = bottleneck ? 4 : 1;
  const int nblk[4] = {size == 18 ? 2 : 3, size == 18 ? 2 : 4, size == 18 ? 2 : 6, size == 18 ? 2 : 3};
  P.D = 512 * expansion;

  // ---- layer table in torchvision parameter order ----
  add_conv(P, "conv1", "bn1", 3, 64, 7, 2, 3, 224, 224);
  P.stage_param_begin[0] = 0;
  int inC = 64, H = 56;
  for (int L = 0; L < 4; ++L) {
    const int planes = 64 << L;
    if (L > 0) P.stage_param_begin[L] = P.n_params;
    for (int b = 0; b < nblk[L]; ++b) {
      const int stride = (b == 0 && L > 0) ? 2 : 1;
      char pre[64];
      snprintf(pre, sizeof pre, "layer%d.%d", L + 1, b);
      const std::string p(pre);
      BlockSpec B;
      B.stage = L; B.ds = -1;
      const int Hout = H / stride;
      if (bottleneck) {
        B.nconv = 3;
        B.conv[0] = add_conv(P, p + ".conv1", p + ".bn1", inC, planes, 1, 1, 0, H, H);
        B.conv[1] = add_conv(P, p + ".conv2", p + ".bn2", planes, planes, 3, stride, 1, H, H);
        B.conv[2] = add_conv(P, p + ".conv3", p + ".bn3", planes, planes * 4, 1, 1, 0, Hout, Hout);
      } else {
        B.nconv = 2;
        B.conv[0] = add_conv(P, p + ".conv1", p + ".bn1", inC, planes, 3, stride, 1, H, H);
        B.conv[1] = add_conv(P, p + ".conv2", p + ".bn2", planes, planes, 3, 1, 1, Hout, Hout);
        B.conv[2] = -1;
      }
      if (stride != 1 || inC != planes * expansion)
        B.ds = add_conv(P, p + ".downsample.0", p + ".downsample.1", inC, planes * expansion, 1, stride, 0, H, H);
      B.Ho = B.Wo = Hout; B.Co = planes * expansion;
      P.blocks.push_back(B);
      inC = planes * expansion; H = Hout;
    }
  }
  P.stage_param_begin[4] = P.n_params;

  // ---- arena layout ----
  long long off = 0;
  auto take = [&](long long n) { long long o = off; off = align64(off + n); return o; };
  const long long Fll = F;
  // arena offsets are in floats whatever the storage type; a bf16 tensor of n elements takes n/2 of them
  auto act = [&](long long n) { return dtype == DT_BF16 ? (n + 1) / 2 : n; };
  // private normalised copy of the input frames (the stem's weight gradient re-reads it in backward): fp32 channel-interleaved
  // rows, or for bf16 plans the padded bf16 image of stem_bf16.hip
  P.col_off = take(dtype == DT_BF16 ? (long long)((stem_xn16_bytes(F) + 3) / 4) : Fll * 3 * 224 * 224);
  long long gmax = 0, partial_max = 0, wmax = 0, wgp_max = 0;
  auto act_elems = [&](const ConvSpec& c) { return Fll * c.Ho * c.Wo * c.Co; };
  for (size_t i = 0; i < P.convs.size(); ++i) {
    ConvSpec& c = P.convs[i];
    c.Y_off = take(act(act_elems(c)));
    c.coef_off = take(6LL * c.Co);
    if (act(act_elems(c)) > gmax) gmax = act(act_elems(c));
    const int M = F * c.Ho * c.Wo;
    c.stats_rows = gather_gemm_grid_m(M, c.Co);
    long long pr = (long long)c.stats_rows * 2 * c.Co;
    if (pr > partial_max) partial_max = pr;
    pr = (long long)(i == 0 ? bn_bwd_pool_partial_rows(F, c.Ho, c.Wo, c.Co) : bn_bwd_partial_rows(M, c.Co, dtype)) * 2 * c.Co;
    if (pr > partial_max) partial_max = pr;
    pr *= 2;                                                    // two sets: the paired first pass of a downsample block's tail (bn_backward_pair)
    if (i > 0 && pr > partial_max) partial_max = pr;
    pr = ((long long)bnred_partial_rows(M) + 4) * 2 * c.Co;     // EPI_BNRED: one row per 64 result rows (+ one per stride-2 parity class)
    if (pr > partial_max) partial_max = pr;
    const long long welems = (long long)c.Co * c.k * c.k * c.Ci;
    if (welems > wmax) wmax = welems;
    if (i == 0) {
      const long long need = dtype == DT_BF16 ? (long long)stem_wgrad16_ws_floats() : (long long)stem_wgrad_ws_floats() + 64 * 160;
      if (need > wgp_max) wgp_max = need;
    } else {
      const int split = dtype == DT_BF16 ? wgrad_bf16_pick_split(M, c.Co, c.Ci, c.k * c.k) : wgrad_pick_split(M, c.Co, c.Ci, c.k * c.k);
      if (welems * split > wgp_max) wgp_max = welems * split;
    }
  }
  // stem: P0 (pooled) and the argmax bytes; the pre-pool activation Z0 is never materialised (bn.hip: fused stem tail)
  P.P0_off = take(act(Fll * 56 * 56 * 64));
  P.amax_off = take((Fll * 56 * 56 * 64 + 3) / 4);
  long long cur_in = P.P0_off;
  for (auto& B : P.blocks) {
    B.in_off = cur_in;
    for (int j = 0; j < B.nconv - 1; ++j) P.convs[B.conv[j]].Z_off = take(act(act_elems(P.convs[B.conv[j]])));
    B.out_off = take(act(Fll * B.Ho * B.Wo * B.Co));
    B.mask_off = take((Fll * B.Ho * B.Wo * B.Co + 31) / 32);
    cur_in = B.out_off;
  }
  P.partial_off = take(partial_max);
  P.acc_off = take(64LL * 2 * 2048 * 2);  // doubles: 64 slices x 2 x Cmax, in float units x2
  {   // one dgrad weight image per conv layer after the stem (the stem has no input gradient)
    int tiles = 0;
    long long we = 0;
    for (size_t i = 1; i < P.convs.size(); ++i) {
      ConvSpec& c = P.convs[i];
      c.wt_off = we;
      P.wt_tab.push_back(WtEntry{c.w_off, we, c.Co, c.k * c.k, c.Ci, 0});
      P.wt_tile0.push_back(tiles);
      tiles += ceil_div(c.Ci, 32) * ceil_div(c.Co, 32) * c.k * c.k;
      we += align64((long long)c.Co * c.k * c.k * c.Ci);
    }
    P.wt_tile0.push_back(tiles);
    P.wt_elems = we;
    P.wt_off = take(dtype == DT_BF16 ? (we + 1) / 2 : we);
  }
  if (dtype == DT_BF16) P.w16_off = take((P.n_params + 1) / 2);
  P.wgp_off = take(wgp_max);
  P.ctr_off = take(5LL * (long long)P.convs.size() * 8);
  P.gmax = gmax;
  for (int g = 0; g < 5; ++g) P.G_off[g] = take(gmax);
  {
    long long emax = 0;
    for (size_t bi = 0; bi < P.blocks.size(); ++bi)
      if (P.blocks[bi].ds >= 0) emax = std::max(emax, act(Fll * P.blocks[bi].Ho * P.blocks[bi].Wo * P.blocks[bi].Co));
    if (emax > 0) P.E_off = take(emax);
  }
  P.arena_floats = off;
  return Pp;
}

// ---------------------------------------------------------------------------------------------------------
struct Ctx {
  Plan& P;
  const float* params;
  float* grads;
  float* bufs;
  float* arena;
  hipStream_t s;
  int training;
  int accumulate;
  int dt;
};

static void fill_taps_fwd(GatherGemmParams& g, int k, int pad) {
  int t = 0;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) {
      g.dy[t] = (signed char)(kh - pad); g.dx[t] = (signed char)(kw - pad); g.wt[t] = (unsigned char)(kh * k + kw); ++t;
    }
  g.ntaps = t;
}

// X / W / Y (and dY / Wt / dX / add0 / add1, X / dY below) are fp32 tensors, or bf16 tensors behind float-typed pointers when dt == DT_BF16
int conv_forward_launch(const float* X, const float* W, float* Y, float* stats, const float* bias, int N, int Hi, int Wi, int Ci,
                        int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s) {
  GatherGemmParams g;
  memset(&g, 0, sizeof g);
  g.dtype = dt;
  const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
  g.A = X; g.B = W; g.out = Y; g.stats = stats; g.bias = bias;
  g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ci = Ci;
  g.Hg = Ho; g.Wg = Wo; g.Ho = Ho; g.Wo = Wo; g.Nc = Co;
  g.is = stride; g.os = 1; g.ooy = 0; g.oox = 0;
  g.M = N * Ho * Wo;
  g.T = k * k;
  fill_taps_fwd(g, k, pad);
  g.flags = flags;
  g.simple_rows = (k == 1 && stride == 1 && pad == 0) ? 1 : 0;
  return launch_gather_gemm(g, s);
}

// Inference forward (round 6): the convolution stores [relu]( acc * scale[co] + shift[co] [+ what `out` already holds] ) — eval-mode
// BatchNorm, the residual join and the ReLU in the conv's own store (flags: EPI_AFFINE [| EPI_ACCUM] [| EPI_RELU]).
static void fill_forward_params(GatherGemmParams& g, const float* X, const float* W, float* out, const float* scale, const float* shift,
                                int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int flags, int dt) {
  memset(&g, 0, sizeof g);
  g.dtype = dt;
  const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
  g.A = X; g.B = W; g.out = out; g.bn_scale = scale; g.bn_shift = shift;
  g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ci = Ci;
  g.Hg = Ho; g.Wg = Wo; g.Ho = Ho; g.Wo = Wo; g.Nc = Co;
  g.is = stride; g.os = 1; g.ooy = 0; g.oox = 0;
  g.M = N * Ho * Wo;
  g.T = k * k;
  fill_taps_fwd(g, k, pad);
  g.flags = flags;
  g.simple_rows = (k == 1 && stride == 1 && pad == 0) ? 1 : 0;
}
bool conv_forward_affine_fusable(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int flags, int dt) {
  GatherGemmParams g;
  static const float one = 1.f;   // (the query looks at shapes, flags and whether the coefficient pointers are set — never dereferenced)
  fill_forward_params(g, nullptr, nullptr, nullptr, &one, &one, N, Hi, Wi, Ci, Co, k, stride, pad, flags, dt);
  return gather_gemm_fuses_affine(g);
}
int conv_forward_launch_affine(const float* X, const float* W, float* out, const float* scale, const float* shift, int N, int Hi, int Wi,
                               int Ci, int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s) {
  GatherGemmParams g;
  fill_forward_params(g, X, W, out, scale, shift, N, Hi, Wi, Ci, Co, k, stride, pad, flags, dt);
  return launch_gather_gemm(g, s);
}

// dX[N,Hi,Wi,Ci] = dgrad of conv(k, stride, pad) given dY[N,Ho,Wo,Co] and Wt[Ci][k*k][Co]
int conv_dgrad_launch(const float* dY, const float* Wt, float* dX, const float* add0, const float* add1, const unsigned* addbits,
                      int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s) {
  return conv_dgrad_launch_br(dY, Wt, dX, add0, add1, addbits, N, Hi, Wi, Ci, Co, k, stride, pad, flags, dt, nullptr, s);
}
int conv_dgrad_launch_br(const float* dY, const float* Wt, float* dX, const float* add0, const float* add1,
                         const unsigned* addbits, int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int flags,
                         int dt, BnRedArgs* br, hipStream_t s) {
  const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
  R3M_REQUIRE(stride == 1 || stride == 2, "dgrad: stride %d", stride);
  GatherGemmParams g;
  memset(&g, 0, sizeof g);
  g.dtype = dt;
  if (br) {
    flags |= EPI_BNRED;
    g.bn_y = br->Y; g.bn_bits = br->bits; g.bn_scale = br->scale; g.bn_shift = br->shift; g.bn_mean = br->mean;
    g.stats = br->partial;
    br->rows_out = 0;
  }
  g.A = dY; g.B = Wt; g.out = dX; g.add0 = add0; g.add1 = add1; g.addbits = addbits;
  g.N = N; g.Hi = Ho; g.Wi = Wo; g.Ci = Co;   // the GEMM "input" is dY
  g.Ho = Hi; g.Wo = Wi; g.Nc = Ci;
  g.is = 1; g.T = k * k; g.flags = flags;
  if (stride == 1) {
    g.Hg = Hi; g.Wg = Wi; g.os = 1; g.ooy = g.oox = 0;
    g.M = N * Hi * Wi;
    int t = 0;
    for (int kh = 0; kh < k; ++kh)
      for (int kw = 0; kw < k; ++kw) {
        g.dy[t] = (signed char)(pad - kh); g.dx[t] = (signed char)(pad - kw); g.wt[t] = (unsigned char)(kh * k + kw); ++t;
      }
    g.ntaps = t;
    g.simple_rows = (k == 1 && pad == 0) ? 1 : 0;
    if (br) br->rows_out = bnred_partial_rows(g.M);
    return launch_gather_gemm(g, s);
  }
  // stride 2: one launch per output parity class; class (py,px) only sees taps with (py+pad-kh), (px+pad-kw) even
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      GatherGemmParams c = g;
      c.Hg = (Hi - py + 1) / 2; c.Wg = (Wi - px + 1) / 2;
      if (c.Hg <= 0 || c.Wg <= 0) continue;
      c.os = 2; c.ooy = py; c.oox = px;
      c.M = N * c.Hg * c.Wg;
      int t = 0;
      for (int kh = 0; kh < k; ++kh) {
        if ((py + pad - kh) & 1) continue;
        for (int kw = 0; kw < k; ++kw) {
          if ((px + pad - kw) & 1) continue;
          c.dy[t] = (signed char)((py + pad - kh) / 2); c.dx[t] = (signed char)((px + pad - kw) / 2);
          c.wt[t] = (unsigned char)(kh * k + kw); ++t;
        }
      }
      c.ntaps = t;
      c.simple_rows = 0;
      R3M_REQUIRE(!(br && t == 0), "dgrad: EPI_BNRED on a parity class without taps (1x1 stride-2) is not supported");
      if (t == 0 && (flags & EPI_ACCUM) && !(flags & EPI_MASKED_ADD)) continue;  // nothing to add
      if (br) {                                          // every parity class appends its own partial rows
        c.stats = br->partial + (long long)br->rows_out * 2 * c.Nc;
        br->rows_out += bnred_partial_rows(c.M);
      }
      if (int e = launch_gather_gemm(c, s)) return e;
    }
  return 0;
}

int conv_wgrad_launch(const float* X, const float* dY, float* dW, float* partial_ws, int N, int Hi, int Wi, int Ci, int Co, int k,
                      int stride, int pad, int accumulate, int dt, hipStream_t s) {
  WgradParams w;
  memset(&w, 0, sizeof w);
  w.dtype = dt;
  w.Ho = (Hi + 2 * pad - k) / stride + 1; w.Wo = (Wi + 2 * pad - k) / stride + 1;
  w.dY = dY; w.X = X; w.out = partial_ws;
  w.N = N; w.Hi = Hi; w.Wi = Wi; w.Ci = Ci; w.Co = Co;
  w.KH = w.KW = k; w.stride = stride; w.pad = pad;
  w.M = N * w.Ho * w.Wo;
  w.simple_rows = (k == 1 && stride == 1 && pad == 0) ? 1 : 0;
  if (dt == DT_BF16) {
    const int split = wgrad_bf16_pick_split(w.M, Co, Ci, k * k);
    if (int e = launch_wgrad_bf16(w, split, s)) return e;
    return launch_wgrad_reduce(partial_ws, dW, (long long)Co * k * k * Ci, split, accumulate, s);
  }
  const int split = wgrad_pick_split(w.M, Co, Ci, k * k);
  if (int e = launch_wgrad(w, split, s)) return e;
  return launch_wgrad_reduce(partial_ws, dW, (long long)Co * k * k * Ci, split, accumulate, s);
}

size_t conv_wgrad_ws_floats(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int dt) {
  const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
  const int split = dt == DT_BF16 ? wgrad_bf16_pick_split(N * Ho * Wo, Co, Ci, k * k) : wgrad_pick_split(N * Ho * Wo, Co, Ci, k * k);
  return (size_t)split * Co * k * k * Ci;
}

static int g_fused_inference = 1;   // r3m_debug_set_fused_inference: 0 = inference forwards run the unfused eval sequence (A/B, tests)
int engine_set_fused_inference(int on) { const int old = g_fused_inference; g_fused_inference = on ? 1 : 0; return old; }

#define TRY(x)              \
  do {                      \
    if (int e_ = (x)) return e_; \
  } while (0)

static float* coef(Ctx& c, const ConvSpec& L, int which) { return c.arena + L.coef_off + (long long)which * L.Co; }

// conv -> (training: batch statistics -> coefficients | eval: running statistics -> coefficients)
static int conv_bn_coeffs(Ctx& c, const ConvSpec& L, const float* X, const float* W, int Ci_eff, int k_eff, int stride_eff,
                          int pad_eff, int Hi_eff, int Wi_eff, int N_eff) {
  Plan& P = c.P;
  float* Y = c.arena + L.Y_off;
  float* partial = c.arena + P.partial_off;
  double* acc = reinterpret_cast<double*>(c.arena + P.acc_off);
  gg_set_tile_counters(reinterpret_cast<unsigned*>(c.arena + P.ctr_off) + (&L - P.convs.data()) * 8);
  {
    // the counters are handed to "the next launch of this thread": if the launcher returns before it reaches launch_gather_gemm
    // (a failed requirement), nothing later on this thread may inherit them
    struct DropCounters { ~DropCounters() { gg_set_tile_counters(nullptr, 0); } } drop;
    TRY(conv_forward_launch(X, W, Y, partial, nullptr, N_eff, Hi_eff, Wi_eff, Ci_eff, L.Co, k_eff, stride_eff, pad_eff,
                            c.training ? EPI_STATS : 0, c.dt, c.s));
  }
  const float* gamma = c.params + L.gamma_off;
  const float* beta = c.params + L.beta_off;
  if (c.training) {
    const long long count = (long long)P.F * L.Ho * L.Wo;
    TRY(launch_bn_stats_reduce(partial, L.stats_rows, L.Co, acc, c.s));
    TRY(launch_bn_finalize_rows(acc, L.stats_rows, count, gamma, beta, c.bufs + L.rm_off, c.bufs + L.rv_off, 0.1f, 1e-5f,
                                coef(c, L, 0), coef(c, L, 1), coef(c, L, 2), coef(c, L, 3), L.Co, c.s));
  } else {
    TRY(launch_bn_eval_coeffs(gamma, beta, c.bufs + L.rm_off, c.bufs + L.rv_off, 1e-5f, coef(c, L, 0), coef(c, L, 1),
                              coef(c, L, 2), coef(c, L, 3), L.Co, c.s));
  }
  return 0;
}

// forward weight operand of layer L: the fp32 master, or its slice of the bf16 image made at the top of plan_forward
static const float* fwd_weights(Ctx& c, const ConvSpec& L) {
  if (c.dt != DT_BF16) return c.params + L.w_off;
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(c.arena + c.P.w16_off) + L.w_off * 2);
}

static int conv_bn(Ctx& c, const ConvSpec& L, const float* X) {
  return conv_bn_coeffs(c, L, X, fwd_weights(c, L), L.Ci, L.k, L.stride, L.pad, L.Hi, L.Wi, c.P.F);
}

int plan_forward_src(Plan& P, const float* x_nchw, const FrameSource* crop, const float* params, float* bufs, float* arena,
                     float* h_out, int training, hipStream_t s);
int plan_forward(Plan& P, const float* x_nchw, const float* params, float* bufs, float* arena, float* h_out, int training,
                 hipStream_t s) {
  return plan_forward_src(P, x_nchw, nullptr, params, bufs, arena, h_out, training, s);
}

// frames come either as [F,3,224,224] fp32 0..255 (x_nchw) or as raw clips + crop boxes (crop: rc / rctraj resampled inside the
// stem pre-pass)
int plan_forward_src(Plan& P, const float* x_nchw, const FrameSource* crop, const float* params, float* bufs, float* arena,
                     float* h_out, int training, hipStream_t s) {
  // training: 1 = batch statistics (+ running-statistics update), 0 = running statistics with everything a backward needs kept,
  // 2 = INFERENCE (round 6): running statistics, nothing kept — BatchNorm, residual join and ReLU ride in the convolutions' stores
  const bool infer = training == 2;
  if (infer) training = 0;
  Ctx c{P, params, nullptr, bufs, arena, s, training, 0, P.dtype};
  P.last_training = training;
  P.next_stage = infer ? -3 : 0;   // a new forward invalidates whatever an unfinished backward left behind (-3: nothing to differentiate)
  P.dout_fused_rows = 0;
  const int F = P.F;
  const int dt = P.dtype;
  // tile queues of this pass's persistent-kernel launches (conv_pw.hip): 8 counters per conv, zeroed here, each used by one launch
  if (hipMemsetAsync(arena + P.ctr_off, 0, P.convs.size() * 8 * sizeof(unsigned), s) != hipSuccess) {
    set_last_error("resnet_forward: cannot reset the tile queues");
    return 1;
  }
  if (dt == DT_BF16) TRY(launch_convert_bf16(params, arena + P.w16_off, P.n_params, s));   // bf16 image of every weight (45 MB for ResNet-50)
  // ---- stem: x/255 -> Normalize -> conv1 7x7/2 straight from the NCHW frames (csrc/conv.hip stem_fwd_kernel) ----
  const ConvSpec& L0 = P.convs[0];
  // normalised, channel-interleaved copy of the frames (0.6 MB/frame): read by the stem forward now and by its weight
  // gradient in backward (the caller's tensor may be gone by then)
  if (crop) {
    if (dt == DT_BF16) TRY(launch_stem_prep16_crop(*crop, arena + P.col_off, F, s));
    else TRY(launch_stem_prep_crop(*crop, arena + P.col_off, F, s));
  } else if (dt == DT_BF16) {
    TRY(launch_stem_prep16(x_nchw, arena + P.col_off, F, s));
  } else {
    TRY(launch_stem_prep(x_nchw, arena + P.col_off, F, s));
  }
  {
    float* partial = arena + P.partial_off;
    double* acc = reinterpret_cast<double*>(arena + P.acc_off);
    if (dt == DT_BF16) TRY(launch_stem_fwd16(arena + P.col_off, params + L0.w_off, arena + L0.Y_off, training ? partial : nullptr, F, s));
    else TRY(launch_stem_fwd(arena + P.col_off, params + L0.w_off, arena + L0.Y_off, training ? partial : nullptr, F, dt, s));
    if (training) {
      TRY(launch_bn_stats_reduce(partial, L0.stats_rows, 64, acc, s));
      TRY(launch_bn_finalize_rows(acc, L0.stats_rows, (long long)F * 12544, params + L0.gamma_off, params + L0.beta_off,
                                  bufs + L0.rm_off, bufs + L0.rv_off, 0.1f, 1e-5f, coef(c, L0, 0), coef(c, L0, 1), coef(c, L0, 2),
                                  coef(c, L0, 3), 64, s));
    } else {
      TRY(launch_bn_eval_coeffs(params + L0.gamma_off, params + L0.beta_off, bufs + L0.rm_off, bufs + L0.rv_off, 1e-5f,
                                coef(c, L0, 0), coef(c, L0, 1), coef(c, L0, 2), coef(c, L0, 3), 64, s));
    }
  }
  // BatchNorm + ReLU + MaxPool in one pass over Y0
  TRY(launch_bn_relu_maxpool_fwd(arena + L0.Y_off, coef(c, L0, 2), coef(c, L0, 3), arena + P.P0_off,
                                 reinterpret_cast<unsigned char*>(arena + P.amax_off), F, 112, 112, 64, dt, s));
  // ---- residual stages ----
  if (infer && g_fused_inference) {
    // Inference: per block, every convolution stores its activated output itself — inner convs relu(bn(conv)), the downsample conv
    // bn(conv), the last conv relu(bn(conv) + residual) accumulating ONTO the residual (the block input, in place, or the downsample
    // result): no raw conv output, no bn_act_fwd pass, no mask bits. A block any of whose launches runs a kernel without these
    // epilogues (odd shapes; never a ResNet layer) takes the unfused sequence below instead.
    const float* cur_in = arena + P.blocks[0].in_off;
    for (const BlockSpec& B : P.blocks) {
      const int nlast = B.conv[B.nconv - 1];
      bool fus = true;
      for (int j = 0; j < B.nconv; ++j) {
        const ConvSpec& L = P.convs[B.conv[j]];
        const int fl = j < B.nconv - 1 ? (EPI_AFFINE | EPI_RELU) : (EPI_AFFINE | EPI_ACCUM | EPI_RELU);
        fus = fus && conv_forward_affine_fusable(F, L.Hi, L.Wi, L.Ci, L.Co, L.k, L.stride, L.pad, fl, dt);
      }
      if (B.ds >= 0) {
        const ConvSpec& Ld = P.convs[B.ds];
        fus = fus && conv_forward_affine_fusable(F, Ld.Hi, Ld.Wi, Ld.Ci, Ld.Co, Ld.k, Ld.stride, Ld.pad, EPI_AFFINE, dt);
      }
      auto coeffs = [&](const ConvSpec& L) {
        return launch_bn_eval_coeffs(params + L.gamma_off, params + L.beta_off, bufs + L.rm_off, bufs + L.rv_off, 1e-5f, coef(c, L, 0),
                                     coef(c, L, 1), coef(c, L, 2), coef(c, L, 3), L.Co, s);
      };
      auto fused = [&](const ConvSpec& L, const float* X, float* out, int flags) {
        gg_set_tile_counters(reinterpret_cast<unsigned*>(arena + P.ctr_off) + (&L - P.convs.data()) * 8);
        struct DropCounters { ~DropCounters() { gg_set_tile_counters(nullptr, 0); } } drop;
        return conv_forward_launch_affine(X, fwd_weights(c, L), out, coef(c, L, 2), coef(c, L, 3), F, L.Hi, L.Wi, L.Ci, L.Co, L.k,
                                          L.stride, L.pad, flags, dt, s);
      };
      if (fus) {
        const float* cur = cur_in;
        for (int j = 0; j < B.nconv - 1; ++j) {
          const ConvSpec& L = P.convs[B.conv[j]];
          TRY(coeffs(L));
          TRY(fused(L, cur, arena + L.Z_off, EPI_AFFINE | EPI_RELU));
          cur = arena + L.Z_off;
        }
        float* res = const_cast<float*>(cur_in);       // identity block: the sum replaces the block input
        if (B.ds >= 0) {
          const ConvSpec& Ld = P.convs[B.ds];
          TRY(coeffs(Ld));
          res = arena + Ld.Y_off;
          TRY(fused(Ld, cur_in, res, EPI_AFFINE));
        }
        const ConvSpec& LL = P.convs[nlast];
        TRY(coeffs(LL));
        TRY(fused(LL, cur, res, EPI_AFFINE | EPI_ACCUM | EPI_RELU));
        cur_in = res;
      } else {
        const float* cur = cur_in;
        for (int j = 0; j < B.nconv; ++j) {
          const ConvSpec& L = P.convs[B.conv[j]];
          TRY(conv_bn(c, L, cur));
          if (j < B.nconv - 1) {
            TRY(launch_bn_act_fwd(arena + L.Y_off, coef(c, L, 2), coef(c, L, 3), nullptr, nullptr, nullptr, arena + L.Z_off,
                                  (long long)F * L.Ho * L.Wo, L.Co, 1, nullptr, dt, s));
            cur = arena + L.Z_off;
          }
        }
        const ConvSpec& LL = P.convs[nlast];
        const long long rows = (long long)F * B.Ho * B.Wo;
        if (B.ds >= 0) {
          const ConvSpec& Ld = P.convs[B.ds];
          TRY(conv_bn(c, Ld, cur_in));
          TRY(launch_bn_act_fwd(arena + LL.Y_off, coef(c, LL, 2), coef(c, LL, 3), arena + Ld.Y_off, coef(c, Ld, 2), coef(c, Ld, 3),
                                arena + B.out_off, rows, B.Co, 1, nullptr, dt, s));
        } else {
          TRY(launch_bn_act_fwd(arena + LL.Y_off, coef(c, LL, 2), coef(c, LL, 3), cur_in, nullptr, nullptr, arena + B.out_off, rows,
                                B.Co, 1, nullptr, dt, s));
        }
        cur_in = arena + B.out_off;
      }
    }
    const BlockSpec& lastb = P.blocks.back();
    TRY(launch_avgpool_fwd(cur_in, h_out, F, lastb.Ho * lastb.Wo, lastb.Co, dt, s));
    return 0;
  }
  for (const BlockSpec& B : P.blocks) {
    const float* Xin = arena + B.in_off;
    const float* cur = Xin;
    for (int j = 0; j < B.nconv; ++j) {
      const ConvSpec& L = P.convs[B.conv[j]];
      TRY(conv_bn(c, L, cur));
      if (j < B.nconv - 1) {
        const long long rows = (long long)F * L.Ho * L.Wo;
        TRY(launch_bn_act_fwd(arena + L.Y_off, coef(c, L, 2), coef(c, L, 3), nullptr, nullptr, nullptr, arena + L.Z_off, rows,
                              L.Co, 1, nullptr, dt, s));
        cur = arena + L.Z_off;
      }
    }
    const ConvSpec& LL = P.convs[B.conv[B.nconv - 1]];
    const long long rows = (long long)F * B.Ho * B.Wo;
    unsigned* mask = reinterpret_cast<unsigned*>(arena + B.mask_off);   // [z > 0] bits of the block output, for backward
    if (B.ds >= 0) {
      const ConvSpec& Ld = P.convs[B.ds];
      TRY(conv_bn(c, Ld, Xin));
      TRY(launch_bn_act_fwd(arena + LL.Y_off, coef(c, LL, 2), coef(c, LL, 3), arena + Ld.Y_off, coef(c, Ld, 2), coef(c, Ld, 3),
                            arena + B.out_off, rows, B.Co, 1, mask, dt, s));
    } else {
      TRY(launch_bn_act_fwd(arena + LL.Y_off, coef(c, LL, 2), coef(c, LL, 3), Xin, nullptr, nullptr, arena + B.out_off, rows,
                            B.Co, 1, mask, dt, s));
    }
  }
  const BlockSpec& last = P.blocks.back();
  TRY(launch_avgpool_fwd(arena + last.out_off, h_out, F, last.Ho * last.Wo, last.Co, dt, s));
  return 0;
}

// BatchNorm(+ReLU / residual mask) backward of layer L: dZ -> dY, parameter gradients into the flat gradient buffer
// fused_rows > 0: the dgrad that produced dZ already wrote this BatchNorm's backward partials (EPI_BNRED, sum(g) and
// sum(g (y - mean)) per 64 rows) into the partial buffer: the stand-alone first pass is skipped
// first pass + combine: the parameter gradients and the two per-channel coefficients c1 / c2 the second pass needs
static int bn_backward_sums(Ctx& c, const ConvSpec& L, const float* dZ, const unsigned* Zbits, int fused_rows) {
  Plan& P = c.P;
  const long long rows = (long long)P.F * L.Ho * L.Wo;
  float* partial = c.arena + P.partial_off;
  double* acc = reinterpret_cast<double*>(c.arena + P.acc_off);
  const float* Y = c.arena + L.Y_off;
  int prow = fused_rows;
  if (!fused_rows) {
    TRY(launch_bn_bwd_reduce(dZ, nullptr, Zbits, Y, coef(c, L, 2), coef(c, L, 3), coef(c, L, 0), coef(c, L, 1), partial, rows, L.Co, c.dt, c.s));
    prow = bn_bwd_partial_rows(rows, L.Co, c.dt);
  }
  TRY(launch_bn_stats_reduce(partial, prow, L.Co, acc, c.s));
  return launch_bn_bwd_finalize_rows(acc, prow, rows, P.last_training, c.grads + L.gamma_off, c.grads + L.beta_off, coef(c, L, 4),
                                     coef(c, L, 5), c.accumulate, L.Co, c.s, fused_rows ? coef(c, L, 1) : nullptr);
}
static int bn_backward(Ctx& c, const ConvSpec& L, const float* dZ, const unsigned* Zbits, float* dY, int fused_rows = 0) {
  TRY(bn_backward_sums(c, L, dZ, Zbits, fused_rows));
  const long long rows = (long long)c.P.F * L.Ho * L.Wo;
  return launch_bn_bwd_apply(dZ, nullptr, Zbits, c.arena + L.Y_off, coef(c, L, 2), coef(c, L, 3), coef(c, L, 0), coef(c, L, 1), coef(c, L, 4),
                             coef(c, L, 5), dY, rows, L.Co, c.dt, c.s);
}
// The two BatchNorms that feed a downsample block's add + ReLU (its last convolution's and the downsample convolution's) see the SAME
// masked output gradient: their second passes run as ONE launch that reads dOut and the mask bits once (bn.hip, bn_bwd_apply2).
// The sums of the two are taken one after the other (they share the partial / accumulator scratch).
static int bn_backward_pair(Ctx& c, const ConvSpec& L, const ConvSpec& Ld, const float* dZ, const unsigned* Zbits, float* dY, float* dYd,
                            int fused_rows) {
  R3M_REQUIRE(L.Co == Ld.Co && L.Ho == Ld.Ho && L.Wo == Ld.Wo && Zbits, "bn_backward_pair: the two BatchNorms must have one shape and mask bits");
  const long long rows = (long long)c.P.F * L.Ho * L.Wo;
  if (!fused_rows && bn_bwd_reduce2_available(L.Co, c.dt)) {
    // both first passes are stand-alone (bf16 plans): one launch, two partial sets, then the two combines one after the other
    Plan& P = c.P;
    float* partial = c.arena + P.partial_off;
    double* acc = reinterpret_cast<double*>(c.arena + P.acc_off);
    const int prow = bn_bwd_partial_rows(rows, L.Co, c.dt);
    const long long set = (long long)prow * 2 * L.Co;
    TRY(launch_bn_bwd_reduce2(dZ, Zbits, c.arena + L.Y_off, c.arena + L.coef_off, c.arena + Ld.Y_off, c.arena + Ld.coef_off, partial, set,
                              rows, L.Co, c.dt, c.s));
    const ConvSpec* two[2] = {&L, &Ld};
    for (int k = 0; k < 2; ++k) {
      const ConvSpec& Q = *two[k];
      TRY(launch_bn_stats_reduce(partial + k * set, prow, Q.Co, acc, c.s));
      TRY(launch_bn_bwd_finalize_rows(acc, prow, rows, P.last_training, c.grads + Q.gamma_off, c.grads + Q.beta_off, coef(c, Q, 4),
                                      coef(c, Q, 5), c.accumulate, Q.Co, c.s, nullptr));
    }
  } else {
    TRY(bn_backward_sums(c, L, dZ, Zbits, fused_rows));
    TRY(bn_backward_sums(c, Ld, dZ, Zbits, 0));
  }
  return launch_bn_bwd_apply2(dZ, Zbits, c.arena + L.Y_off, c.arena + L.coef_off, dY, c.arena + Ld.Y_off, c.arena + Ld.coef_off, dYd, rows,
                              L.Co, c.dt, c.s);
}

static int wgrad(Ctx& c, const ConvSpec& L, const float* X, const float* dY) {
  return conv_wgrad_launch(X, dY, c.grads + L.w_off, c.arena + c.P.wgp_off, c.P.F, L.Hi, L.Wi, L.Ci, L.Co, L.k, L.stride, L.pad,
                           c.accumulate, c.dt, c.s);
}

// bn_of: the conv layer whose BatchNorm consumes dX as its dz (its Y has dX's shape), bn_bits: that BatchNorm's output mask bits
// (block outputs) or null (mask recomputed from Y). *fused_rows_out receives the partial-row count (0 = not fused).
static int dgrad(Ctx& c, const ConvSpec& L, const float* dY, float* dX, int flags, const float* add0, const unsigned* addbits,
                 const ConvSpec* bn_of = nullptr, const unsigned* bn_bits = nullptr, int* fused_rows_out = nullptr) {
  // the layer's [Ci][k*k][Co] weight image was built at the start of this backward (plan_backward, stage 0)
  float* Wt = c.dt == DT_BF16 ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(c.arena + c.P.wt_off) + L.wt_off)
                              : c.arena + c.P.wt_off + L.wt_off;
  if (fused_rows_out) *fused_rows_out = 0;
  gg_set_tile_counters(reinterpret_cast<unsigned*>(c.arena + c.P.ctr_off) + (c.P.convs.size() + 4 * (&L - c.P.convs.data())) * 8, 4);   // backward part
  // 1x1 stride-2 dgrads leave three of four parity classes without taps (plain zero / no-op launches): not fused
  const bool fuse = bn_of && fused_rows_out && c.P.fuse_bnred && !(L.stride == 2 && L.k == 1);
  struct DropCounters { ~DropCounters() { gg_set_tile_counters(nullptr, 0); } } drop;   // sets this layer did not use stay unused
  if (!fuse)
    return conv_dgrad_launch(dY, Wt, dX, add0, nullptr, addbits, c.P.F, L.Hi, L.Wi, L.Ci, L.Co, L.k, L.stride, L.pad, flags, c.dt, c.s);
  BnRedArgs br{c.arena + bn_of->Y_off, bn_bits, c.arena + bn_of->coef_off + 2LL * bn_of->Co, c.arena + bn_of->coef_off + 3LL * bn_of->Co,
               c.arena + bn_of->coef_off, c.arena + c.P.partial_off, 0};
  TRY(conv_dgrad_launch_br(dY, Wt, dX, add0, nullptr, addbits, c.P.F, L.Hi, L.Wi, L.Ci, L.Co, L.k, L.stride, L.pad, flags, c.dt, &br, c.s));
  *fused_rows_out = br.rows_out;
  return 0;
}

// Backward stages: 0 = avgpool + layer4, 1 = layer3, 2 = layer2, 3 = layer1 + stem. The gradient w.r.t. the current
// block output lives in one of five arena buffers (roles rotate: D = dOut, A0/A1 = dY alternating, B, C); the roles are
// carried across calls so stages can be issued one by one (the data-parallel wrapper launches the RCCL all-reduce of a
// finished stage's gradient slice in between).
//
// Streams: wgrad(L) only needs dY_L and the saved activations, and nothing on the critical path needs its result before
// the optimizer. It runs on a side stream, ordered by events so that it overlaps ONLY with the HBM-bound BatchNorm-backward
// passes of the next layer down (reduce / apply: ~6 TB/s, no MFMA) and never with the MFMA-bound dgrad:
//     main:  bn_bwd(L) -> [wait wgrad(L+1)] -> dgrad(L) -> bn_bwd(L-1) -> [wait wgrad(L)] -> dgrad(L-1) -> ...
//     side:                                    wgrad(L)  (starts when dgrad(L) has finished)
// A bandwidth-bound and a matrix-bound kernel share the CUs without stealing each other's bottleneck resource, and the
// per-launch timings of the dominant kernel class (gather-GEMM on the main stream) stay unperturbed. Each stage ends with
// a join. Off by default (R3M_SIDE_STREAM=1 enables it): measured neutral, see side_init().
static int side_init(Plan& P) {
  if (P.use_side < 0) {
    // Opt-in. Measured on ResNet-50 F=1280 (profiles/r01 notes in DESIGN.md): co-running wgrad with the BatchNorm-backward
    // passes lengthens the wgrad launches by about the BatchNorm time (the two do not overlap usefully on gfx950 even
    // though one is HBM-bound and the other MFMA-bound) -> step time unchanged (364.9 vs 364.5 ms). Kept for experiments.
    // 2 (round 5 experiment): wgrad(L) starts TOGETHER with dgrad(L) (both wait for dY_L only) and nothing on the main stream waits
    // for it before its dY buffer is rewritten: the two GEMMs fill each other's tile-quantisation tails (every launch of the
    // 1280-frame step has 490 k tiles for 512 slots).
    P.use_side = R3M_ENV_INT("R3M_SIDE_STREAM", 0);
  }
  if (!P.bnred_init) {
    P.bnred_init = true;
    // fp32 plans only: there the dgrad is MFMA-bound and the extra epilogue loads ride under other blocks' matrix work (A/B on one
    // box, probe build: 343.0 / 341.7 ms -> 338.5 / 339.0 ms per ResNet-50 step). bf16 plans are HBM/epilogue-bound already and
    // measured slightly SLOWER with it (ResNet-50 95.6 -> 96.2 ms, ResNet-34 97.0 -> 97.9 ms), so they keep the stand-alone reduce.
    // R3M_BNRED (probe builds): 0 = off everywhere, 2 = on for bf16 too.
    {
      const int v = R3M_ENV_INT("R3M_BNRED", 1);
      if (v != 1) P.fuse_bnred = v == 2;       // probe builds: 0 = off everywhere, 2 = on for bf16 too; 1 = the plan's own setting
    }
  }
  if (!P.use_side || P.side) return 0;
  if (hipStreamCreateWithFlags(&P.side, hipStreamNonBlocking) != hipSuccess) { set_last_error("side stream: create failed"); return 1; }
  hipEvent_t* evs[4] = {&P.ev_dy, &P.ev_wg[0], &P.ev_wg[1], &P.ev_join};
  for (auto ev : evs)
    if (hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) { set_last_error("side stream: event create failed"); return 1; }
  return 0;
}

int plan_backward(Plan& P, const float* dh, const float* params, float* grads, float* arena, int stage_begin, int stage_end,
                  int accumulate, int* gd_io, hipStream_t s) {
  Ctx c{P, params, grads, nullptr, arena, s, P.last_training, accumulate, P.dtype};
  const int dt = P.dtype;
  R3M_REQUIRE(stage_begin >= 0 && stage_end <= 4 && stage_begin < stage_end, "resnet_backward: stages [%d, %d) outside [0, 4)", stage_begin, stage_end);
  R3M_REQUIRE(P.next_stage != -3, "resnet_backward: the last forward on this plan ran in inference mode (training = 2): nothing was kept for a backward");
  R3M_REQUIRE(P.next_stage != -1, "resnet_backward: no forward has run on this plan");
  // stage 0 may always (re)start a backward over the saved activations (retain_graph); any other stage must continue the
  // sequence the previous call left off at — its inputs (running output gradient, pending EPI_BNRED partials) live in the plan
  R3M_REQUIRE(stage_begin == 0 || stage_begin == P.next_stage,
              "resnet_backward: stage %d requested but the plan expects stage %d (stages run 0..3 in order after each forward; "
              "stage 0 restarts)", stage_begin, P.next_stage);
  P.next_stage = -2;           // poisoned while in flight: after a failed call only stage 0 (a restart) is accepted
  TRY(side_init(P));
  if (stage_begin == 0) {       // the weights are final since the last optimizer step: all dgrad weight images in one launch
    if (hipMemsetAsync(arena + P.ctr_off + P.convs.size() * 8, 0, 4 * P.convs.size() * 8 * sizeof(unsigned), s) != hipSuccess) {
      set_last_error("resnet_backward: cannot reset the tile queues");
      return 1;
    }
    if (!P.d_wt_tab) {
      // failure-atomic: the plan's pointers are set only after both allocations and both uploads succeeded (a half-initialised
      // pair would make the next backward skip this block and launch transpose_w_all on a null / uninitialised table)
      const size_t tb = P.wt_tab.size() * sizeof(WtEntry), ib = P.wt_tile0.size() * sizeof(int);
      WtEntry* d_tab = nullptr;
      int* d_tile0 = nullptr;
      const bool ok = hipMalloc(reinterpret_cast<void**>(&d_tab), tb) == hipSuccess &&
                      hipMalloc(reinterpret_cast<void**>(&d_tile0), ib) == hipSuccess &&
                      hipMemcpy(d_tab, P.wt_tab.data(), tb, hipMemcpyHostToDevice) == hipSuccess &&
                      hipMemcpy(d_tile0, P.wt_tile0.data(), ib, hipMemcpyHostToDevice) == hipSuccess;
      if (!ok) {
        (void)hipGetLastError();
        if (d_tab) (void)hipFree(d_tab);
        if (d_tile0) (void)hipFree(d_tile0);
        set_last_error("resnet_backward: cannot allocate / upload the weight-image table (%zu + %zu bytes)", tb, ib);
        return 1;
      }
      P.d_wt_tab = d_tab;
      P.d_wt_tile0 = d_tile0;
    }
    TRY(launch_transpose_w_all(params, arena + P.wt_off, P.d_wt_tab, P.d_wt_tile0, (int)P.wt_tab.size(), P.wt_tile0.back(), P.dtype, s));
  }
  const bool side_on = P.use_side && P.side;
  Ctx cs = c;                       // context whose launches go to the side stream
  cs.s = side_on ? P.side : s;
  const int F = P.F;
  // buffer roles: role[0]=D, role[1]=A0, role[2]=A1, role[3]=B, role[4]=C  (indices into G_off), rotated per block
  int* role = P.roles;
  auto Gp = [&](int r) { return arena + P.G_off[role[r]]; };

  const bool side_co = side_on && P.use_side == 2;   // wgrad(L) runs beside dgrad(L)
  // mode 2: call right BEFORE dgrad(L) is enqueued — the side stream may start wgrad(L) as soon as dY_L is complete
  auto mark_dy = [&]() -> int {
    if (!side_co) return 0;
    if (hipEventRecord(P.ev_dy, s) != hipSuccess || hipStreamWaitEvent(P.side, P.ev_dy, 0) != hipSuccess) {
      set_last_error("side stream: event ordering failed");
      return 1;
    }
    return 0;
  };
  // call right AFTER dgrad(L) was enqueued on the main stream: wgrad(L) starts on the side stream once that dgrad is done (mode 1)
  auto wgrad_async = [&](const ConvSpec& L, const float* X, const float* dY, int ai) -> int {
    if (!side_on) return wgrad(c, L, X, dY);
    if (!side_co && (hipEventRecord(P.ev_dy, s) != hipSuccess || hipStreamWaitEvent(P.side, P.ev_dy, 0) != hipSuccess)) {
      set_last_error("side stream: event ordering failed");
      return 1;
    }
    TRY(wgrad(cs, L, X, dY));
    if (hipEventRecord(P.ev_wg[ai], P.side) != hipSuccess) { set_last_error("side stream: record failed"); return 1; }
    P.wg_pending[ai] = true;
    return 0;
  };
  // before the main stream overwrites A[ai], the wgrad that last read it must be done
  auto acquire_A = [&](int ai) -> int {
    if (side_on && P.wg_pending[ai]) {
      if (hipStreamWaitEvent(s, P.ev_wg[ai], 0) != hipSuccess) { set_last_error("side stream: wait failed"); return 1; }
      P.wg_pending[ai] = false;
    }
    return 0;
  };
  // before an MFMA-bound kernel goes to the main stream: every wgrad in flight must have finished
  auto wait_wgrads = [&]() -> int {
    if (side_co) return mark_dy();      // mode 2: no wait; the dY just written is what the side stream waits for
    TRY(acquire_A(0));
    return acquire_A(1);
  };
  auto join_side = [&]() -> int {
    if (!side_on) return 0;
    if (hipEventRecord(P.ev_join, P.side) != hipSuccess || hipStreamWaitEvent(s, P.ev_join, 0) != hipSuccess) {
      set_last_error("side stream: join failed");
      return 1;
    }
    P.wg_pending[0] = P.wg_pending[1] = false;
    return 0;
  };
  auto next_A = [&](int* ai) -> float* {
    *ai = P.a_next;
    P.a_next ^= 1;
    return Gp(1 + *ai);
  };

  for (int st = stage_begin; st < stage_end; ++st) {
    const int layer = 3 - st;
    if (st == 0) {
      const BlockSpec& last = P.blocks.back();
      for (int r = 0; r < 5; ++r) role[r] = r;
      P.a_next = 0;
      P.wg_pending[0] = P.wg_pending[1] = false;
      P.dout_fused_rows = 0;     // the last block's output gradient comes from the pool: its BatchNorm runs the stand-alone reduce
      TRY(launch_avgpool_bwd(dh, Gp(0), F, last.Ho * last.Wo, last.Co, dt, s));
    }
    for (int bi = (int)P.blocks.size() - 1; bi >= 0; --bi) {
      const BlockSpec& B = P.blocks[bi];
      if (B.stage != layer) continue;
      const float* dOut = Gp(0);
      const unsigned* Out = reinterpret_cast<const unsigned*>(arena + B.mask_off);   // [out > 0] bits
      const float* Xin = arena + B.in_off;
      float* Gb = Gp(3);
      float* Gc = Gp(4);
      // last conv of the block: its BatchNorm output joined the residual add, mask comes from the block output
      const float* dz = dOut;
      const unsigned* zmask = Out;
      // partials of the BatchNorm that consumes dz, written by the dgrad that produced dz (EPI_BNRED) — 0: none
      int dz_fused = P.dout_fused_rows;
      P.dout_fused_rows = 0;
      // the block whose output gradient this block's last dgrad completes, and the BatchNorm (its last conv's) that will read it
      const BlockSpec* Bprev = bi > 0 ? &P.blocks[bi - 1] : nullptr;
      const ConvSpec* Lprev_last = Bprev ? &P.convs[Bprev->conv[Bprev->nconv - 1]] : nullptr;
      const unsigned* prev_bits = Bprev ? reinterpret_cast<const unsigned*>(arena + Bprev->mask_off) : nullptr;
      int ai;
      // downsample block: both tail BatchNorms in one second pass, the downsample one's dY parked in E until the end of the block
      // (R3M_BN_PAIR=0 in probe builds: two separate passes, for A/B)
      const bool pair = B.ds >= 0 && P.E_off >= 0 && B.nconv >= 2 && P.bn_pair && R3M_ENV_INT("R3M_BN_PAIR", 1) != 0 && P.convs[B.ds].Co >= 8;
      float* const dYd_pair = pair ? arena + P.E_off : nullptr;
      for (int j = B.nconv - 1; j >= 1; --j) {
        const ConvSpec& L = P.convs[B.conv[j]];
        const ConvSpec& Lprev = P.convs[B.conv[j - 1]];
        float* dY = next_A(&ai);
        TRY(acquire_A(ai));
        if (pair && j == B.nconv - 1) TRY(bn_backward_pair(c, L, P.convs[B.ds], dz, zmask, dY, dYd_pair, dz_fused));
        else
        TRY(bn_backward(c, L, dz, zmask, dY, dz_fused));     // HBM-bound: overlaps the previous layer's wgrad
        TRY(wait_wgrads());
        TRY(dgrad(c, L, dY, Gb, 0, nullptr, nullptr, &Lprev, nullptr, &dz_fused));   // Gb = dz of Lprev's BatchNorm + its partials
        TRY(wgrad_async(L, arena + Lprev.Z_off, dY, ai));
        dz = Gb; zmask = nullptr;   // Gb is consumed by the next bn_backward before a later dgrad rewrites it
      }
      const ConvSpec& L1 = P.convs[B.conv[0]];
      float* dY1 = next_A(&ai);
      TRY(acquire_A(ai));
      TRY(bn_backward(c, L1, dz, zmask, dY1, dz_fused));
      TRY(wait_wgrads());
      if (B.ds >= 0) {
        const ConvSpec& Ld = P.convs[B.ds];
        TRY(dgrad(c, L1, dY1, Gc, 0, nullptr, nullptr));
        TRY(wgrad_async(L1, Xin, dY1, ai));
        if (pair) {                                     // dY of the downsample BatchNorm has been waiting in E since the block's first pass
          TRY(dgrad(c, Ld, dYd_pair, Gc, EPI_ACCUM, nullptr, nullptr));
          TRY(wgrad_async(Ld, Xin, dYd_pair, ai));      // (side stream: ordered behind conv1's wgrad, same event slot)
        } else {
          int ad;
          float* dYd = next_A(&ad);
          TRY(acquire_A(ad));
          TRY(bn_backward(c, Ld, dOut, Out, dYd));        // overlaps wgrad(conv1); always the stand-alone reduce (second consumer of dOut)
          TRY(wait_wgrads());
          TRY(dgrad(c, Ld, dYd, Gc, EPI_ACCUM, nullptr, nullptr));
          TRY(wgrad_async(Ld, Xin, dYd, ad));
        }
      } else {
        // Gc = dgrad + masked residual gradient = the previous block's COMPLETE output gradient: also emit the partials of the
        // BatchNorm that will consume it (the previous block's last one, masked by that block's output bits)
        TRY(dgrad(c, L1, dY1, Gc, EPI_MASKED_ADD, dOut, Out, Lprev_last, prev_bits, &P.dout_fused_rows));
        TRY(wgrad_async(L1, Xin, dY1, ai));
      }
      // C becomes the gradient of the previous block's output; the old D is free (only the main stream ever read it)
      const int t = role[0]; role[0] = role[4]; role[4] = t;
    }
    if (st == 3) {
      // stem: maxpool + BN/ReLU backward (fused) -> conv1 weight gradient (no input gradient)
      const ConvSpec& L0 = P.convs[0];
      float* Gc = Gp(4);
      {   // MaxPool backward gathered inside both BatchNorm-backward passes (no dZ0 tensor)
        const unsigned char* am = reinterpret_cast<const unsigned char*>(arena + P.amax_off);
        const long long rows = (long long)F * 112 * 112;
        float* partial = arena + P.partial_off;
        double* acc = reinterpret_cast<double*>(arena + P.acc_off);
        TRY(launch_bn_bwd_reduce_pool(Gp(0), am, arena + L0.Y_off, coef(c, L0, 2), coef(c, L0, 3), coef(c, L0, 0), coef(c, L0, 1), partial,
                                      F, 112, 112, 64, dt, s));
        const int prow = bn_bwd_pool_partial_rows(F, 112, 112, 64);
        TRY(launch_bn_stats_reduce(partial, prow, 64, acc, s));
        TRY(launch_bn_bwd_finalize_rows(acc, prow, rows, P.last_training, grads + L0.gamma_off, grads + L0.beta_off, coef(c, L0, 4),
                                        coef(c, L0, 5), accumulate, 64, s));
        TRY(launch_bn_bwd_apply_pool(Gp(0), am, arena + L0.Y_off, coef(c, L0, 2), coef(c, L0, 3), coef(c, L0, 0), coef(c, L0, 1),
                                     coef(c, L0, 4), coef(c, L0, 5), Gc, F, 112, 112, 64, dt, s));
      }
      TRY(join_side());   // the stem wgrad shares the split-K scratch with the side stream's wgrads
      if (dt == DT_BF16) TRY(launch_stem_wgrad16(arena + P.col_off, Gc, grads + L0.w_off, arena + P.wgp_off, F, accumulate, s));
      else TRY(launch_stem_wgrad(arena + P.col_off, Gc, grads + L0.w_off, arena + P.wgp_off, F, accumulate, dt, s));
    }
    TRY(join_side());     // a finished stage's gradients are complete on the main stream (all-reduce hook, Adam)
  }
  *gd_io = role[0];
  P.next_stage = stage_end == 4 ? 0 : stage_end;
  return 0;
}

// ---- accessors for the C ABI ----
int plan_out_dim(Plan* P) { return P->D; }
int plan_dtype(Plan* P) { return P->dtype; }
long long plan_num_params(Plan* P) { return P->n_params; }
long long plan_num_buffers(Plan* P) { return P->n_buffers; }
long long plan_arena_floats(Plan* P) { return P->arena_floats; }
int plan_num_tensors(Plan* P) { return (int)P->tensors.size(); }
int plan_tensor_info(Plan* P, int i, char* name, int cap, int* kind, long long* offset, int* ndim, int* shape4) {
  R3M_REQUIRE(i >= 0 && i < (int)P->tensors.size(), "tensor_info: index %d out of range", i);
  const TensorInfo& t = P->tensors[i];
  if (name && cap > 0) { strncpy(name, t.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (kind) *kind = t.kind;
  if (offset) *offset = t.offset;
  if (ndim) *ndim = t.ndim;
  if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = t.shape[k];
  return 0;
}
// backward stage s covers layer (4 - s); stage 3 also covers the stem (its params sit before layer1's)
int plan_stage_range(Plan* P, int stage, long long* off, long long* count) {
  R3M_REQUIRE(stage >= 0 && stage < 4, "stage_range: stage %d", stage);
  const int L = 3 - stage;
  const long long b = P->stage_param_begin[L], e = P->stage_param_begin[L + 1];
  if (off) *off = b;
  if (count) *count = e - b;
  return 0;
}
void plan_destroy(Plan* P) {
  if (P->side) {
    (void)hipStreamDestroy(P->side);
    for (hipEvent_t ev : {P->ev_dy, P->ev_wg[0], P->ev_wg[1], P->ev_join})
      if (ev) (void)hipEventDestroy(ev);
  }
  if (P->d_wt_tab) (void)hipFree(P->d_wt_tab);
  if (P->d_wt_tile0) (void)hipFree(P->d_wt_tile0);
  delete P;
}
int* plan_gd(Plan* P) { return &P->gd; }
// per-plan option: 1 = BatchNorm-backward partials from the dgrad epilogues (EPI_BNRED), 0 = stand-alone reduce passes
int plan_set_bn_pair(Plan* P, int on) { const int old = P->bn_pair; P->bn_pair = on ? 1 : 0; return old; }
int plan_set_fuse_bnred(Plan* P, int on) { const int old = P->fuse_bnred; P->fuse_bnred = on ? 1 : 0; return old; }

}  // namespace r3m
