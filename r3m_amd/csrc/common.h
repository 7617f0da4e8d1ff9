// r3m_amd — internal declarations shared by the HIP translation units.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, fp32-input MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// ---- experiment switches -------------------------------------------------------------------------------------
// The SHIPPED library (r3m_amd/csrc/build.sh) has one code path per shape: it reads no environment variables and contains
// none of the timing probes (some of which deliberately produce wrong results). Builds with -DR3M_PROBES
// (tools/build_ab.sh none probes) bring both back so the A/B measurements quoted in DESIGN.md can be repeated.
#ifdef R3M_PROBES
#include <cstdlib>
#define R3M_ENV_INT(name, dflt) ([]() -> int { static const int v_ = []() { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }(); return v_; }())
#define R3M_PROBE(p) ((p).debug)
#else
#define R3M_ENV_INT(name, dflt) (dflt)
#define R3M_PROBE(p) 0
#endif

namespace r3m {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (thread-local message, C-ABI returns non-zero on failure) ----
void set_last_error(const char* fmt, ...);
int  check_launch(const char* what);  // hipGetLastError() -> 0 / code (+ message)

#define R3M_REQUIRE(cond, ...)                  \
  do {                                          \
    if (!(cond)) {                              \
      r3m::set_last_error(__VA_ARGS__);         \
      return 1;                                 \
    }                                           \
  } while (0)

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Opt-in for > 64 KiB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize), cached per call site AND per device:
// function attributes belong to a device, and launches arrive from several host threads (forward: caller, backward: autograd
// engine). The cache is a plain int per device — a racing pair of threads sets the same attribute twice, which is idempotent.
struct DynLdsOptIn { int bytes[32] = {0}; };
int ensure_dyn_lds(DynLdsOptIn& cache, const void* fn, int bytes, const char* what);

// ---- epilogue flags of the gather-GEMM (conv fwd / dgrad / linear) ----
enum : int {
  EPI_STATS      = 1,   // emit per-(row-block, column) sum / sum-of-squares partials (BatchNorm statistics)
  EPI_ACCUM      = 2,   // out = acc + out_old
  EPI_MASKED_ADD = 4,   // out = acc + add0 * (add1 > 0)   (residual gradient joining a dgrad)
  EPI_BIAS       = 8,   // out = acc + bias[col]
  EPI_RELU       = 16,  // out = max(out, 0)
  EPI_MASK_OUT   = 32,  // out = (add1 > 0) ? acc : 0        (ReLU backward fused into a Linear dgrad)
  EPI_BNRED      = 64,  // dgrad only: the result IS the gradient dz entering a BatchNorm(+ReLU) whose pre-normalisation output is
                        // bn_y — also emit that BatchNorm's backward partials per 64 result rows: sum(g), sum(g * (y - mean)),
                        // g = dz * [relu mask] (mask: bn_bits, or recomputed as fma(y, bn_scale, bn_shift) > 0). Replaces the
                        // stand-alone first pass of BatchNorm backward (one read of dz and of y saved per layer).
  EPI_AFFINE     = 128, // forward only (round 6, the inference path): out = acc * bn_scale[col] + bn_shift[col] — eval-mode BatchNorm
                        // (running statistics) applied where the convolution result is stored; then EPI_ACCUM (+ the residual already
                        // in `out`) and EPI_RELU in that order: relu(bn(conv(x)) + identity) of a block tail in ONE store.
};

// rows of the EPI_BNRED partial buffer one launch writes ([rows][2][Nc] floats): one per 64 GEMM rows, tile-independent
static inline int bnred_partial_rows(long long M) { return (int)((M + 63) / 64); }

constexpr int MAX_TAPS = 49;

// activation storage type of a launch / an encoder plan
enum : int { DT_F32 = 0, DT_BF16 = 1 };

// One launch of the gather-GEMM:  out[row(m), :] = sum_t  in[pix(m) + (dy_t, dx_t), :] * B[:, wt_t, :]^T
struct GatherGemmParams {
  const float* A;      // input activations, NHWC [N, Hi, Wi, Ci]
  const float* B;      // weights [Nc][T][Ci]  (row = output column, K contiguous)
  float* out;          // output activations, NHWC [N, Ho, Wo, Nc]
  const float* add0;   // EPI_MASKED_ADD: gradient tensor, same shape as out
  const float* add1;   // EPI_MASKED_ADD / EPI_MASK_OUT: mask source (post-ReLU activation), same shape as out
  const unsigned* addbits;  // EPI_MASKED_ADD: optional 1-bit/element ReLU mask of that activation (used instead of add1)
  const float* bias;   // EPI_BIAS: [Nc]
  float* stats;        // EPI_STATS: [gridM][2][Nc];  EPI_BNRED: [bnred_partial_rows(M)][2][Nc]
  const float* bn_y;        // EPI_BNRED: the consumer BatchNorm's input (conv output), same shape / storage type as out
  const unsigned* bn_bits;  // EPI_BNRED: 1-bit ReLU mask of the BatchNorm(+residual) output, or null -> recompute from bn_y
  const float* bn_scale;    // EPI_BNRED: [Nc] forward coefficients (mask recomputation)
  const float* bn_shift;
  const float* bn_mean;     // EPI_BNRED: [Nc] batch (or running) mean
  int N, Hi, Wi, Ci;
  int Hg, Wg;          // GEMM row grid per image: m = (n*Hg + gy)*Wg + gx
  int Ho, Wo, Nc;      // output tensor dims (Nc = GEMM N)
  int is;              // input pixel = (gy*is + dy_t, gx*is + dx_t)
  int os, ooy, oox;    // output pixel = (gy*os + ooy, gx*os + oox)
  int M;               // N*Hg*Wg
  int T;               // taps in the weight tensor (B row stride = T*Ci)
  int ntaps;           // taps visited by this launch
  int flags;
  int simple_rows;     // 1: 1x1 / stride 1 / no padding -> input row offset = m*Ci (no pixel decode)
  int debug;           // timing probes (R3M_GG_DEBUG), 0 in production
  int dtype;           // DT_F32: A/B/out/add0/add1 are fp32; DT_BF16: they address bf16 tensors (stats/bias stay fp32)
  unsigned* tile_ctr;  // persistent kernels (conv_pw.hip): 8 zeroed counters = per-XCD tile queues, or null -> static tile assignment
  signed char dy[MAX_TAPS];
  signed char dx[MAX_TAPS];
  unsigned char wt[MAX_TAPS];
  int tap[MAX_TAPS];   // packed by the launcher: (dy & 255) | (dx & 255) << 8 | wt << 16  (dword table -> scalar loads)
};

struct WgradParams {
  const float* dY;   // [M][Co]  (NHWC rows of the conv output gradient)
  const float* X;    // conv input, NHWC [N, Hi, Wi, Ci]
  float* out;        // [splitK][Co][T][Ci] partials (or the gradient itself when splitK == 1)
  int N, Hi, Wi, Ci;
  int Ho, Wo, Co;
  int KH, KW, stride, pad;
  int M;               // N*Ho*Wo
  int rows_per_split;  // multiple of 32
  int simple_rows;     // 1x1 stride-1: X row offset = m*Ci
  int tilesN;          // ceil(Ci / BN)
  int interleave;      // 1: spread the next K step's DMA pieces between this step's MFMAs (set by the launcher)
  int dtype;           // DT_F32 / DT_BF16 storage of dY and X (out is always fp32)
  int gx;              // (co tile, ci tile, tap) blocks per split; the launch is 1-D: gx * splitK blocks
  int xcd;             // 1: XCD-aware block order — all blocks of one split (same dY / X rows) run on ONE XCD and share its L2
  int debug;           // timing probes (probe builds, R3M_WG_DEBUG: 1 no DMA at all, 2 no X pieces, 4 no dY pieces); 0 in production
};

// EPI_BNRED request of a dgrad (engine.hip conv_dgrad_launch_br): the result is the dz of a BatchNorm whose input is Y (same
// shape as dX); partial rows -> `partial`, count -> rows_out
struct BnRedArgs {
  const float* Y;
  const unsigned* bits;      // 1-bit ReLU mask of the BatchNorm(+residual) output, or null: recompute from Y, scale, shift
  const float* scale;
  const float* shift;
  const float* mean;
  float* partial;
  int rows_out;              // partial rows written (all launches of the dgrad)
};
int engine_set_fused_inference(int on);
int conv_dgrad_launch_br(const float* dY, const float* Wt, float* dX, const float* add0, const float* add1, const unsigned* addbits,
                         int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int flags, int dt, BnRedArgs* br,
                         hipStream_t s);

// ---- launchers (conv.hip) ----
int launch_gather_gemm(const GatherGemmParams& p, hipStream_t s);
double gather_gemm_alg_bytes(const GatherGemmParams& p, int elem_bytes);
bool gather_gemm_fuses_affine(const GatherGemmParams& p);   // inference forward: the launch's kernel has the EPI_AFFINE epilogues
int gather_gemm_grid_m(int M, int Nc);   // number of row blocks the launcher will use (stats partial rows)
// The engine hands the NEXT launch_gather_gemm of this thread 8 zeroed device counters (dynamic tile queues of the persistent
// kernel: a block that finds its CU shared with another stream's kernel — RCCL during an overlapped all-reduce — simply takes
// fewer tiles). Consumed (and cleared) by that launch whether or not it uses them; launches without it assign tiles statically.
void gg_set_tile_counters(unsigned* ctr8, int sets = 1);
void gg_route_record_begin(int* out, int cap);   // dry run on this thread: launch_gather_gemm records its kernel family and launches nothing
int gg_route_record_end();                       // -> launches seen since begin
bool wgrad_rowwin_eligible(const WgradParams& p);                       // wgrad_win.hip: 3x3 / stride 1 / pad 1, fp32, whole 64- or 128-wide tiles
int launch_wgrad_rowwin(WgradParams& p, int splitK, hipStream_t s);
int gg_set_dynamic_tiles(int on);        // diagnostic switch (r3m_debug_set_dynamic_tiles): 0 = ignore the counters, assign statically
bool pw_gemm_eligible(const GatherGemmParams& p);
int pw_gemm_form(const GatherGemmParams& p);            // 0 none, 1 pointwise, 2 gather, 3 gather with strided output rows          // conv_pw.hip: persistent kernel for 1x1 / stride-1 launches (fp32)
int launch_pw_gemm(const GatherGemmParams& p, hipStream_t s);
int launch_wgrad(const WgradParams& p, int splitK, hipStream_t s);
int wgrad_pick_split(int M, int Co, int Ci, int T);
int launch_wgrad_reduce(const float* partial, float* dW, long long n, int splitK, int accumulate, hipStream_t s);
struct WtEntry { long long w_off, wt_off; int Co, T, Ci, pad_; };
int launch_transpose_w_all(const float* params, void* wt, const WtEntry* tab, const int* tile0, int n, int tiles, int dt, hipStream_t s);
int launch_transpose_w(const float* W, float* Wt, int Co, int T, int Ci, hipStream_t s);
int launch_stem_prep(const float* x_nchw, float* xn, int F, hipStream_t s);
struct FrameSource;   // augment_dev.h: raw clips + crop boxes
int launch_stem_prep_crop(const FrameSource& src, float* xn, int F, hipStream_t s);
int launch_stem_prep16_crop(const FrameSource& src, void* xn16, int F, hipStream_t s);
int launch_stem_fwd(const float* xn, const float* w147, void* y, float* stats, int F, int dt, hipStream_t s);
size_t stem_wgrad_ws_floats();
int launch_stem_wgrad(const float* xn, const void* dY, float* dw147, float* ws, int F, int accumulate, int dt, hipStream_t s);

// ---- launchers (conv_bf16.hip) ----
int launch_gather_gemm_bf16(const GatherGemmParams& p, hipStream_t s);
int launch_wgrad_bf16(const WgradParams& p, int splitK, hipStream_t s);
int wgrad_bf16_pick_split(int M, int Co, int Ci, int T);
int launch_convert_bf16(const float* src, void* dst, long long n, hipStream_t s);
// conv_pw16.hip: persistent warp-specialised kernel of the bf16 plans (dense / parity-strided output rows)
#ifdef R3M_PROBES
int pw16_form(const GatherGemmParams& p);               // 0 none, 1 pointwise, 2 gather, 3 gather with strided output rows
int launch_pw16(const GatherGemmParams& p, hipStream_t s);
int pw16_set_mode(int mode);                            // diagnostic (r3m_debug_set_pw16): 0 = per-tile kernels everywhere; returns the old value
#else                                                   // shipped library: the experiment is not compiled in
static inline int pw16_form(const GatherGemmParams&) { return 0; }
static inline int launch_pw16(const GatherGemmParams&, hipStream_t) { return 1; }
static inline int pw16_set_mode(int) { return -1; }
#endif
int gg16_route(const GatherGemmParams& p);              // conv_bf16.hip: kernel family of a bf16 launch (r3m_debug_conv_route): 30 gather, 31 halo, 32 kernel-row, 33 probe-build persistent kernel
// conv_row16.hip (round 6): persistent kernel-row kernel of the 128-multiple-wide bf16 3x3 / stride-1 launches
bool row16_eligible(const GatherGemmParams& p);
int launch_conv3x3_row_bf16(const GatherGemmParams& p, hipStream_t s);
int row16_set_mode(int mode);                           // diagnostic (r3m_debug_set_conv3x3_bf16): 0 = per-tile halo kernels; returns the old value
int launch_transpose_w_bf16(const float* W, void* Wt, int Co, int T, int Ci, hipStream_t s);

// ---- launchers (stem_bf16.hip): the stem on the bf16 MFMA, from a padded bf16 image of the normalised frames ----
size_t stem_xn16_bytes(int F);
int launch_stem_prep16(const float* x_nchw, void* xn16, int F, hipStream_t s);
int launch_stem_fwd16(const void* xn16, const float* w147, void* y, float* stats, int F, hipStream_t s);
size_t stem_wgrad16_ws_floats();
int launch_stem_wgrad16(const void* xn16, const void* dY, float* dw147, float* ws, int F, int accumulate, hipStream_t s);

// ---- launchers (bn.hip) ----
size_t bn_acc_bytes(int C);   // fp64 slice accumulator: [slices <= max(64, min(256, 131072 / C))][2][C]
int launch_bn_stats_reduce(const float* partials, int rows, int C, double* acc /* bn_acc_bytes(C) */, hipStream_t s);
int launch_bn_finalize_rows(const double* acc, int stat_rows, long long count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* mean,
                            float* invstd, float* scale, float* shift, int C, hipStream_t s);
int launch_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          float eps, float* mean, float* invstd, float* scale, float* shift, int C, hipStream_t s);
// activation tensors are void*: fp32 (dt = DT_F32) or bf16 (DT_BF16); coefficients / partials / statistics are fp32
int launch_bn_act_fwd(const void* Y, const float* scale, const float* shift, const void* R, const float* scale2,
                      const float* shift2, void* Z, long long rows, int C, int relu, unsigned* maskbits, int dt, hipStream_t s);
int bn_bwd_partial_rows(long long rows, int C, int dt);
int launch_bn_bwd_reduce(const void* dZ, const void* Zmask, const unsigned* Zbits, const void* Y, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float* partials, long long rows, int C,
                         int dt, hipStream_t s);
int launch_bn_bwd_finalize_rows(const double* acc, int stat_rows, long long count, int use_batch_stats, float* dgamma,
                                float* dbeta, float* c1, float* c2, int accumulate, int C, hipStream_t s,
                                const float* second_sum_scale = nullptr /* EPI_BNRED partials: sum(g (y - mean)) * invstd[c] */);
int launch_bn_bwd_apply(const void* dZ, const void* Zmask, const unsigned* Zbits, const void* Y, const float* scale,
                        const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, void* dY,
                        long long rows, int C, int dt, hipStream_t s);
// stem tail fused (Z0 / dZ0 never materialised)
int launch_bn_relu_maxpool_fwd(const void* Y, const float* scale, const float* shift, void* P, unsigned char* amax, int N, int Hi,
                               int Wi, int C, int dt, hipStream_t s);
int bn_bwd_pool_partial_rows(int N, int Hi, int Wi, int C);
// the two BatchNorms of a downsample block's tail (same masked gradient) in one pass; coefA / coefB = [6][C] {mean, invstd, scale, shift, c1, c2}
int launch_bn_bwd_apply2(const void* dZ, const unsigned* Zbits, const void* YA, const float* coefA, void* dYA, const void* YB,
                         const float* coefB, void* dYB, long long rows, int C, int dt, hipStream_t s);
bool bn_bwd_reduce2_available(int C, int dt);
int launch_bn_bwd_reduce2(const void* dZ, const unsigned* Zbits, const void* YA, const float* coefA, const void* YB, const float* coefB,
                          float* partials, long long set_stride /* floats between the two partial sets */, long long rows, int C, int dt,
                          hipStream_t s);
int launch_bn_bwd_reduce_pool(const void* dP, const unsigned char* amax, const void* Y, const float* scale, const float* shift,
                              const float* mean, const float* invstd, float* partials, int N, int Hi, int Wi, int C, int dt,
                              hipStream_t s);
int launch_bn_bwd_apply_pool(const void* dP, const unsigned char* amax, const void* Y, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* c1, const float* c2, void* dY, int N, int Hi, int Wi,
                             int C, int dt, hipStream_t s);
int launch_maxpool_fwd(const void* Z, void* P, unsigned char* amax, int N, int Hi, int Wi, int C, int dt, hipStream_t s);
int launch_maxpool_bwd(const void* dP, const unsigned char* amax, void* dZ, int N, int Hi, int Wi, int C, int dt, hipStream_t s);
int launch_avgpool_fwd(const void* X, float* H, int N, int HW, int C, int dt, hipStream_t s);
int launch_avgpool_bwd(const float* dH, void* dX, int N, int HW, int C, int dt, hipStream_t s);

// ---- optional per-kernel-class HIP-event timing (bench.py roofline; off by default, zero cost when off) ----
enum { KC_GEMM_WIDE = 0, KC_GEMM_NARROW = 1, KC_WGRAD_WIDE = 2, KC_WGRAD_NARROW = 3, KC_COUNT = 4 };
void prof_begin(int kclass, double flops, int M, int N, int K, int taps, hipStream_t s);   // start event (no-op when disabled)
void prof_bytes(double bytes);                               // algorithmic HBM bytes of the launch just opened (roofline of the HBM-bound path)
void prof_end(hipStream_t s);                                // records the stop event

}  // namespace r3m
