// r3m_amd — extern "C" surface of libr3m_hip.so (declared in include/r3m_hip.h).
#include "common.h"
#include "augment_dev.h"
#include "../../include/r3m_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

namespace r3m {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

// R3M_TRACE=1: name every launch on stderr and synchronise after it (localises a faulting kernel; debugging only)
static int trace_mode() {
  const int t = R3M_ENV_INT("R3M_TRACE", 0) != 0;
  return t;
}

int check_launch(const char* what) {
  if (trace_mode()) {
    fprintf(stderr, "[r3m] %s\n", what);
    fflush(stderr);
    hipError_t se = hipDeviceSynchronize();
    if (se != hipSuccess) { set_last_error("%s: %s (at sync)", what, hipGetErrorString(se)); return (int)se; }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

int ensure_dyn_lds(DynLdsOptIn& cache, const void* fn, int bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) { set_last_error("%s: hipGetDevice failed", what); return 1; }
  const bool cached = dev < (int)(sizeof(cache.bytes) / sizeof(cache.bytes[0]));   // device indices beyond the cache: set it every time
  if (cached && bytes <= __atomic_load_n(&cache.bytes[dev], __ATOMIC_RELAXED)) return 0;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    set_last_error("%s: cannot reserve %d bytes of LDS", what, bytes);
    return 1;
  }
  if (cached) __atomic_store_n(&cache.bytes[dev], bytes, __ATOMIC_RELAXED);
  return 0;
}

// ---- per-kernel-class event timing ----
// Module state behind ONE mutex ("no mutable globals beyond lazily-initialised, mutex-guarded module state"):
// forward launches come from the caller's thread, backward launches from an autograd engine thread, collect() from the
// benchmark's thread. A launch's begin / bytes / end triple runs on one thread; the slot it fills is thread-local between
// begin and end, so two threads launching concurrently get two slots. The fast path (profiling off) is one relaxed load.
struct ProfEvent { hipEvent_t a, b; int kclass; double flops, bytes; int M, N, K, taps; bool closed; };
static std::atomic<bool> g_prof_on{false};
static std::atomic<unsigned> g_prof_classes{0xFFFFFFFFu};   // bit k: launches of class k are bracketed (r3m_profile_classes)
static std::mutex g_prof_mu;
static std::deque<ProfEvent> g_prof_pool;        // deque: growing never moves a slot another thread is filling
static size_t g_prof_used = 0;                   // guarded by g_prof_mu
static thread_local ProfEvent* t_prof_open = nullptr;

void prof_begin(int kclass, double flops, int M, int N, int K, int taps, hipStream_t s) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  if (!((g_prof_classes.load(std::memory_order_relaxed) >> kclass) & 1u)) return;
  ProfEvent* e = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_used == g_prof_pool.size()) {
      ProfEvent ne{};
      // timing-only events: no system-scope fence when they are recorded (hipEventDisableSystemFence) — the default flags write
      // back and invalidate the caches at every record, which is what a bracket mostly cost the stream and the kernel that follows
      if (hipEventCreateWithFlags(&ne.a, hipEventDisableSystemFence) != hipSuccess ||
          hipEventCreateWithFlags(&ne.b, hipEventDisableSystemFence) != hipSuccess) return;
      g_prof_pool.push_back(ne);
    }
    e = &g_prof_pool[g_prof_used++];
    e->closed = false;
  }
  e->kclass = kclass; e->flops = flops; e->bytes = 0.0; e->M = M; e->N = N; e->K = K; e->taps = taps;
  (void)hipEventRecord(e->a, s);
  t_prof_open = e;
}
// algorithmic HBM bytes of the launch opened by prof_begin (operands read once + results written once)
void prof_bytes(double bytes) {
  if (t_prof_open) t_prof_open->bytes = bytes;
}
void prof_end(hipStream_t s) {
  if (!t_prof_open) return;
  (void)hipEventRecord(t_prof_open->b, s);
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    t_prof_open->closed = true;
  }
  t_prof_open = nullptr;
}

int debug_occupancy(int* out4);
// engine.hip
struct Plan;
Plan* plan_create(int size, int F, int dtype);
int plan_dtype(Plan*);
int plan_forward(Plan& P, const float* x_nchw, const float* params, float* bufs, float* arena, float* h_out, int training,
                 hipStream_t s);
int plan_forward_src(Plan& P, const float* x_nchw, const FrameSource* crop, const float* params, float* bufs, float* arena,
                     float* h_out, int training, hipStream_t s);
int plan_backward(Plan& P, const float* dh, const float* params, float* grads, float* arena, int stage_begin, int stage_end,
                  int accumulate, int* gd_io, hipStream_t s);
int conv_forward_launch(const float* X, const float* W, float* Y, float* stats, const float* bias, int N, int Hi, int Wi, int Ci,
                        int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s);
int conv_dgrad_launch(const float* dY, const float* Wt, float* dX, const float* add0, const float* add1, const unsigned* addbits,
                      int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int flags, int dt, hipStream_t s);
int conv_wgrad_launch(const float* X, const float* dY, float* dW, float* partial_ws, int N, int Hi, int Wi, int Ci, int Co, int k,
                      int stride, int pad, int accumulate, int dt, hipStream_t s);
size_t conv_wgrad_ws_floats(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int dt);
// accessors implemented in engine.hip
int plan_out_dim(Plan*);
long long plan_num_params(Plan*);
long long plan_num_buffers(Plan*);
long long plan_arena_floats(Plan*);
int plan_num_tensors(Plan*);
int plan_tensor_info(Plan*, int i, char* name, int cap, int* kind, long long* offset, int* ndim, int* shape4);
int plan_stage_range(Plan*, int stage, long long* off, long long* count);
void plan_destroy(Plan*);
int* plan_gd(Plan*);
int plan_set_fuse_bnred(Plan*, int on);
int plan_set_bn_pair(Plan*, int on);
// loss.hip / adam.hip
size_t loss_workspace_floats(int B);
int launch_tcn_lp_loss(const float* alle, const int* perm, const int* iperm, float* dalle, float* ws, int B, int D, int l2dist,
                       float l2w, float l1w, float tcnw, hipStream_t s);
int launch_lang_infonce(const float* scores, const float* mask, float* dscore, float* ws, int B, float langw, hipStream_t s);
int launch_loss_finalize(float* ws, int B, int have_lang, float* metrics, float l2w, float l1w, float tcnw, float langw,
                         hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                long long step, float grad_scale, hipStream_t s);
int launch_sgd(float* p, const float* g, float* momentum_buf, long long n, double lr, double momentum, double dampening,
               double weight_decay, int nesterov, long long step, float grad_scale, hipStream_t s);
// augment.hip
int launch_crop_resize(const void* in, int in_is_u8, const int* boxes, float* out, long long N, int C, int Hi, int Wi, int Ho,
                       int Wo, int frames_per_box, hipStream_t s);
int launch_resize_crop(const void* in, int in_is_u8, float* out, long long N, int C, int Hi, int Wi, int full_Ho, int full_Wo,
                       int top, int left, int Ho, int Wo, hipStream_t s);
// lang.hip
long long langrew_num_params(int D, int H, int LD);
size_t langrew_ws_floats(int B, int D, int H, int LD);
int langrew_forward(const float* alle, const float* feats, const int* perm, const float* params, float* scores, float* ws, int B,
                    int D, int H, int LD, int dt, hipStream_t s);
int langrew_backward(const float* dscore, const int* iperm, const float* params, float* grads, float* dalle, float* ws, int B, int D,
                     int H, int LD, int accumulate, int dt, hipStream_t s);
size_t langrew_call_ws_floats(int R, int D, int H, int LD);
int langrew_call_forward(const float* e0, const float* eg, const float* le, const float* params, float* score, float* ws, int R,
                         int D, int H, int LD, hipStream_t s);
int langrew_call_backward(const float* dscore, const float* params, float* grads, float* de0, float* deg, float* dle, float* ws,
                          int R, int D, int H, int LD, int accumulate, hipStream_t s);

}  // namespace r3m

using namespace r3m;

#define S(x) reinterpret_cast<hipStream_t>(x)
#define PLAN(h) reinterpret_cast<r3m::Plan*>(h)

extern "C" {

int r3m_abi_version(void) { return 1; }

int r3m_debug_occupancy(int* out4) { return debug_occupancy(out4); }
__global__ __launch_bounds__(256) void occupy_kernel(long long ticks) {   // wall_clock64(): constant-rate counter (100 MHz)
  extern __shared__ char occ_lds[];
  occ_lds[threadIdx.x] = 1;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int r3m_debug_occupy(int blocks, int lds_bytes, double milliseconds, r3m_stream_t stream) {
  R3M_REQUIRE(blocks > 0 && lds_bytes >= 256 && lds_bytes <= 160 * 1024 && milliseconds >= 0.0, "debug_occupy: bad arguments");
  static r3m::DynLdsOptIn oi;
  if (int e = r3m::ensure_dyn_lds(oi, reinterpret_cast<const void*>(occupy_kernel), lds_bytes, "debug_occupy")) return e;
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), (long long)(milliseconds * 1e5));
  return r3m::check_launch("debug_occupy");
}
int r3m_debug_set_dynamic_tiles(int on) { return r3m::gg_set_dynamic_tiles(on); }
int r3m_debug_set_pw16(int mode) { return r3m::pw16_set_mode(mode); }
int r3m_debug_set_conv3x3_bf16(int mode) { return r3m::row16_set_mode(mode); }
int r3m_debug_set_fused_inference(int on) { return r3m::engine_set_fused_inference(on); }
int r3m_debug_conv_route(int N, int H, int W, int Ci, int Co, int k, int stride, int pad, int dgrad, int flags, int mask_bits, int dtype,
                         int* routes, int cap) {
  R3M_REQUIRE(routes && cap >= 1, "debug_conv_route: routes buffer");
  // no launch happens: the pointers only have to look like the ones the epilogue flags ask for
  const unsigned* some_bits = reinterpret_cast<const unsigned*>(static_cast<uintptr_t>(64));
  r3m::gg_route_record_begin(routes, cap);
  int rc;
  if (!dgrad) {
    rc = r3m::conv_forward_launch(nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Ci, Co, k, stride, pad, flags, dtype, nullptr);
  } else {
    r3m::BnRedArgs br{nullptr, mask_bits ? some_bits : nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    const bool bn = (flags & 64) != 0;      // EPI_BNRED: requested through the BnRedArgs, as the engine does
    rc = r3m::conv_dgrad_launch_br(nullptr, nullptr, nullptr, nullptr, nullptr, (flags & 4) ? some_bits : nullptr, N, H, W, Ci, Co, k, stride, pad,
                                   flags & ~64, dtype, bn ? &br : nullptr, nullptr);
  }
  const int n = r3m::gg_route_record_end();
  return rc ? -1 : n;
}
void r3m_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on.store(on != 0, std::memory_order_relaxed);
  if (!on) g_prof_used = 0;
}
// Which kernel classes prof_begin brackets (bit k = class k; default all). A pair of event records costs the stream a few microseconds:
// bench.py brackets only the dominant class inside its timed steps (2.8 -> ~1.6 ms per ResNet-50 step). Returns the old mask.
unsigned r3m_profile_classes(unsigned mask) { return g_prof_classes.exchange(mask, std::memory_order_relaxed); }
// optional: every launch since the last collect as CSV rows (class,M,N,K,taps,ms,gflop) into a host file
static FILE* g_prof_dump = nullptr;              // guarded by g_prof_mu
static double g_prof_bytes[KC_COUNT];            // guarded by g_prof_mu
int r3m_profile_dump_to(const char* path) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_dump) { fclose(g_prof_dump); g_prof_dump = nullptr; }
  if (path && *path) {
    g_prof_dump = fopen(path, "w");
    if (!g_prof_dump) { set_last_error("profile_dump_to: cannot open %s", path); return 1; }
    fprintf(g_prof_dump, "class,M,N,K,taps,ms,gflop,alg_mbytes\n");
  }
  return 0;
}
int r3m_profile_collect_bytes(double* bytes) {   // algorithmic bytes per class of the launches seen by the LAST collect()
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int k = 0; k < KC_COUNT; ++k) bytes[k] = g_prof_bytes[k];
  return 0;
}
int r3m_profile_collect(double* ms, long long* launches, double* flops) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int k = 0; k < KC_COUNT; ++k) { ms[k] = 0.0; launches[k] = 0; flops[k] = 0.0; g_prof_bytes[k] = 0.0; }
  for (size_t i = 0; i < g_prof_used; ++i) {
    ProfEvent& e = g_prof_pool[i];
    if (!e.closed) continue;                       // a launch still between begin and end on another thread
    if (hipEventSynchronize(e.b) != hipSuccess) { set_last_error("profile_collect: event sync failed"); return 1; }
    float t = 0.f;
    if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) { set_last_error("profile_collect: elapsed failed"); return 1; }
    ms[e.kclass] += t; launches[e.kclass] += 1; flops[e.kclass] += e.flops; g_prof_bytes[e.kclass] += e.bytes;
    if (g_prof_dump) fprintf(g_prof_dump, "%d,%d,%d,%d,%d,%.4f,%.3f,%.3f\n", e.kclass, e.M, e.N, e.K, e.taps, t, e.flops * 1e-9, e.bytes * 1e-6);
  }
  g_prof_used = 0;
  if (g_prof_dump) fflush(g_prof_dump);
  return 0;
}
const char* r3m_last_error(void) { return g_err; }

r3m_resnet_t r3m_resnet_create(int size, int frames) { return reinterpret_cast<r3m_resnet_t>(plan_create(size, frames, DT_F32)); }
r3m_resnet_t r3m_resnet_create_dt(int size, int frames, int dtype) { return reinterpret_cast<r3m_resnet_t>(plan_create(size, frames, dtype)); }
int r3m_resnet_dtype(r3m_resnet_t h) { return plan_dtype(PLAN(h)); }
void r3m_resnet_destroy(r3m_resnet_t h) { if (h) plan_destroy(PLAN(h)); }
int r3m_resnet_out_dim(r3m_resnet_t h) { return plan_out_dim(PLAN(h)); }
long long r3m_resnet_num_params(r3m_resnet_t h) { return plan_num_params(PLAN(h)); }
long long r3m_resnet_num_buffers(r3m_resnet_t h) { return plan_num_buffers(PLAN(h)); }
long long r3m_resnet_arena_bytes(r3m_resnet_t h) { return plan_arena_floats(PLAN(h)) * 4; }
int r3m_resnet_num_tensors(r3m_resnet_t h) { return plan_num_tensors(PLAN(h)); }
int r3m_resnet_tensor_info(r3m_resnet_t h, int i, char* name, int name_cap, int* kind, long long* offset, int* ndim, int* shape4) {
  return plan_tensor_info(PLAN(h), i, name, name_cap, kind, offset, ndim, shape4);
}
int r3m_resnet_stage_range(r3m_resnet_t h, int stage, long long* offset, long long* count) {
  return plan_stage_range(PLAN(h), stage, offset, count);
}
int r3m_resnet_forward(r3m_resnet_t h, const float* x, const float* params, float* buffers, void* arena, float* h_out, int training,
                       r3m_stream_t stream) {
  R3M_REQUIRE(h && x && params && buffers && arena && h_out, "resnet_forward: null argument");
  R3M_REQUIRE(training >= 0 && training <= 2, "resnet_forward: training=%d (0 eval, 1 train, 2 inference)", training);
  return plan_forward(*PLAN(h), x, params, buffers, static_cast<float*>(arena), h_out, training, S(stream));
}
int r3m_resnet_forward_crop(r3m_resnet_t h, const void* frames, int frames_are_u8, const int* boxes, int frames_per_box, int Hi, int Wi,
                            const float* params, float* buffers, void* arena, float* h_out, int training, r3m_stream_t stream) {
  R3M_REQUIRE(h && frames && boxes && params && buffers && arena && h_out, "resnet_forward_crop: null argument");
  R3M_REQUIRE(frames_per_box >= 1 && Hi >= 1 && Wi >= 1, "resnet_forward_crop: frames_per_box=%d, frames %dx%d", frames_per_box, Hi, Wi);
  R3M_REQUIRE(training >= 0 && training <= 2, "resnet_forward_crop: training=%d (0 eval, 1 train, 2 inference)", training);
  const FrameSource src{frames, frames_are_u8, boxes, frames_per_box, Hi, Wi};
  return plan_forward_src(*PLAN(h), nullptr, &src, params, buffers, static_cast<float*>(arena), h_out, training, S(stream));
}
int r3m_resnet_set_fused_bn_reduce(r3m_resnet_t h, int on) { return h ? plan_set_fuse_bnred(PLAN(h), on) : -1; }
int r3m_resnet_set_bn_pair(r3m_resnet_t h, int on) { return h ? plan_set_bn_pair(PLAN(h), on) : -1; }
int r3m_resnet_backward(r3m_resnet_t h, const float* dh, const float* params, float* grads, void* arena, int stage_begin,
                        int stage_end, int accumulate, r3m_stream_t stream) {
  R3M_REQUIRE(h && dh && params && grads && arena, "resnet_backward: null argument");
  R3M_REQUIRE(0 <= stage_begin && stage_begin <= stage_end && stage_end <= 4, "resnet_backward: stages [%d,%d)", stage_begin, stage_end);
  return plan_backward(*PLAN(h), dh, params, grads, static_cast<float*>(arena), stage_begin, stage_end, accumulate, plan_gd(PLAN(h)),
                       S(stream));
}

int r3m_conv2d_stats_rows(int N, int Hi, int Wi, int Co, int k, int stride, int pad) {
  const int Ho = (Hi + 2 * pad - k) / stride + 1, Wo = (Wi + 2 * pad - k) / stride + 1;
  return gather_gemm_grid_m(N * Ho * Wo, Co);
}
static int check_dt(int dtype, const char* what) {
  R3M_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "%s: dtype %d (0 = fp32, 1 = bf16)", what, dtype);
  return 0;
}
#define FP(x) static_cast<const float*>(x)
#define FPM(x) static_cast<float*>(x)
int r3m_convert_bf16(const float* src, void* dst, long long n, r3m_stream_t stream) {
  R3M_REQUIRE(src && dst, "convert_bf16: null argument");
  return launch_convert_bf16(src, dst, n, S(stream));
}
int r3m_conv2d_fwd_dt(const void* x, const void* w, void* y, float* stats, int N, int Hi, int Wi, int Ci, int Co, int k, int stride,
                      int pad, int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "conv2d_fwd")) return 1;
  return conv_forward_launch(FP(x), FP(w), FPM(y), stats, nullptr, N, Hi, Wi, Ci, Co, k, stride, pad, stats ? EPI_STATS : 0, dtype, S(stream));
}
int r3m_conv2d_fwd(const float* x, const float* w, float* y, float* stats, int N, int Hi, int Wi, int Ci, int Co, int k, int stride,
                   int pad, r3m_stream_t stream) {
  return r3m_conv2d_fwd_dt(x, w, y, stats, N, Hi, Wi, Ci, Co, k, stride, pad, DT_F32, stream);
}
size_t r3m_conv2d_dgrad_workspace_bytes(int Ci, int Co, int k) { return (size_t)Ci * Co * k * k * 4; }
int r3m_conv2d_dgrad_dt(const void* dy, const float* w, void* dx, void* ws, size_t ws_bytes, int N, int Hi, int Wi, int Ci, int Co,
                        int k, int stride, int pad, int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "conv2d_dgrad")) return 1;
  R3M_REQUIRE(ws_bytes >= r3m_conv2d_dgrad_workspace_bytes(Ci, Co, k), "conv2d_dgrad: workspace too small");
  float* Wt = static_cast<float*>(ws);
  if (dtype == DT_BF16) { if (int e = launch_transpose_w_bf16(w, Wt, Co, k * k, Ci, S(stream))) return e; }
  else if (int e = launch_transpose_w(w, Wt, Co, k * k, Ci, S(stream))) return e;
  return conv_dgrad_launch(FP(dy), Wt, FPM(dx), nullptr, nullptr, nullptr, N, Hi, Wi, Ci, Co, k, stride, pad, 0, dtype, S(stream));
}
int r3m_conv2d_dgrad_bnred_rows(int N, int Hi, int Wi, int stride) {
  if (stride == 1) return bnred_partial_rows((long long)N * Hi * Wi);
  int rows = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int Hg = (Hi - py + 1) / 2, Wg = (Wi - px + 1) / 2;
      if (Hg > 0 && Wg > 0) rows += bnred_partial_rows((long long)N * Hg * Wg);
    }
  return rows;
}
int r3m_conv2d_dgrad_bnred_dt(const void* dy, const float* w, void* dx, void* ws, size_t ws_bytes, int N, int Hi, int Wi, int Ci, int Co,
                              int k, int stride, int pad, const void* residual_grad, const unsigned* residual_bits, const void* bn_y,
                              const unsigned* bn_bits, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                              float* partials, int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "conv2d_dgrad_bnred")) return 1;
  R3M_REQUIRE(dy && w && dx && ws && bn_y && bn_mean && partials, "conv2d_dgrad_bnred: null argument");
  R3M_REQUIRE(bn_bits || (bn_scale && bn_shift), "conv2d_dgrad_bnred: pass the mask bits or scale + shift to recompute the mask");
  R3M_REQUIRE(!residual_grad || residual_bits, "conv2d_dgrad_bnred: a residual gradient needs its mask bits");
  R3M_REQUIRE(!(k == 1 && stride == 2), "conv2d_dgrad_bnred: 1x1 stride-2 (parity classes without taps) is not supported");
  R3M_REQUIRE(ws_bytes >= r3m_conv2d_dgrad_workspace_bytes(Ci, Co, k), "conv2d_dgrad_bnred: workspace too small");
  float* Wt = static_cast<float*>(ws);
  if (dtype == DT_BF16) { if (int e = launch_transpose_w_bf16(w, Wt, Co, k * k, Ci, S(stream))) return e; }
  else if (int e = launch_transpose_w(w, Wt, Co, k * k, Ci, S(stream))) return e;
  BnRedArgs br{FP(bn_y), bn_bits, bn_scale, bn_shift, bn_mean, partials, 0};
  return conv_dgrad_launch_br(FP(dy), Wt, FPM(dx), FP(residual_grad), nullptr, residual_bits, N, Hi, Wi, Ci, Co, k, stride, pad,
                              residual_grad ? EPI_MASKED_ADD : 0, dtype, &br, S(stream));
}
int r3m_conv2d_dgrad(const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes, int N, int Hi, int Wi, int Ci, int Co,
                     int k, int stride, int pad, r3m_stream_t stream) {
  return r3m_conv2d_dgrad_dt(dy, w, dx, ws, ws_bytes, N, Hi, Wi, Ci, Co, k, stride, pad, DT_F32, stream);
}
size_t r3m_conv2d_wgrad_workspace_bytes_dt(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int dtype) {
  return conv_wgrad_ws_floats(N, Hi, Wi, Ci, Co, k, stride, pad, dtype) * 4;
}
size_t r3m_conv2d_wgrad_workspace_bytes(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad) {
  return r3m_conv2d_wgrad_workspace_bytes_dt(N, Hi, Wi, Ci, Co, k, stride, pad, DT_F32);
}
int r3m_conv2d_wgrad_dt(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, int N, int Hi, int Wi, int Ci, int Co,
                        int k, int stride, int pad, int accumulate, int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "conv2d_wgrad")) return 1;
  R3M_REQUIRE(ws_bytes >= r3m_conv2d_wgrad_workspace_bytes_dt(N, Hi, Wi, Ci, Co, k, stride, pad, dtype), "conv2d_wgrad: workspace too small");
  return conv_wgrad_launch(FP(x), FP(dy), dw, static_cast<float*>(ws), N, Hi, Wi, Ci, Co, k, stride, pad, accumulate, dtype, S(stream));
}
int r3m_conv2d_wgrad(const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, int N, int Hi, int Wi, int Ci, int Co,
                     int k, int stride, int pad, int accumulate, r3m_stream_t stream) {
  return r3m_conv2d_wgrad_dt(x, dy, dw, ws, ws_bytes, N, Hi, Wi, Ci, Co, k, stride, pad, accumulate, DT_F32, stream);
}
int r3m_stem_prep(const float* x, float* xn, int frames, r3m_stream_t stream) {
  R3M_REQUIRE(x && xn, "stem_prep: null argument");
  return launch_stem_prep(x, xn, frames, S(stream));
}
int r3m_stem_conv_fwd_dt(const float* xn, const float* w_ohwi, void* y, float* stats, int frames, int dtype, r3m_stream_t stream) {
  R3M_REQUIRE(xn && w_ohwi && y, "stem_conv_fwd: null argument");
  if (check_dt(dtype, "stem_conv_fwd")) return 1;
  return launch_stem_fwd(xn, w_ohwi, y, stats, frames, dtype, S(stream));
}
int r3m_stem_conv_fwd(const float* xn, const float* w_ohwi, float* y, float* stats, int frames, r3m_stream_t stream) {
  return r3m_stem_conv_fwd_dt(xn, w_ohwi, y, stats, frames, DT_F32, stream);
}
size_t r3m_stem_conv_wgrad_workspace_bytes(void) { return (stem_wgrad_ws_floats() + 64 * 160) * 4; }
int r3m_stem_conv_wgrad_dt(const float* xn, const void* dy, float* dw_ohwi, void* ws, size_t ws_bytes, int frames, int accumulate,
                           int dtype, r3m_stream_t stream) {
  R3M_REQUIRE(xn && dy && dw_ohwi && ws, "stem_conv_wgrad: null argument");
  if (check_dt(dtype, "stem_conv_wgrad")) return 1;
  R3M_REQUIRE(ws_bytes >= r3m_stem_conv_wgrad_workspace_bytes(), "stem_conv_wgrad: workspace too small");
  return launch_stem_wgrad(xn, dy, dw_ohwi, static_cast<float*>(ws), frames, accumulate, dtype, S(stream));
}
int r3m_stem_conv_wgrad(const float* xn, const float* dy, float* dw_ohwi, void* ws, size_t ws_bytes, int frames, int accumulate,
                        r3m_stream_t stream) {
  return r3m_stem_conv_wgrad_dt(xn, dy, dw_ohwi, ws, ws_bytes, frames, accumulate, DT_F32, stream);
}

// stem on the bf16 MFMA (what bf16 plans run): padded bf16 image of the normalised frames, forward, weight gradient
size_t r3m_stem_xn16_bytes(int frames) { return stem_xn16_bytes(frames); }
int r3m_stem_prep_bf16(const float* x, void* xn16, int frames, r3m_stream_t stream) {
  R3M_REQUIRE(x && xn16, "stem_prep_bf16: null argument");
  return launch_stem_prep16(x, xn16, frames, S(stream));
}
int r3m_stem_prep_crop(const void* frames, int frames_are_u8, const int* boxes, int frames_per_box, int Hi, int Wi, void* xn_out, int F,
                       int dtype, r3m_stream_t stream) {
  R3M_REQUIRE(frames && boxes && xn_out, "stem_prep_crop: null argument");
  R3M_REQUIRE(frames_per_box >= 1 && Hi >= 1 && Wi >= 1 && F >= 1, "stem_prep_crop: frames_per_box=%d, %d frames of %dx%d", frames_per_box, F, Hi, Wi);
  if (check_dt(dtype, "stem_prep_crop")) return 1;
  const FrameSource src{frames, frames_are_u8, boxes, frames_per_box, Hi, Wi};
  return dtype == DT_BF16 ? launch_stem_prep16_crop(src, xn_out, F, S(stream)) : launch_stem_prep_crop(src, static_cast<float*>(xn_out), F, S(stream));
}
int r3m_stem_conv_fwd_bf16(const void* xn16, const float* w_ohwi, void* y, float* stats, int frames, r3m_stream_t stream) {
  R3M_REQUIRE(xn16 && w_ohwi && y, "stem_conv_fwd_bf16: null argument");
  return launch_stem_fwd16(xn16, w_ohwi, y, stats, frames, S(stream));
}
size_t r3m_stem_conv_wgrad_bf16_workspace_bytes(void) { return stem_wgrad16_ws_floats() * 4; }
int r3m_stem_conv_wgrad_bf16(const void* xn16, const void* dy, float* dw_ohwi, void* ws, size_t ws_bytes, int frames, int accumulate,
                             r3m_stream_t stream) {
  R3M_REQUIRE(xn16 && dy && dw_ohwi && ws, "stem_conv_wgrad_bf16: null argument");
  R3M_REQUIRE(ws_bytes >= r3m_stem_conv_wgrad_bf16_workspace_bytes(), "stem_conv_wgrad_bf16: workspace too small");
  return launch_stem_wgrad16(xn16, dy, dw_ohwi, static_cast<float*>(ws), frames, accumulate, S(stream));
}

// workspace: [partials: bn_bwd_partial_rows*2*C floats][acc: 64*2*C doubles]
static size_t bn_acc_off(long long rows, int C) {
  size_t p = (size_t)bn_bwd_partial_rows(rows, C, DT_F32) * 2 * C * 4;   // the fp32 geometry has the most rows
  return (p + 255) / 256 * 256;
}
static size_t bn_c12_off(long long rows, int C) { return bn_acc_off(rows, C) + bn_acc_bytes(C); }
size_t r3m_bn_workspace_bytes(long long rows, int C) { return bn_c12_off(rows, C) + (size_t)2 * C * 4; }

int r3m_bn_train_coeffs(const float* stats, int stats_rows, long long count, const float* gamma, const float* beta, float* rm,
                        float* rv, float momentum, float eps, float* coef, void* ws, size_t ws_bytes, int C, r3m_stream_t stream) {
  R3M_REQUIRE(ws_bytes >= bn_acc_bytes(C), "bn_train_coeffs: workspace too small (need %zu)", bn_acc_bytes(C));
  double* acc = static_cast<double*>(ws);
  if (int e = launch_bn_stats_reduce(stats, stats_rows, C, acc, S(stream))) return e;
  return launch_bn_finalize_rows(acc, stats_rows, count, gamma, beta, rm, rv, momentum, eps, coef, coef + C, coef + 2 * C,
                                 coef + 3 * C, C, S(stream));
}
int r3m_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* coef, int C,
                       r3m_stream_t stream) {
  return launch_bn_eval_coeffs(gamma, beta, rm, rv, eps, coef, coef + C, coef + 2 * C, coef + 3 * C, C, S(stream));
}
int r3m_bn_act_fwd_dt(const void* y, const float* coef, const void* r, const void* y2, const float* coef2, void* z, long long rows,
                      int C, int relu, unsigned* maskbits, int dtype, r3m_stream_t stream) {
  R3M_REQUIRE(!(r && y2), "bn_act_fwd: pass either r (identity) or y2/coef2 (downsample), not both");
  if (check_dt(dtype, "bn_act_fwd")) return 1;
  if (y2) return launch_bn_act_fwd(y, coef + 2 * C, coef + 3 * C, y2, coef2 + 2 * C, coef2 + 3 * C, z, rows, C, relu, maskbits, dtype, S(stream));
  return launch_bn_act_fwd(y, coef + 2 * C, coef + 3 * C, r, nullptr, nullptr, z, rows, C, relu, maskbits, dtype, S(stream));
}
int r3m_bn_act_fwd(const float* y, const float* coef, const float* r, const float* y2, const float* coef2, float* z, long long rows,
                   int C, int relu, unsigned* maskbits, r3m_stream_t stream) {
  return r3m_bn_act_fwd_dt(y, coef, r, y2, coef2, z, rows, C, relu, maskbits, DT_F32, stream);
}
int r3m_bn_bwd_dt(const void* dz, const void* zmask, const unsigned* zbits, const void* y, const float* coef, float* dgamma,
                  float* dbeta, void* dy, void* ws, size_t ws_bytes, long long rows, int C, int use_batch_stats, int accumulate,
                  int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "bn_bwd")) return 1;
  R3M_REQUIRE(ws_bytes >= r3m_bn_workspace_bytes(rows, C), "bn_bwd: workspace too small (need %zu)", r3m_bn_workspace_bytes(rows, C));
  float* partial = static_cast<float*>(ws);
  double* acc = reinterpret_cast<double*>(static_cast<char*>(ws) + bn_acc_off(rows, C));
  float* c12 = reinterpret_cast<float*>(static_cast<char*>(ws) + bn_c12_off(rows, C));
  const float *mean = coef, *invstd = coef + C, *scale = coef + 2 * C, *shift = coef + 3 * C;
  if (int e = launch_bn_bwd_reduce(dz, zmask, zbits, y, scale, shift, mean, invstd, partial, rows, C, dtype, S(stream))) return e;
  const int prow = bn_bwd_partial_rows(rows, C, dtype);
  if (int e = launch_bn_stats_reduce(partial, prow, C, acc, S(stream))) return e;
  if (int e = launch_bn_bwd_finalize_rows(acc, prow, rows, use_batch_stats, dgamma, dbeta, c12, c12 + C, accumulate, C, S(stream))) return e;
  return launch_bn_bwd_apply(dz, zmask, zbits, y, scale, shift, mean, invstd, c12, c12 + C, dy, rows, C, dtype, S(stream));
}
int r3m_bn_bwd(const float* dz, const float* zmask, const unsigned* zbits, const float* y, const float* coef, float* dgamma,
               float* dbeta, float* dy, void* ws, size_t ws_bytes, long long rows, int C, int use_batch_stats, int accumulate, r3m_stream_t stream) {
  return r3m_bn_bwd_dt(dz, zmask, zbits, y, coef, dgamma, dbeta, dy, ws, ws_bytes, rows, C, use_batch_stats, accumulate, DT_F32, stream);
}
// stem tail fused: BatchNorm + ReLU + MaxPool(3,2,1) forward, and its backward (MaxPool gather inside both BN-backward passes)
int r3m_bn_relu_maxpool_fwd_dt(const void* y, const float* coef, void* p, unsigned char* am, int N, int Hi, int Wi, int C, int dtype,
                               r3m_stream_t stream) {
  R3M_REQUIRE(y && coef && p && am, "bn_relu_maxpool_fwd: null argument");
  if (check_dt(dtype, "bn_relu_maxpool_fwd")) return 1;
  return launch_bn_relu_maxpool_fwd(y, coef + 2 * C, coef + 3 * C, p, am, N, Hi, Wi, C, dtype, S(stream));
}
int r3m_bn_maxpool_bwd_dt(const void* dp, const unsigned char* am, const void* y, const float* coef, float* dgamma, float* dbeta, void* dy,
                          void* ws, size_t ws_bytes, int N, int Hi, int Wi, int C, int use_batch_stats, int accumulate, int dtype,
                          r3m_stream_t stream) {
  R3M_REQUIRE(dp && am && y && coef && dgamma && dbeta && dy && ws, "bn_maxpool_bwd: null argument");
  if (check_dt(dtype, "bn_maxpool_bwd")) return 1;
  const long long rows = (long long)N * Hi * Wi;
  R3M_REQUIRE(ws_bytes >= r3m_bn_workspace_bytes(rows, C), "bn_maxpool_bwd: workspace too small (need %zu)", r3m_bn_workspace_bytes(rows, C));
  float* partial = static_cast<float*>(ws);
  double* acc = reinterpret_cast<double*>(static_cast<char*>(ws) + bn_acc_off(rows, C));
  float* c12 = reinterpret_cast<float*>(static_cast<char*>(ws) + bn_c12_off(rows, C));
  const float *mean = coef, *invstd = coef + C, *scale = coef + 2 * C, *shift = coef + 3 * C;
  if (int e = launch_bn_bwd_reduce_pool(dp, am, y, scale, shift, mean, invstd, partial, N, Hi, Wi, C, dtype, S(stream))) return e;
  const int prow = bn_bwd_pool_partial_rows(N, Hi, Wi, C);
  if (int e = launch_bn_stats_reduce(partial, prow, C, acc, S(stream))) return e;
  if (int e = launch_bn_bwd_finalize_rows(acc, prow, rows, use_batch_stats, dgamma, dbeta, c12, c12 + C, accumulate, C, S(stream))) return e;
  return launch_bn_bwd_apply_pool(dp, am, y, scale, shift, mean, invstd, c12, c12 + C, dy, N, Hi, Wi, C, dtype, S(stream));
}
int r3m_maxpool_fwd_dt(const void* z, void* p, unsigned char* am, int N, int Hi, int Wi, int C, int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "maxpool_fwd")) return 1;
  return launch_maxpool_fwd(z, p, am, N, Hi, Wi, C, dtype, S(stream));
}
int r3m_maxpool_bwd_dt(const void* dp, const unsigned char* am, void* dz, int N, int Hi, int Wi, int C, int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "maxpool_bwd")) return 1;
  return launch_maxpool_bwd(dp, am, dz, N, Hi, Wi, C, dtype, S(stream));
}
int r3m_avgpool_fwd_dt(const void* x, float* h, int N, int HW, int C, int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "avgpool_fwd")) return 1;
  return launch_avgpool_fwd(x, h, N, HW, C, dtype, S(stream));
}
int r3m_avgpool_bwd_dt(const float* dh, void* dx, int N, int HW, int C, int dtype, r3m_stream_t stream) {
  if (check_dt(dtype, "avgpool_bwd")) return 1;
  return launch_avgpool_bwd(dh, dx, N, HW, C, dtype, S(stream));
}
int r3m_maxpool_fwd(const float* z, float* p, unsigned char* am, int N, int Hi, int Wi, int C, r3m_stream_t stream) {
  return launch_maxpool_fwd(z, p, am, N, Hi, Wi, C, DT_F32, S(stream));
}
int r3m_maxpool_bwd(const float* dp, const unsigned char* am, float* dz, int N, int Hi, int Wi, int C, r3m_stream_t stream) {
  return launch_maxpool_bwd(dp, am, dz, N, Hi, Wi, C, DT_F32, S(stream));
}
int r3m_avgpool_fwd(const float* x, float* h, int N, int HW, int C, r3m_stream_t stream) { return launch_avgpool_fwd(x, h, N, HW, C, DT_F32, S(stream)); }
int r3m_avgpool_bwd(const float* dh, float* dx, int N, int HW, int C, r3m_stream_t stream) { return launch_avgpool_bwd(dh, dx, N, HW, C, DT_F32, S(stream)); }

int r3m_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int relu, r3m_stream_t stream) {
  return conv_forward_launch(x, w, y, nullptr, bias, M, 1, 1, K, N, 1, 1, 0, (bias ? EPI_BIAS : 0) | (relu ? EPI_RELU : 0), DT_F32, S(stream));
}

long long r3m_langrew_num_params(int D, int hidden, int lang_dim) { return langrew_num_params(D, hidden, lang_dim); }
size_t r3m_langrew_workspace_bytes(int B, int D, int hidden, int lang_dim) { return langrew_ws_floats(B, D, hidden, lang_dim) * 4; }
int r3m_langrew_forward_dt(const float* alle, const float* feats, const int* perm, const float* params, float* scores, void* ws,
                           size_t ws_bytes, int B, int D, int hidden, int lang_dim, int dtype, r3m_stream_t stream) {
  R3M_REQUIRE(alle && feats && perm && params && scores && ws, "langrew_forward: null argument");
  R3M_REQUIRE(ws_bytes >= r3m_langrew_workspace_bytes(B, D, hidden, lang_dim), "langrew_forward: workspace too small");
  if (check_dt(dtype, "langrew_forward")) return 1;
  return langrew_forward(alle, feats, perm, params, scores, static_cast<float*>(ws), B, D, hidden, lang_dim, dtype, S(stream));
}
int r3m_langrew_backward_dt(const float* dscore, const int* iperm, const float* params, float* grads, float* dalle, void* ws,
                            size_t ws_bytes, int B, int D, int hidden, int lang_dim, int accumulate, int dtype, r3m_stream_t stream) {
  R3M_REQUIRE(dscore && iperm && params && grads && ws, "langrew_backward: null argument");
  R3M_REQUIRE(ws_bytes >= r3m_langrew_workspace_bytes(B, D, hidden, lang_dim), "langrew_backward: workspace too small");
  if (check_dt(dtype, "langrew_backward")) return 1;
  return langrew_backward(dscore, iperm, params, grads, dalle, static_cast<float*>(ws), B, D, hidden, lang_dim, accumulate, dtype, S(stream));
}
int r3m_langrew_forward(const float* alle, const float* feats, const int* perm, const float* params, float* scores, void* ws,
                        size_t ws_bytes, int B, int D, int hidden, int lang_dim, r3m_stream_t stream) {
  return r3m_langrew_forward_dt(alle, feats, perm, params, scores, ws, ws_bytes, B, D, hidden, lang_dim, DT_F32, stream);
}
int r3m_langrew_backward(const float* dscore, const int* iperm, const float* params, float* grads, float* dalle, void* ws,
                         size_t ws_bytes, int B, int D, int hidden, int lang_dim, int accumulate, r3m_stream_t stream) {
  return r3m_langrew_backward_dt(dscore, iperm, params, grads, dalle, ws, ws_bytes, B, D, hidden, lang_dim, accumulate, DT_F32, stream);
}

size_t r3m_langrew_call_workspace_bytes(int R, int D, int hidden, int lang_dim) { return langrew_call_ws_floats(R, D, hidden, lang_dim) * 4; }
int r3m_langrew_call_forward(const float* e0, const float* eg, const float* le, const float* params, float* score, void* ws,
                             size_t ws_bytes, int R, int D, int hidden, int lang_dim, r3m_stream_t stream) {
  R3M_REQUIRE(e0 && eg && le && params && score && ws, "langrew_call_forward: null argument");
  R3M_REQUIRE(R > 0, "langrew_call_forward: R=%d rows", R);
  R3M_REQUIRE(ws_bytes >= r3m_langrew_call_workspace_bytes(R, D, hidden, lang_dim), "langrew_call_forward: workspace too small");
  return langrew_call_forward(e0, eg, le, params, score, static_cast<float*>(ws), R, D, hidden, lang_dim, S(stream));
}
int r3m_langrew_call_backward(const float* dscore, const float* params, float* grads, float* de0, float* deg, float* dle, void* ws,
                              size_t ws_bytes, int R, int D, int hidden, int lang_dim, int accumulate, r3m_stream_t stream) {
  R3M_REQUIRE(dscore && params && grads && ws, "langrew_call_backward: null argument");
  R3M_REQUIRE(R > 0, "langrew_call_backward: R=%d rows", R);
  R3M_REQUIRE(ws_bytes >= r3m_langrew_call_workspace_bytes(R, D, hidden, lang_dim), "langrew_call_backward: workspace too small");
  return langrew_call_backward(dscore, params, grads, de0, deg, dle, static_cast<float*>(ws), R, D, hidden, lang_dim, accumulate,
                               S(stream));
}

int r3m_crop_resize(const void* frames, int frames_are_u8, const int* boxes, float* out, long long N, int C, int Hi, int Wi, int Ho,
                    int Wo, int frames_per_box, r3m_stream_t stream) {
  R3M_REQUIRE(frames && boxes && out, "crop_resize: null argument");
  return launch_crop_resize(frames, frames_are_u8, boxes, out, N, C, Hi, Wi, Ho, Wo, frames_per_box, S(stream));
}

int r3m_resize_crop(const void* frames, int frames_are_u8, float* out, long long N, int C, int Hi, int Wi, int resized_h, int resized_w,
                    int top, int left, int Ho, int Wo, r3m_stream_t stream) {
  R3M_REQUIRE(frames && out, "resize_crop: null argument");
  return launch_resize_crop(frames, frames_are_u8, out, N, C, Hi, Wi, resized_h, resized_w, top, left, Ho, Wo, S(stream));
}

size_t r3m_loss_workspace_bytes(int B) { return loss_workspace_floats(B) * 4; }
int r3m_loss_tcn_lp(const float* alle, const int* perm, const int* iperm, float* dalle, void* ws, size_t ws_bytes, int B, int D,
                    int l2dist, float l2w, float l1w, float tcnw, r3m_stream_t stream) {
  R3M_REQUIRE(ws_bytes >= r3m_loss_workspace_bytes(B), "loss: workspace too small");
  return launch_tcn_lp_loss(alle, perm, iperm, dalle, static_cast<float*>(ws), B, D, l2dist, l2w, l1w, tcnw, S(stream));
}
int r3m_loss_lang_infonce(const float* scores, const float* mask, float* dscore, void* ws, size_t ws_bytes, int B, float langw,
                          r3m_stream_t stream) {
  R3M_REQUIRE(ws_bytes >= r3m_loss_workspace_bytes(B), "loss: workspace too small");
  return launch_lang_infonce(scores, mask, dscore, static_cast<float*>(ws), B, langw, S(stream));
}
int r3m_loss_finalize(void* ws, size_t ws_bytes, int B, int have_lang, float* metrics, float l2w, float l1w, float tcnw, float langw,
                      r3m_stream_t stream) {
  R3M_REQUIRE(ws_bytes >= r3m_loss_workspace_bytes(B), "loss: workspace too small");
  return launch_loss_finalize(static_cast<float*>(ws), B, have_lang, metrics, l2w, l1w, tcnw, langw, S(stream));
}

int r3m_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2, double eps,
                  long long step, float grad_scale, r3m_stream_t stream) {
  return launch_adam(p, g, m, v, n, lr, b1, b2, eps, step, grad_scale, S(stream));
}

int r3m_sgd_step(float* p, const float* g, float* momentum_buf, long long n, double lr, double momentum, double dampening,
                 double weight_decay, int nesterov, long long step, float grad_scale, r3m_stream_t stream) {
  R3M_REQUIRE(p && g, "sgd_step: null argument");
  return launch_sgd(p, g, momentum_buf, n, lr, momentum, dampening, weight_decay, nesterov, step, grad_scale, S(stream));
}

}  // extern "C"
