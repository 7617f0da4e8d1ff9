/* r3m_hip.h — C ABI of libr3m_hip.so, the MI355X (gfx950) hot path of R3M representation pre-training.
 *
 * The reference (facebookresearch/r3m, mounted at /root/reference) has no FFI of its own: everything below replaces
 * work the reference reaches through PyTorch / torchvision / torch.optim. Each entry point cites the reference call
 * site(s) whose arithmetic it takes over.  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked host;
 *   - no allocation, no ownership transfer, no hidden synchronisation: work is enqueued on `stream` and returns;
 *   - outputs and workspaces are pre-allocated by the caller (`*_workspace_bytes` / `*_arena_bytes` queries);
 *   - return 0 on success, non-zero on failure with a thread-local message in r3m_last_error();
 *   - activations are NHWC fp32, conv weights OHWI fp32 (= torch OIHW tensors with channels_last strides);
 *   - re-entrant; a `r3m_resnet_t` handle must not be used from two threads at once.
 */
#ifndef R3M_HIP_H
#define R3M_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* r3m_stream_t;  /* hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
typedef struct r3m_resnet* r3m_resnet_t;

int r3m_abi_version(void);
const char* r3m_last_error(void);

/* Measurement aid for bench.py (not part of the replaced reference surface): while enabled, every conv GEMM launch is
 * bracketed by HIP events on its own stream. Classes: 0 gather-GEMM 128x128 tile (conv fwd/dgrad, Linear), 1 gather-GEMM
 * 256x64 tile (64-channel layers), 2 wgrad 128x128, 3 wgrad 64x64. collect() sums elapsed ms / launches / algorithmic
 * FLOPs per class since the last collect (arrays of 4) and resets. */
int r3m_debug_occupancy(int* out4);   /* resident blocks/CU predicted for {gemm128x128, gemm128x128 8-wave, gemm256x64, wgrad128} */
/* Diagnostic: `blocks` workgroups of 256 threads that each hold `lds_bytes` of LDS and idle for `milliseconds` on `stream` — a stand-in
 * for another stream's long-running kernel (an RCCL collective overlapped with backward) when measuring how the compute kernels
 * behave with part of the CUs' LDS / wave slots taken (tools/occupy_ab.py). Returns immediately; does nothing else. */
int r3m_debug_occupy(int blocks, int lds_bytes, double milliseconds, r3m_stream_t stream);
/* Diagnostic: 0 = the encoder's persistent-kernel launches assign tiles statically, 1 (default) = per-XCD tile queues. Returns the old value. */
int r3m_debug_set_dynamic_tiles(int on);
/* Diagnostic, PROBE BUILDS ONLY (-DR3M_PROBES; the shipped library does not contain the kernel and returns -1): 1 = eligible bf16
   forward / dgrad launches run the round-5 persistent big-tile experiment csrc/conv_pw16.hip (pointwise + gather forms), 3 = + its
   3x3 window form; 0 = off. Measured not faster inside the step. Returns the old value. */
int r3m_debug_set_pw16(int mode);
/* Diagnostic (same-process A/B, tests): 1 (default) = the bf16 plans' 3x3 / stride-1 launches run the persistent kernel-row kernels of
   csrc/conv_row16.hip; 0 = the per-tile halo kernels of csrc/conv_bf16.hip (rounds 3-5). The two accumulate the taps in different orders:
   stored elements agree to 1 bf16 ulp, not bit for bit. Returns the old value. */
int r3m_debug_set_conv3x3_bf16(int mode);
/* Diagnostic (same-process A/B, tests): 1 (default) = forwards with training = 2 run the fused inference sequence; 0 = they run the
   training = 0 kernel sequence (conv, then a stand-alone BatchNorm + ReLU pass). Returns the old value. */
int r3m_debug_set_fused_inference(int on);
/* Diagnostic, runs without a GPU: which kernel family the gather-GEMM dispatch (csrc/conv.hip gg_route) picks for every launch of one
   convolution forward (dgrad = 0; flags: 1 = BatchNorm statistics) or input gradient (dgrad = 1; flags: 2 accumulate, 4 masked residual
   join, 64 BatchNorm-backward partials, mask_bits = 1: their ReLU mask comes as bits) — nothing is launched. routes[i]: 1 = 3x3 window
   kernel, 11 / 12 / 13 = persistent kernel (pointwise / gather / strided-output form), 20 = 16-wide-K kernel, 21 = gather kernel,
   22 = generic kernel; bf16 launches: 30 = gather kernel, 31 = per-tile 3x3 halo kernel, 32 = persistent
   kernel-row 3x3 kernel (csrc/conv_row16.hip). Returns the number of launches (a stride-2 dgrad has up to four), -1 on error. */
int r3m_debug_conv_route(int N, int H, int W, int Ci, int Co, int k, int stride, int pad, int dgrad, int flags, int mask_bits, int dtype,
                         int* routes, int cap);
void r3m_profile_enable(int on);
/* which kernel classes are bracketed while profiling is on: bit k = class k (0 conv fwd/dgrad 128-wide, 1 64-wide, 2 / 3 the weight
   gradients); default all. Each bracket costs the stream two event records. Returns the old mask. */
unsigned r3m_profile_classes(unsigned mask);
int r3m_profile_collect(double* ms, long long* launches, double* flops);
int r3m_profile_collect_bytes(double* bytes);     /* algorithmic HBM bytes per class (operands + results once) of the launches of the last collect() */
int r3m_profile_dump_to(const char* host_path);   /* also write one CSV row per launch at collect(); NULL/"" stops */

/* ---------------- encoder engine -----------------------------------------------------------------------------
 * Replaces torchvision.models.resnet{18,34,50}(pretrained=False) with fc=Identity as built by R3M.__init__
 * (r3m/models/models_r3m.py:44-52,62-63) and run by R3M.forward (models_r3m.py:84-100: x/255 -> Normalize -> convnet);
 * backward replaces autograd through that graph (r3m/trainer.py:157).                                            */
r3m_resnet_t r3m_resnet_create(int size /*18|34|50*/, int frames);
void r3m_resnet_destroy(r3m_resnet_t h);
int r3m_resnet_out_dim(r3m_resnet_t h);                 /* 512 | 512 | 2048 (models_r3m.py:45,48,51) */
long long r3m_resnet_num_params(r3m_resnet_t h);        /* floats in the flat parameter / gradient buffers */
long long r3m_resnet_num_buffers(r3m_resnet_t h);       /* floats in the flat running-statistics buffer */
long long r3m_resnet_arena_bytes(r3m_resnet_t h);       /* activation + scratch arena for `frames` */
int r3m_resnet_num_tensors(r3m_resnet_t h);
/* i-th tensor in torchvision state-dict order. kind: 0 conv weight (shape O,I,kh,kw; stored OHWI), 1 bn weight,
 * 2 bn bias (offsets into params), 3 running_mean, 4 running_var (offsets into buffers). */
int r3m_resnet_tensor_info(r3m_resnet_t h, int i, char* name, int name_cap, int* kind, long long* offset, int* ndim,
                           int* shape4);
/* backward stage s (0: avgpool+layer4, 1: layer3, 2: layer2, 3: layer1+stem) finishes params [offset, offset+count) */
int r3m_resnet_stage_range(r3m_resnet_t h, int stage, long long* offset, long long* count);
/* x: [frames,3,224,224] fp32 NCHW in 0..255 (models_r3m.py:96: "Input must be [0, 255]"); h_out: [frames, out_dim].
 * training=1: batch statistics + running-stat update (momentum 0.1, eps 1e-5); 0: running statistics, everything a backward needs
 * kept (fine-tuning with frozen statistics); 2: INFERENCE — running statistics and nothing kept: BatchNorm, the residual join and the
 * ReLU are applied where each convolution stores its result (no raw conv outputs, no stand-alone BatchNorm passes, no mask bits), an
 * identity block's sum overwrites its input. What `load_r3m(...).eval()` under torch.no_grad() runs (/root/reference/r3m/__init__.py:
 * 72-75, r3m/example.py:19-33). fp32: bit-identical to training=0; bf16: one rounding per stored tensor instead of two (closer to
 * float64). r3m_resnet_backward after it returns an error. */
int r3m_resnet_forward(r3m_resnet_t h, const float* x, const float* params, float* buffers, void* arena, float* h_out,
                       int training, r3m_stream_t stream);
/* dh: [frames, out_dim]. Runs stages [stage_begin, stage_end) in order. The stages of one backward share state inside the plan
 * (the running output gradient, buffer roles, BatchNorm partials written by one stage's dgrad epilogues for the next): after
 * each forward they MUST be called in the order 0,1,2,3 (in one call or several); stage 0 may restart a backward over the same
 * forward at any time; any other out-of-order stage, or a backward before the first forward, returns non-zero.
 * accumulate=0 overwrites grads, 1 adds to them. */
/* The same forward fed from RAW clips through crop boxes: the rc / rctraj RandomResizedCrop(224) of the reference's loader
 * (r3m/utils/data_loaders.py:47-50,88-102) resampled INSIDE the stem pre-pass — one gather-bilinear pass from uint8 (or float
 * 0..255) frames [F,3,Hi,Wi] straight into the normalised stem image; the cropped fp32 frames are never materialised.
 * boxes[f / frames_per_box] = {top, left, height, width}; frames_per_box = 5 (rctraj: one box per clip) or 1 (rc).
 * Bit-identical to r3m_crop_resize followed by r3m_resnet_forward. */
int r3m_resnet_forward_crop(r3m_resnet_t h, const void* frames, int frames_are_u8, const int* boxes, int frames_per_box, int Hi,
                            int Wi, const float* params, float* buffers, void* arena, float* h_out, int training,
                            r3m_stream_t stream);
/* Per-plan switch of the backward schedule: 1 = the first pass of BatchNorm backward is computed inside the epilogue of the dgrad
 * that produces its dz (EPI_BNRED; default for fp32 plans), 0 = stand-alone reduce passes (default for bf16 plans, where the
 * fused form measured slower). Both schedules compute the same sums (different summation order). Returns the previous value. */
int r3m_resnet_set_fused_bn_reduce(r3m_resnet_t h, int on);
/* Per-plan switch: 1 (default) = the two BatchNorms that feed a downsample block's add + ReLU (the block's last one and the downsample
 * branch's) run their backward passes as ONE launch each — they see the same masked output gradient, which is then read once instead
 * of twice; 0 = separate passes. Bit-identical gradients either way. Returns the previous value. */
int r3m_resnet_set_bn_pair(r3m_resnet_t h, int on);
int r3m_resnet_backward(r3m_resnet_t h, const float* dh, const float* params, float* grads, void* arena, int stage_begin,
                        int stage_end, int accumulate, r3m_stream_t stream);

/* ---------------- single operators (parity-tested one by one) -------------------------------------------------
 * conv2d fwd / dgrad / wgrad: ATen conv2d + autograd under torchvision ResNet.forward (call site models_r3m.py:99). */
int r3m_conv2d_stats_rows(int N, int Hi, int Wi, int Co, int k, int stride, int pad);
/* stats (optional): [stats_rows][2][Co] per-row-block sum / sum of squares of y (BatchNorm statistics partials) */
int r3m_conv2d_fwd(const float* x, const float* w_ohwi, float* y, float* stats, int N, int Hi, int Wi, int Ci, int Co, int k,
                   int stride, int pad, r3m_stream_t stream);
size_t r3m_conv2d_dgrad_workspace_bytes(int Ci, int Co, int k);
int r3m_conv2d_dgrad(const float* dy, const float* w_ohwi, float* dx, void* workspace, size_t workspace_bytes, int N, int Hi,
                     int Wi, int Ci, int Co, int k, int stride, int pad, r3m_stream_t stream);
size_t r3m_conv2d_wgrad_workspace_bytes(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad);
int r3m_conv2d_wgrad(const float* x, const float* dy, float* dw_ohwi, void* workspace, size_t workspace_bytes, int N, int Hi,
                     int Wi, int Ci, int Co, int k, int stride, int pad, int accumulate, r3m_stream_t stream);
/* the stem as the engine runs it (models_r3m.py:97-99 + torchvision conv1), no patch matrix in HBM:
 *   r3m_stem_prep      x [frames,3,224,224] fp32 NCHW in 0..255 -> xn [frames,224,224,3] = (x/255 - mean)/std (NHWC)
 *   r3m_stem_conv_fwd  xn, w_ohwi [64,7,7,3] -> y [frames,112,112,64] (NHWC) (+ BatchNorm partials [frames*49][2][64])
 *   r3m_stem_conv_wgrad  xn, dy (same layout as y) -> dw_ohwi [64,7,7,3] */
int r3m_stem_prep(const float* x_nchw, float* xn, int frames, r3m_stream_t stream);
int r3m_stem_conv_fwd(const float* xn, const float* w_ohwi, float* y, float* stats, int frames, r3m_stream_t stream);
size_t r3m_stem_conv_wgrad_workspace_bytes(void);
int r3m_stem_conv_wgrad(const float* xn, const float* dy, float* dw_ohwi, void* workspace, size_t workspace_bytes, int frames,
                        int accumulate, r3m_stream_t stream);

/* BatchNorm2d(eps, momentum) train/eval + ReLU + residual add (torchvision BasicBlock/Bottleneck).
 * coef: [4][C] = mean, invstd, scale = gamma*invstd, shift = beta - mean*scale. */
size_t r3m_bn_workspace_bytes(long long rows, int C);
int r3m_bn_train_coeffs(const float* stats, int stats_rows, long long count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* coef, void* workspace,
                        size_t workspace_bytes, int C, r3m_stream_t stream);
int r3m_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                       float* coef, int C, r3m_stream_t stream);
/* z = [relu](scale*y + shift [+ r] [+ scale2*y2 + shift2]) ; pass r for the identity branch, or y2+coef2 for a
 * downsample branch (then r must be NULL). maskbits (optional): 1 bit per element = [z > 0], rows*C/8 bytes (float4 index i
 * owns nibble i&7 of 32-bit word i>>3) — what the backward reads instead of z. */
int r3m_bn_act_fwd(const float* y, const float* coef, const float* r, const float* y2, const float* coef2, float* z,
                   long long rows, int C, int relu, unsigned* maskbits, r3m_stream_t stream);
/* g = dz * [mask], mask = zbits (bit mask from r3m_bn_act_fwd) if given, else (zmask > 0) if given, else
 * (scale*y+shift > 0); dgamma = sum g*yhat, dbeta = sum g,
 * dy = scale*(g - mean(g) - yhat*mean(g*yhat)) (batch stats) or scale*g (use_batch_stats=0). */
int r3m_bn_bwd(const float* dz, const float* zmask, const unsigned* zbits, const float* y, const float* coef, float* dgamma,
               float* dbeta, float* dy, void* workspace, size_t workspace_bytes, long long rows, int C, int use_batch_stats, int accumulate,
               r3m_stream_t stream);
/* MaxPool2d(3,2,1) and AdaptiveAvgPool2d(1)+flatten, NHWC */
int r3m_maxpool_fwd(const float* z, float* p, unsigned char* argmax, int N, int Hi, int Wi, int C, r3m_stream_t stream);
int r3m_maxpool_bwd(const float* dp, const unsigned char* argmax, float* dz, int N, int Hi, int Wi, int C, r3m_stream_t stream);
int r3m_avgpool_fwd(const float* x, float* h, int N, int HW, int C, r3m_stream_t stream);
int r3m_avgpool_bwd(const float* dh, float* dx, int N, int HW, int C, r3m_stream_t stream);

/* ---------------- mixed precision: bf16 activations (BASELINE configs[2], [4]) -------------------------------------
 * The `_dt` variants take `dtype`: R3M_DT_F32 (identical to the plain entry points above) or R3M_DT_BF16. With bf16 every
 * ACTIVATION tensor (x, y, z, r, dy, dz, dx, dp ...) is NHWC bfloat16; what stays fp32: master weights, weight / BatchNorm
 * gradients, BatchNorm statistics partials and coefficients, the embedding h / dh, and all
 * accumulation (v_mfma_f32_32x32x16_bf16). This is the counterpart of running the reference's encoder call
 * (r3m/models/models_r3m.py:99, backward at r3m/trainer.py:157) under torch.autocast(bfloat16); the reference itself is
 * fp32 only. Channel counts must be multiples of 64 (every ResNet-18/34/50 layer behind the stem is).
 *   r3m_conv2d_fwd_dt    w: bf16 [Co][k][k][Ci] for R3M_DT_BF16 (make it with r3m_convert_bf16 from the fp32 master)
 *   r3m_conv2d_dgrad_dt  w: the fp32 master [Co][k][k][Ci] (transposed + converted into the workspace)
 *   r3m_conv2d_wgrad_dt  dw: fp32 */
enum { R3M_DT_F32 = 0, R3M_DT_BF16 = 1 };
r3m_resnet_t r3m_resnet_create_dt(int size /*18|34|50*/, int frames, int dtype);
int r3m_resnet_dtype(r3m_resnet_t h);
int r3m_convert_bf16(const float* src, void* dst_bf16, long long n /* multiple of 4 */, r3m_stream_t stream);
int r3m_conv2d_fwd_dt(const void* x, const void* w_ohwi, void* y, float* stats, int N, int Hi, int Wi, int Ci, int Co, int k,
                      int stride, int pad, int dtype, r3m_stream_t stream);
int r3m_conv2d_dgrad_dt(const void* dy, const float* w_ohwi, void* dx, void* workspace, size_t workspace_bytes, int N, int Hi,
                        int Wi, int Ci, int Co, int k, int stride, int pad, int dtype, r3m_stream_t stream);
/* dgrad whose epilogue ALSO emits the first pass of the consumer BatchNorm's backward (what the engine runs for every BatchNorm
 * whose dz has one producing dgrad): dx [N,Hi,Wi,Ci] is the gradient entering BatchNorm(+ReLU) with pre-normalisation input bn_y
 * (same shape / dtype as dx) — optionally plus a masked residual gradient (dx = dgrad + residual_grad * [residual_bits], the
 * join at a block input). partials [r3m_conv2d_dgrad_bnred_rows][2][Ci] (fp32): per 64 result rows, sum(g) and
 * sum(g * (y - mean)) with g = dx * [mask]; mask = bn_bits (1 bit per element of the BatchNorm's block output) or, when
 * bn_bits is NULL, recomputed as fma(y, bn_scale, bn_shift) > 0. Equals r3m_bn_bwd's reduce pass on the stored dx
 * (replaces torch's batch_norm_backward reduction under the ResNet graph, call site trainer.py:157). */
int r3m_conv2d_dgrad_bnred_rows(int N, int Hi, int Wi, int stride);
int r3m_conv2d_dgrad_bnred_dt(const void* dy, const float* w_ohwi, void* dx, void* workspace, size_t workspace_bytes, int N, int Hi,
                              int Wi, int Ci, int Co, int k, int stride, int pad, const void* residual_grad,
                              const unsigned* residual_bits, const void* bn_y, const unsigned* bn_bits, const float* bn_scale,
                              const float* bn_shift, const float* bn_mean, float* partials, int dtype, r3m_stream_t stream);
size_t r3m_conv2d_wgrad_workspace_bytes_dt(int N, int Hi, int Wi, int Ci, int Co, int k, int stride, int pad, int dtype);
int r3m_conv2d_wgrad_dt(const void* x, const void* dy, float* dw_ohwi, void* workspace, size_t workspace_bytes, int N, int Hi,
                        int Wi, int Ci, int Co, int k, int stride, int pad, int accumulate, int dtype, r3m_stream_t stream);
int r3m_stem_conv_fwd_dt(const float* xn, const float* w_ohwi, void* y, float* stats, int frames, int dtype, r3m_stream_t stream);
int r3m_stem_conv_wgrad_dt(const float* xn, const void* dy, float* dw_ohwi, void* workspace, size_t workspace_bytes, int frames,
                           int accumulate, int dtype, r3m_stream_t stream);
/* The stem on the bf16 MFMA (bf16 plans): r3m_stem_prep_bf16 writes the normalised frames as a padded, channel-interleaved bf16
 * image xn16[frames][232][704] (r3m_stem_xn16_bytes), from which conv1 forward (y: bf16 [frames,112,112,64], stats as above) and its
 * weight gradient (dy bf16, dw fp32 [64,7,7,3]) stage their operands by plain contiguous copies. Same call sites as r3m_stem_*. */
size_t r3m_stem_xn16_bytes(int frames);
int r3m_stem_prep_bf16(const float* x_nchw, void* xn16, int frames, r3m_stream_t stream);
/* The same pre-pass reading RAW clips through crop boxes (what r3m_resnet_forward_crop runs first): frames [F,3,Hi,Wi] uint8 or
 * float 0..255, boxes[f / frames_per_box] = {top, left, height, width}; dtype R3M_DT_F32 writes xn (layout of r3m_stem_prep),
 * R3M_DT_BF16 writes xn16 (layout of r3m_stem_prep_bf16). Bit-identical to r3m_crop_resize followed by r3m_stem_prep[_bf16]. */
int r3m_stem_prep_crop(const void* frames, int frames_are_u8, const int* boxes, int frames_per_box, int Hi, int Wi, void* xn_out,
                       int F, int dtype, r3m_stream_t stream);
int r3m_stem_conv_fwd_bf16(const void* xn16, const float* w_ohwi, void* y, float* stats, int frames, r3m_stream_t stream);
size_t r3m_stem_conv_wgrad_bf16_workspace_bytes(void);
int r3m_stem_conv_wgrad_bf16(const void* xn16, const void* dy, float* dw_ohwi, void* workspace, size_t workspace_bytes, int frames,
                             int accumulate, r3m_stream_t stream);
int r3m_bn_act_fwd_dt(const void* y, const float* coef, const void* r, const void* y2, const float* coef2, void* z,
                      long long rows, int C, int relu, unsigned* maskbits, int dtype, r3m_stream_t stream);
int r3m_bn_bwd_dt(const void* dz, const void* zmask, const unsigned* zbits, const void* y, const float* coef, float* dgamma,
                  float* dbeta, void* dy, void* workspace, size_t workspace_bytes, long long rows, int C, int use_batch_stats,
                  int accumulate, int dtype, r3m_stream_t stream);
/* Stem tail fused (what the engine runs after conv1): z = relu(bn(y)) is pooled on the fly — the activated tensor and its
 * gradient, the two largest tensors of the network, are never written. Same arithmetic as r3m_bn_act_fwd + r3m_maxpool_fwd and
 * r3m_maxpool_bwd + r3m_bn_bwd (mask recomputed from y); workspace = r3m_bn_workspace_bytes(N*Hi*Wi, C). */
int r3m_bn_relu_maxpool_fwd_dt(const void* y, const float* coef, void* p, unsigned char* argmax, int N, int Hi, int Wi, int C,
                               int dtype, r3m_stream_t stream);
int r3m_bn_maxpool_bwd_dt(const void* dp, const unsigned char* argmax, const void* y, const float* coef, float* dgamma, float* dbeta,
                          void* dy, void* workspace, size_t workspace_bytes, int N, int Hi, int Wi, int C, int use_batch_stats,
                          int accumulate, int dtype, r3m_stream_t stream);
int r3m_maxpool_fwd_dt(const void* z, void* p, unsigned char* argmax, int N, int Hi, int Wi, int C, int dtype, r3m_stream_t stream);
int r3m_maxpool_bwd_dt(const void* dp, const unsigned char* argmax, void* dz, int N, int Hi, int Wi, int C, int dtype,
                       r3m_stream_t stream);
int r3m_avgpool_fwd_dt(const void* x, float* h, int N, int HW, int C, int dtype, r3m_stream_t stream);
int r3m_avgpool_bwd_dt(const float* dh, void* dx, int N, int HW, int C, int dtype, r3m_stream_t stream);

/* nn.Linear (+ReLU) of LanguageReward.pred (r3m/models/models_language.py:43-51): y[M,N] = x[M,K] w[N,K]^T + b */
int r3m_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int relu,
                   r3m_stream_t stream);

/* RandomResizedCrop resample of the rc / rctraj augmentations (r3m/utils/data_loaders.py:47-50,81-102): crop box
 * boxes[n / frames_per_box] = {top, left, height, width} (host-drawn), bilinear resize to Ho x Wo (align_corners=False),
 * on x/255, result * 255. frames: [N,C,Hi,Wi] uint8 or float in 0..255; out: [N,C,Ho,Wo] float. frames_per_box = 5 for
 * rctraj (one box per clip), 1 for rc. */
int r3m_crop_resize(const void* frames, int frames_are_u8, const int* boxes, float* out, long long N, int C, int Hi, int Wi,
                    int Ho, int Wo, int frames_per_box, r3m_stream_t stream);
/* R3M.forward's branch for inputs that are not 224 x 224 (r3m/models/models_r3m.py:85-90: transforms.Resize(256) +
 * CenterCrop(224) on x/255): bilinear resize of the whole frame to resized_h x resized_w (align_corners=False, no antialias) of
 * which only the Ho x Wo window at (top, left) is computed. frames [N,C,Hi,Wi] uint8 or float 0..255 -> out [N,C,Ho,Wo] float
 * 0..255. The caller derives resized_h/w and the window with torchvision's rounding (r3m_amd/augment.py). */
int r3m_resize_crop(const void* frames, int frames_are_u8, float* out, long long N, int C, int Hi, int Wi, int resized_h,
                    int resized_w, int top, int left, int Ho, int Wo, r3m_stream_t stream);

/* LanguageReward, all 15 evaluations of a step batched (r3m/trainer.py:72-92 calling r3m/models/models_r3m.py:78-81 and
 * r3m/models/models_language.py:43-55). alle [B,5,D]; feats [B,lang_dim] = frozen sentence features (LangEncoder output,
 * NOT permuted); perm/iperm [9][B] int32 = the reference's torch.randperm draws in order (a,b,c) x 3 (trainer.py:86-92) and
 * their inverses; params/grads = flat pred.{0,2,4,6,8}.{weight,bias} in state-dict order (r3m_langrew_num_params floats).
 * forward leaves the activations in `workspace`; backward consumes them, writes parameter gradients (= or +=) and ADDS
 * d/d alle into dalle [B,5,D] (may be NULL). scores/dscore are [15][B] in the reference's call order. */
long long r3m_langrew_num_params(int D, int hidden, int lang_dim);
size_t r3m_langrew_workspace_bytes(int B, int D, int hidden, int lang_dim);
int r3m_langrew_forward(const float* alle, const float* feats, const int* perm, const float* params, float* scores, void* workspace,
                        size_t workspace_bytes, int B, int D, int hidden, int lang_dim, r3m_stream_t stream);
int r3m_langrew_backward(const float* dscore, const int* iperm, const float* params, float* grads, float* dalle, void* workspace,
                         size_t workspace_bytes, int B, int D, int hidden, int lang_dim, int accumulate, r3m_stream_t stream);
/* The same pass with the MLP's tensors stored in `dtype` (R3M_DT_F32: identical to the calls above; R3M_DT_BF16: mixed precision
 * in the manner of torch.autocast(bfloat16) around the reference's get_reward calls (r3m/trainer.py:72-92) — input rows, hidden
 * activations and their gradients bf16, every Linear on the bf16 GEMM kernels with fp32 accumulation; master weights, biases,
 * scores, all parameter gradients and dalle stay fp32 — with one difference: a hidden activation is rounded twice (GEMM result
 * to bf16, then bias + ReLU to bf16) where autocast rounds once). Same workspace size; bf16 needs lang_dim % 64 == 0. */
int r3m_langrew_forward_dt(const float* alle, const float* feats, const int* perm, const float* params, float* scores, void* workspace,
                           size_t workspace_bytes, int B, int D, int hidden, int lang_dim, int dtype, r3m_stream_t stream);
int r3m_langrew_backward_dt(const float* dscore, const int* iperm, const float* params, float* grads, float* dalle, void* workspace,
                            size_t workspace_bytes, int B, int D, int hidden, int lang_dim, int accumulate, int dtype, r3m_stream_t stream);
/* ONE differentiable evaluation score[R] = G(e0[R,D], eg[R,D], le[R,lang_dim]) — the reference's own calling form
 * (R3M.get_reward r3m/models/models_r3m.py:78-81 -> LanguageReward.forward models_language.py:53-55, called 15x with autograd
 * by r3m/trainer.py:72-92). forward leaves the activations in `workspace` (r3m_langrew_call_workspace_bytes(R, ...));
 * backward consumes them, writes parameter gradients (= or +=, flat layout as above) and WRITES the input gradients
 * de0 / deg [R,D] and dle [R,lang_dim] (each may be NULL). */
size_t r3m_langrew_call_workspace_bytes(int R, int D, int hidden, int lang_dim);
int r3m_langrew_call_forward(const float* e0, const float* eg, const float* le, const float* params, float* score, void* workspace,
                             size_t workspace_bytes, int R, int D, int hidden, int lang_dim, r3m_stream_t stream);
int r3m_langrew_call_backward(const float* dscore, const float* params, float* grads, float* de0, float* deg, float* dle,
                              void* workspace, size_t workspace_bytes, int R, int D, int hidden, int lang_dim, int accumulate,
                              r3m_stream_t stream);

/* ---------------- objective (r3m/trainer.py:39-152, R3M.sim models_r3m.py:102-107) ----------------------------
 * alle [B,5,D] (e0, eg, es0, es1, es2 per clip); perm/iperm [6][B] int32: the reference's torch.randperm draws in
 * order (es0-perm, es2-perm) x 3 (trainer.py:136-137) and their inverses. Writes d(full_loss)/d(alle) to dalle
 * (may be NULL: metrics only). workspace: r3m_loss_workspace_bytes(B).  */
size_t r3m_loss_workspace_bytes(int B);
int r3m_loss_tcn_lp(const float* alle, const int* perm, const int* iperm, float* dalle, void* workspace, size_t workspace_bytes,
                    int B, int D, int l2dist, float l2weight, float l1weight, float tcnweight, r3m_stream_t stream);
/* scores [15][B] of the 15 get_reward calls in reference order (trainer.py:72-92); mask [B] (trainer.py:107-109);
 * dscore [15][B] = d(full_loss)/d(score). Same workspace as r3m_loss_tcn_lp. */
int r3m_loss_lang_infonce(const float* scores, const float* mask, float* dscore, void* workspace, size_t workspace_bytes, int B,
                          float langweight, r3m_stream_t stream);
/* metrics[16]: 0 l2loss 1 l1loss 2 l0loss 3 tcnloss 4 aligned 5 rewloss 6-8 rewacc1-3 9 full_loss (trainer.py:55-57,111-117,147-152) */
int r3m_loss_finalize(void* workspace, size_t workspace_bytes, int B, int have_lang, float* metrics, float l2weight,
                      float l1weight, float tcnweight, float langweight, r3m_stream_t stream);

/* ---------------- optimizer: torch.optim.Adam(params, lr) defaults (models_r3m.py:76; trainer.py:156-158) -------
 * one pass over the flat buffers; step counts from 1; grad_scale multiplies g on the fly (1/world_size for SUM all-reduce) */
int r3m_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, double lr, double beta1,
                  double beta2, double eps, long long step, float grad_scale, r3m_stream_t stream);
/* torch.optim.SGD on the same flat buffers (momentum / dampening / weight_decay / nesterov as torch defines them; momentum_buf may
 * be NULL when momentum == 0; step counts from 1 and selects the buffer initialisation). The reference trains with Adam only
 * (models_r3m.py:76); BASELINE.json's north_star names "the SGD/Adam step". */
int r3m_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, double lr, double momentum, double dampening,
                 double weight_decay, int nesterov, long long step, float grad_scale, r3m_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
