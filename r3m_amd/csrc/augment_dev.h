// r3m_amd — the bilinear sample of the rc / rctraj crop, shared by the stand-alone crop kernel (augment.hip) and the stem
// pre-passes that read the raw clips directly (conv.hip stem_prep_crop_kernel, stem_bf16.hip stem_prep16_crop_kernel), so the
// fused path performs the SAME float operations in the same order as crop -> stem_prep and produces the same bits.
#pragma once
#include "common.h"

namespace r3m {

// Source frames of an encoder forward that crops on the fly: frames [N,3,Hi,Wi] uint8 or float (0..255) NCHW, one box
// {top, left, height, width} per `frames_per_box` consecutive frames (5 = rctraj: one box per clip; 1 = rc).
struct FrameSource {
  const void* frames;
  int is_u8;
  const int* boxes;
  int frames_per_box;
  int Hi, Wi;
};

// value (0..255 float) of output pixel (y, x) of the Ho x Wo window at (dst_top, dst_left) of the box region resized to
// full_Ho x full_Wo: ATen upsample_bilinear2d, align_corners=False — src = max(0, (dst + 0.5) * scale - 0.5), scale = in/out —
// applied to p/255 and scaled back by 255 as the reference's loader does (/root/reference/r3m/utils/data_loaders.py:88-102).
template <typename T>
__device__ __forceinline__ float bilinear_sample(const T* __restrict__ plane, int Wi, int top, int left, int bh, int bw, int y, int x,
                                                 int dst_top, int dst_left, int full_Ho, int full_Wo) {
  // Every multiply-add below is written out (fmaf or separate operations) and implicit contraction is off, so (1) every caller
  // produces the SAME bits whatever code surrounds the inlined body (the fused stem pre-pass is tested bit-equal to crop ->
  // stem_prep) and (2) the operation sequence is exactly the one ATen's CPU bilinear kernel executes in its FMA builds
  // (source index = fma(dst + 0.5, scale, -0.5); t = fma(w0, v0, w1 * v1) per axis, x first): bit-identical to
  // F.interpolate(x / 255, mode="bilinear", align_corners=False) * 255 on the CPU (tools/experiments: 0 mismatches).
#pragma clang fp contract(off)
  // (rounding the product first would move the interpolation weight by up to 1.5e-5, i.e. the pixel by up to 4e-3 of 255)
  const float sy = fmaxf(fmaf((float)(y + dst_top) + 0.5f, (float)bh / (float)full_Ho, -0.5f), 0.f);
  const float sx = fmaxf(fmaf((float)(x + dst_left) + 0.5f, (float)bw / (float)full_Wo, -0.5f), 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < bh - 1 ? 1 : 0), x1 = x0 + (x0 < bw - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float v00 = (float)plane[(long long)(top + y0) * Wi + left + x0] / 255.0f;
  const float v01 = (float)plane[(long long)(top + y0) * Wi + left + x1] / 255.0f;
  const float v10 = (float)plane[(long long)(top + y1) * Wi + left + x0] / 255.0f;
  const float v11 = (float)plane[(long long)(top + y1) * Wi + left + x1] / 255.0f;
  const float t0 = fmaf(hx, v00, lx * v01);
  const float t1 = fmaf(hx, v10, lx * v11);
  return fmaf(hy, t0, ly * t1) * 255.0f;
}

// the reference's x/255 -> Normalize for channel c with IEEE divisions (/root/reference/r3m/models/models_r3m.py:97-98)
__device__ __forceinline__ float stem_normalize(float v, int c) {
#pragma clang fp contract(off)
  const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
  const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
  return (v / 255.0f - mean) / sd;
}

}  // namespace r3m
