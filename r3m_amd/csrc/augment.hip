// r3m_amd — on-GPU RandomResizedCrop resample for the `rc` / `rctraj` augmentations of the Ego4D loader
// (/root/reference/r3m/utils/data_loaders.py:47-50,81-102: transforms.RandomResizedCrop(224, scale=(0.2,1.0)) applied to
// x/255 and scaled back by 255; `rctraj` uses ONE box for the 5 stacked frames of a clip, `rc` one per frame).
// The crop boxes are inputs (drawn on the host with torchvision's get_params algorithm, r3m_amd/augment.py); this kernel
// is the crop + bilinear resize (align_corners=False, no antialias, as torchvision 0.8.2's tensor path) in one gather
// pass, uint8 or float frames in, float 0..255 out: HBM-bound, one read of the box region + one write.
#include "common.h"

namespace r3m {

template <typename T>
__global__ __launch_bounds__(256) void crop_resize_kernel(const T* __restrict__ in, const int* __restrict__ boxes,
                                                           float* __restrict__ out, long long total, int C, int Hi, int Wi,
                                                           int Ho, int Wo, int frames_per_box) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % Wo);
  long long t = idx / Wo;
  const int y = (int)(t % Ho); t /= Ho;
  const int c = (int)(t % C);
  const long long n = t / C;
  const int* b = boxes + (n / frames_per_box) * 4;   // top, left, height, width
  const int top = b[0], left = b[1], bh = b[2], bw = b[3];
  // ATen upsample_bilinear2d, align_corners=False: src = max(0, (dst + 0.5) * scale - 0.5), scale = in/out
  const float sy = fmaxf(((float)y + 0.5f) * ((float)bh / (float)Ho) - 0.5f, 0.f);
  const float sx = fmaxf(((float)x + 0.5f) * ((float)bw / (float)Wo) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < bh - 1 ? 1 : 0), x1 = x0 + (x0 < bw - 1 ? 1 : 0);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const T* p = in + (n * C + c) * (long long)Hi * Wi;
  const float v00 = (float)p[(long long)(top + y0) * Wi + left + x0] / 255.0f;
  const float v01 = (float)p[(long long)(top + y0) * Wi + left + x1] / 255.0f;
  const float v10 = (float)p[(long long)(top + y1) * Wi + left + x0] / 255.0f;
  const float v11 = (float)p[(long long)(top + y1) * Wi + left + x1] / 255.0f;
  out[idx] = (hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11)) * 255.0f;
}

int launch_crop_resize(const void* in, int in_is_u8, const int* boxes, float* out, long long N, int C, int Hi, int Wi, int Ho,
                       int Wo, int frames_per_box, hipStream_t s) {
  R3M_REQUIRE(frames_per_box >= 1, "crop_resize: frames_per_box=%d", frames_per_box);
  const long long total = N * C * Ho * Wo;
  if (in_is_u8)
    hipLaunchKernelGGL((crop_resize_kernel<unsigned char>), dim3(ceil_div(total, 256)), dim3(256), 0, s,
                       static_cast<const unsigned char*>(in), boxes, out, total, C, Hi, Wi, Ho, Wo, frames_per_box);
  else
    hipLaunchKernelGGL((crop_resize_kernel<float>), dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const float*>(in),
                       boxes, out, total, C, Hi, Wi, Ho, Wo, frames_per_box);
  return check_launch("crop_resize");
}

}  // namespace r3m
