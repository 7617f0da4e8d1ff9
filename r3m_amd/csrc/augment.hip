// r3m_amd — on-GPU RandomResizedCrop resample for the `rc` / `rctraj` augmentations of the Ego4D loader
// (/root/reference/r3m/utils/data_loaders.py:47-50,81-102: transforms.RandomResizedCrop(224, scale=(0.2,1.0)) applied to
// x/255 and scaled back by 255; `rctraj` uses ONE box for the 5 stacked frames of a clip, `rc` one per frame).
// The crop boxes are inputs (drawn on the host with torchvision's get_params algorithm, r3m_amd/augment.py); this kernel
// is the crop + bilinear resize (align_corners=False, no antialias, as torchvision 0.8.2's tensor path) in one gather
// pass, uint8 or float frames in, float 0..255 out: HBM-bound, one read of the box region + one write.
#include "common.h"
#include "augment_dev.h"

namespace r3m {

template <typename T>
__global__ __launch_bounds__(256) void crop_resize_kernel(const T* __restrict__ in, const int* __restrict__ boxes,
                                                           float* __restrict__ out, long long total, int C, int Hi, int Wi,
                                                           int Ho, int Wo, int frames_per_box, int dst_top, int dst_left,
                                                           int full_Ho, int full_Wo) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % Wo);
  long long t = idx / Wo;
  const int y = (int)(t % Ho); t /= Ho;
  const int c = (int)(t % C);
  const long long n = t / C;
  // box = source region (top, left, height, width); without boxes: the whole frame
  const int* b = boxes ? boxes + (n / frames_per_box) * 4 : nullptr;
  const int top = b ? b[0] : 0, left = b ? b[1] : 0, bh = b ? b[2] : Hi, bw = b ? b[3] : Wi;
  // The region is resized to full_Ho x full_Wo, of which this launch writes the Ho x Wo window at (dst_top, dst_left):
  // crop-then-resize has window == everything, resize-then-centre-crop has region == everything.
  const T* p = in + (n * C + c) * (long long)Hi * Wi;
  out[idx] = bilinear_sample(p, Wi, top, left, bh, bw, y, x, dst_top, dst_left, full_Ho, full_Wo);
}

static int launch_resample(const void* in, int in_is_u8, const int* boxes, float* out, long long N, int C, int Hi, int Wi, int Ho,
                           int Wo, int frames_per_box, int dst_top, int dst_left, int full_Ho, int full_Wo, hipStream_t s) {
  const long long total = N * C * Ho * Wo;
  if (total == 0) return 0;
  if (in_is_u8)
    hipLaunchKernelGGL((crop_resize_kernel<unsigned char>), dim3(ceil_div(total, 256)), dim3(256), 0, s,
                       static_cast<const unsigned char*>(in), boxes, out, total, C, Hi, Wi, Ho, Wo, frames_per_box, dst_top, dst_left,
                       full_Ho, full_Wo);
  else
    hipLaunchKernelGGL((crop_resize_kernel<float>), dim3(ceil_div(total, 256)), dim3(256), 0, s, static_cast<const float*>(in),
                       boxes, out, total, C, Hi, Wi, Ho, Wo, frames_per_box, dst_top, dst_left, full_Ho, full_Wo);
  return check_launch("crop_resize");
}

int launch_crop_resize(const void* in, int in_is_u8, const int* boxes, float* out, long long N, int C, int Hi, int Wi, int Ho,
                       int Wo, int frames_per_box, hipStream_t s) {
  R3M_REQUIRE(frames_per_box >= 1, "crop_resize: frames_per_box=%d", frames_per_box);
  return launch_resample(in, in_is_u8, boxes, out, N, C, Hi, Wi, Ho, Wo, frames_per_box, 0, 0, Ho, Wo, s);
}

// Resize(full) + crop of the Ho x Wo window at (top, left) in ONE pass: R3M.forward's Resize(256) + CenterCrop(224) branch for
// inputs that are not 224 x 224 (/root/reference/r3m/models/models_r3m.py:85-90). Only the window is ever computed.
int launch_resize_crop(const void* in, int in_is_u8, float* out, long long N, int C, int Hi, int Wi, int full_Ho, int full_Wo,
                       int top, int left, int Ho, int Wo, hipStream_t s) {
  R3M_REQUIRE(full_Ho >= 1 && full_Wo >= 1 && top >= 0 && left >= 0 && top + Ho <= full_Ho && left + Wo <= full_Wo,
              "resize_crop: window %dx%d at (%d,%d) outside the %dx%d resized frame", Ho, Wo, top, left, full_Ho, full_Wo);
  return launch_resample(in, in_is_u8, nullptr, out, N, C, Hi, Wi, Ho, Wo, 1, top, left, full_Ho, full_Wo, s);
}

}  // namespace r3m
