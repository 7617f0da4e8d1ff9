// Read-pattern probe for gfx950 (run on the GPU box): what HBM rate does a CU get when it stages an [rows][K] fp32 activation tile
// by LDS DMA, as a function of HOW the tile is walked?
//   A  "K steps" (the GEMM kernels): 256 rows x 128 B per step, 8 steps per 1 KiB row (K = 256): every row's DRAM page is visited
//      eight times, ~2 us apart;
//   B  "K steps of 64": 256 rows x 256 B per step, 4 steps;
//   C  "whole rows": 32 rows x 1 KiB per step (contiguous 32 KiB), 8 steps per 256-row tile — same bytes per step and per tile.
// One step = 8 waves x 4 (A: 5) DMA instructions of 1 KiB; a full wait + barrier per step (the GEMM kernels' ring discipline: the
// next step's loads go out, then the previous step's are waited for); `spin` dependent FMAs per step stand in for the MFMAs.
//   hipcc --offload-arch=gfx950 -O2 readpat.hip -o readpat;  ./readpat [rows_millions] [spin]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void dma16(const void* base, int bytes, void* lds, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const float* A, float* out, int tiles, int spin) {
  extern __shared__ __attribute__((aligned(128))) float smem[];   // 2 stages x 32 KiB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int K = 256, KB = K * 4;            // floats / bytes per row
  float acc = 0.f;
  int stage = 0;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const float* base = A + (long long)t * 256 * K;   // tile: 256 rows x 1 KiB, contiguous 256 KiB
    const int bytes = 256 * KB;
    constexpr int STEPS = MODE == 1 ? 4 : 8;
    for (int s = 0; s < STEPS; ++s) {
      float* lds = smem + stage * 8192;
      if (MODE == 0) {            // 256 rows x 128 B: wave w stages rows 32 w .. +31, 4 instructions of 8 rows
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = wave * 32 + j * 8 + (lane >> 3);
          dma16(base, bytes, lds + (wave * 32 + j * 8) * 32, (unsigned)(r * KB + s * 128 + (lane & 7) * 16));
        }
      } else if (MODE == 1) {     // 256 rows x 256 B (two stages' worth of LDS per step: 64 KiB, still 2 stages -> 128 KiB): 8 instructions of 4 rows
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = wave * 32 + j * 4 + (lane >> 4);
          dma16(base, bytes, smem + stage * 16384 + (wave * 32 + j * 4) * 64, (unsigned)(r * KB + s * 256 + (lane & 15) * 16));
        }
      } else {                    // 32 rows x 1 KiB: wave w stages rows 32 s + 4 w .. +3, 4 instructions of one row each
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = s * 32 + wave * 4 + j;
          dma16(base, bytes, lds + (wave * 4 + j) * 256, (unsigned)(r * KB + lane * 16));
        }
      }
      float a = acc + smem[(stage ^ 1) * (MODE == 1 ? 16384 : 8192) + threadIdx.x];   // touch the stage that landed a step ago
      for (int i = 0; i < spin; ++i) a = __builtin_fmaf(a, 1.0000001f, 1e-9f);
      acc = a;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      stage ^= 1;
    }
  }
  if (acc == 123.456f) out[threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const long long rows = (argc > 1 ? atoll(argv[1]) : 4) * 1000000LL / 256 * 256;
  const int spin = argc > 2 ? atoi(argv[2]) : 0;
  const int tiles = (int)(rows / 256);
  float *A, *out;
  hipMalloc(&A, rows * 1024);
  hipMalloc(&out, 4096);
  hipMemset(A, 0, rows * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[3] = {"A  256 rows x 128 B per step (8 passes over a row)", "B  256 rows x 256 B per step (4 passes)", "C  32 whole rows per step (1 pass)"};
  for (int mode = 0; mode < 3; ++mode) {
    const int lds = mode == 1 ? 131072 : 65536;
    auto fn = mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(fn, dim3(256), dim3(512), lds, 0, A, out, tiles, spin);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%-56s spin %4d: %.3f ms  %.2f TB/s\n", names[mode], spin, ms, rows * 1024.0 / ms * 1e-9);
    }
  }
  return 0;
}
