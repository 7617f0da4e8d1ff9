// Microbenchmark: L2 -> LDS bandwidth of global_load_lds_dwordx4 (L1-missing, L2-resident stream) vs waves per CU and
// outstanding DMA instructions per wave. Build: hipcc --offload-arch=gfx950 -O3 -o l2dma l2dma.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int DEPTH>   // DMA instructions (1 KiB each) in flight per wave before the wait
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, long long region_bytes, int iters, float* sink) {
  __shared__ __attribute__((aligned(128))) unsigned char lds[4 * DEPTH * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* dst = lds + wave * DEPTH * 1024;
  // every block walks its own window of the region; consecutive iterations touch fresh lines (no L1 reuse)
  long long off = ((long long)blockIdx.x * 257 * 1024 + wave * 64 * 1024) % region_bytes;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const char* p = src + (off + d * 1024 + lane * 16) % region_bytes;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    off = (off + DEPTH * 1024 * 4) % region_bytes;
  }
  if (sink && lds[threadIdx.x] == 123 && iters < 0) sink[0] = 1.f;
}

template <int DEPTH>
double run(const char* src, long long region, int blocks, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(blocks), dim3(256), 0, 0, src, region, 10, nullptr);
  hipEventRecord(a);
  hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(blocks), dim3(256), 0, 0, src, region, iters, nullptr);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * 4 * DEPTH * 1024.0 * iters;
  return bytes / (ms * 1e-3) / 1e12;
}

int main() {
  const long long cap = 1LL << 30;
  char* buf;
  hipMalloc(&buf, cap);
  hipMemset(buf, 1, cap);
  const long long regions[] = {8LL << 20, 24LL << 20, 128LL << 20, 1LL << 30};   // fits the 8 L2s (32 MB total) ... HBM
  const int blocks_per_cu[] = {1, 2, 4, 8};
  for (long long region : regions)
    for (int bpc : blocks_per_cu) {
      const int blocks = 256 * bpc;
      printf("region %5lld MB  blocks/CU %d (waves/CU %2d):  depth2 %6.2f  depth4 %6.2f  depth8 %6.2f  depth16 %6.2f TB/s\n", region >> 20, bpc,
             bpc * 4, run<2>(buf, region, blocks, 2000), run<4>(buf, region, blocks, 1000), run<8>(buf, region, blocks, 500),
             run<16>(buf, region, blocks, 250));
    }
  return 0;
}
