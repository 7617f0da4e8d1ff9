// Semantics probe for `buffer_load_dwordx4 ... lds` on gfx950 (run on the GPU box): what lands in LDS for lanes whose offset is
// out of range, and whether the SGPR offset takes part in the range check.   hipcc --offload-arch=gfx950 -O2 bufload.hip -o bufload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* __restrict__ src, float* dst, int nbytes, int soff, int oob_lane, unsigned oob_off) {
  __shared__ __attribute__((aligned(16))) float sm[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) sm[i] = -7.f;           // sentinel: does an OOB lane write 0 or leave it?
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if ((int)threadIdx.x == oob_lane) voff = oob_off;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)sm, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) dst[i] = sm[i];
}
int main() {
  const int N = 4096;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<float> r(256);
  auto run = [&](const char* what, int nbytes, int soff, int lane, unsigned off) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, nbytes, soff, lane, off);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    printf("%-60s lane0 %.0f %.0f lane5 %.0f %.0f %.0f %.0f lane40 %.0f lane63 %.0f %.0f\n", what, r[0], r[1], r[20], r[21], r[22], r[23], r[160], r[252], r[255]);
  };
  run("in range, soff 0", N * 4, 0, -1, 0);
  run("in range, soff 1024 (expect +256)", N * 4, 1024, -1, 0);
  run("lane 5 voffset 0x7FFFF000 (expect 0 or -7 at lane5)", N * 4, 0, 5, 0x7FFFF000u);
  run("num_records 640 B: lanes >= 40 OOB", 640, 0, -1, 0);
  run("num_records 640 B, soff 128: is soffset range-checked? (lane 32+ OOB if yes)", 640, 128, -1, 0);
  run("num_records 650 B (partial 16-B at lane 40)", 650, 0, -1, 0);
  run("num_records 0", 0, 0, -1, 0);
  return 0;
}
