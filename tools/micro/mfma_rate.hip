// micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 per SIMD under different accumulator counts / wave mixes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int NWAVE_MFMA>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
  const int wave = threadIdx.x >> 6;
  if (wave >= NWAVE_MFMA) { __syncthreads(); return; }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  __syncthreads();
}
template <int NACC, int NW>
void run(const char* name, int blocks, int threads, int total_mfma_per_wave) {
  float* out; hipMalloc(&out, blocks * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = total_mfma_per_wave / NACC;
  hipLaunchKernelGGL((k<NACC, NW>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NACC, NW>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double mf = (double)blocks * NW * iters * NACC;
  printf("%-34s blocks %4d thr %3d: %8.1f us  -> %6.1f TFLOP/s, %6.1f ns per MFMA per wave\n", name, blocks, threads, ms * 1e3,
         mf * 4096 / ms / 1e9, ms * 1e6 / (iters * NACC));
  hipFree(out);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d clock %d kHz\n", p.multiProcessorCount, p.clockRate);
  run<1, 4>("1 acc, 4 waves/block", 256, 256, 2160);
  run<2, 4>("2 acc, 4 waves/block", 256, 256, 2160);
  run<4, 4>("4 acc, 4 waves/block", 256, 256, 2160);
  run<1, 4>("1 acc, 4 mfma + 4 idle waves", 256, 512, 2160);
  run<2, 4>("2 acc, 4 mfma + 4 idle waves", 256, 512, 2160);
  run<1, 4>("1 acc, 4 waves, 512 blocks", 512, 256, 2160);
  run<1, 4>("1 acc, 4 waves, 1024 blocks", 1024, 256, 2160);
  run<4, 4>("4 acc, 4 waves, 1024 blocks", 1024, 256, 2160);
  run<1, 4>("1 acc, 4 waves, 256 blk long", 256, 256, 21600);
  run<4, 4>("4 acc, 4 waves, 256 blk long", 256, 256, 21600);
  run<1, 8>("1 acc, 8 mfma waves", 256, 512, 2160);
  return 0;
}
