// Micro-benchmark (MI355X): sustained rate of v_mfma_f32_16x16x4_f32 -- registers only, and with one ds_read_b32 (the A operand) per
// MFMA or per two MFMAs, at 1, 2 and 4 waves per SIMD.  Ceiling for csrc/direct_conv.hip and csrc/bottleneck_conv.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_rate.hip -o deep-video-mvs_amd/lib/mfma_rate && deep-video-mvs_amd/lib/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float float4v __attribute__((ext_vector_type(4)));

template <int ACC, int LDS_EVERY>   // LDS_EVERY: 0 = no LDS reads; n = one ds_read_b32 per n MFMAs
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  __shared__ float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f / (1 + (i & 7));
  __syncthreads();
  float4v acc[ACC];
#pragma unroll
  for (int a = 0; a < ACC; ++a) acc[a] = float4v{0.f, 0.f, 0.f, 0.f};
  float av = 1.0f + lane, bv = 0.5f;
  const int base = (lane >> 4) * 272 + (lane & 15);     // four channel planes 16 banks apart: conflict-free
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int a = 0; a < ACC; ++a) {
        if (LDS_EVERY && ((r * ACC + a) % LDS_EVERY) == 0) av = lds[base + ((it + r * ACC + a) & 15) * 17 + (r & 3) * 1088];
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < ACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  if (s == 123.456f) out[0] = s;
}

template <int ACC, int LDS_EVERY>
void run(const char* name, int waves_per_simd) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 2000;
  const dim3 grid(256 * (waves_per_simd > 2 ? waves_per_simd / 2 : 1)), block(waves_per_simd >= 2 ? 512 : 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_loop<ACC, LDS_EVERY>), grid, block, 0, 0, out, 10);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<ACC, LDS_EVERY>), grid, block, 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double waves = static_cast<double>(grid.x) * block.x / 64;
  const double mfmas = waves * iters * 8.0 * ACC;
  printf("%-52s %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (%.1f clk per MFMA and SIMD at 2.4 GHz)\n", name, waves_per_simd, best, mfmas * 2048 / best / 1e9,
         best * 1e-3 * 2.4e9 / (mfmas / 1024));
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<5, 0>("5 accumulators, registers only", w);
    run<10, 0>("10 accumulators, registers only", w);
    run<5, 1>("5 accumulators, one ds_read_b32 per MFMA", w);
    run<10, 2>("10 accumulators, one ds_read_b32 per two MFMAs", w);
  }
  return 0;
}
