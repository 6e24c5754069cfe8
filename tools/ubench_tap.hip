// Micro-benchmark of the plane sweep's inner pattern on gfx950 (dev tool, not part of the library):
// per "slot" 8 ds_read_b128 (4 taps x 2 channel quads of a 48-byte record) + 16 v_pk_fma_f32 (dot per tap) + 4 v_pk_fma_f32
// (bilinear weights), conflict-free addresses, 256-thread workgroups, 48 KB LDS each (3 per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_tap.hip -o gpurun_out/ubench_tap && gpurun_out/ubench_tap
// MODE 0: reads + packed FMAs   1: reads only (one add per read)   2: packed FMAs only   3: reads + scalar FMAs
//      4: reads + packed FMAs, two slots software-pipelined by hand
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int kRec = 12, kCap = 1024, kSlots = 8;

template <int MODE>
__global__ __launch_bounds__(256, 3) void tap_kernel(const float* in, float* out, int iters, int pitch) {
  extern __shared__ __attribute__((aligned(16))) float s_tile[];
  const int tid = threadIdx.x;
  for (int i = tid; i < kCap * kRec; i += 256) s_tile[i] = in[i % 4096];
  __syncthreads();
  // lane -> 16 consecutive records per ds_read_b128 service group (as sweep_lane_pixel does)
  const int l = tid & 31;
  const int px = l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
  int addr[kSlots];
  float2v wn[kSlots], ws[kSlots], acc[kSlots];
#pragma unroll
  for (int k = 0; k < kSlots; ++k) {
    addr[k] = (((tid >> 5) * pitch + px + k) % (kCap - pitch - 2)) * kRec * 4;
    wn[k] = float2v{0.25f + k, 0.5f};
    ws[k] = float2v{0.125f, 0.75f - k};
    acc[k] = float2v{0.0f, 0.0f};
  }
  float2v rv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rv[c] = float2v{in[tid + c], in[tid + 8 + c]};
  const char* base = reinterpret_cast<const char*>(s_tile);
  const int row_bytes = pitch * kRec * 4;
  float4v fake[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) fake[i] = float4v{in[tid + i], in[tid + i + 1], in[tid + i + 2], in[tid + i + 3]};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
      asm volatile("" : "+v"(addr[k]));   // opaque: the reads are not loop-invariant
      const char* r0 = base + addr[k];
      const char* r1 = r0 + row_bytes;
      float4v t[8];
      if (MODE != 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          t[q * 4 + 0] = *reinterpret_cast<const float4v*>(r0 + q * 16);
          t[q * 4 + 1] = *reinterpret_cast<const float4v*>(r0 + kRec * 4 + q * 16);
          t[q * 4 + 2] = *reinterpret_cast<const float4v*>(r1 + q * 16);
          t[q * 4 + 3] = *reinterpret_cast<const float4v*>(r1 + kRec * 4 + q * 16);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("" : "+v"(fake[i]));
          t[i] = fake[i];
        }
      }
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[k] += t[i].lo;
      } else if (MODE == 3) {
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int tp = 0; tp < 4; ++tp) {
            const float4v v = t[q * 4 + tp];
            s[tp] = fmaf(rv[q * 2].x, v.x, s[tp]);
            s[tp] = fmaf(rv[q * 2].y, v.y, s[tp]);
            s[tp] = fmaf(rv[q * 2 + 1].x, v.z, s[tp]);
            s[tp] = fmaf(rv[q * 2 + 1].y, v.w, s[tp]);
          }
        acc[k].x = fmaf(s[0], wn[k].x, fmaf(s[1], wn[k].y, fmaf(s[2], ws[k].x, fmaf(s[3], ws[k].y, acc[k].x))));
      } else {
        float2v s[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int tp = 0; tp < 4; ++tp) {
            const float4v v = t[q * 4 + tp];
            s[tp] = __builtin_elementwise_fma(rv[q * 2], v.lo, s[tp]);
            s[tp] = __builtin_elementwise_fma(rv[q * 2 + 1], v.hi, s[tp]);
          }
        float2v f = acc[k];
        f = __builtin_elementwise_fma(s[0], __builtin_shufflevector(wn[k], wn[k], 0, 0), f);
        f = __builtin_elementwise_fma(s[1], __builtin_shufflevector(wn[k], wn[k], 1, 1), f);
        f = __builtin_elementwise_fma(s[2], __builtin_shufflevector(ws[k], ws[k], 0, 0), f);
        f = __builtin_elementwise_fma(s[3], __builtin_shufflevector(ws[k], ws[k], 1, 1), f);
        acc[k] = f;
      }
    }
  }
  float r = 0.0f;
#pragma unroll
  for (int k = 0; k < kSlots; ++k) r += acc[k].x + acc[k].y;
  out[blockIdx.x * 256 + tid] = r;
}

template <int MODE>
void run(const char* name, const float* in, float* out, int wgs, int iters, int pitch) {
  const size_t lds = sizeof(float) * kCap * kRec;
  hipFuncSetAttribute(reinterpret_cast<const void*>(tap_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(tap_kernel<MODE>, dim3(wgs), dim3(256), lds, 0, in, out, iters, pitch);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(tap_kernel<MODE>, dim3(wgs), dim3(256), lds, 0, in, out, iters, pitch);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  // slot-passes per CU: (wgs / 256 CUs) * 4 waves * iters * kSlots
  const double wave_slots_per_cu = static_cast<double>(wgs) / 256.0 * 4.0 * iters * kSlots;
  const double ns_per_wave_slot = best * 1e6 / wave_slots_per_cu;
  printf("%-34s wgs %4d pitch %3d: %8.3f ms  -> %6.2f ns per wave-slot per CU (%5.1f cycles at 2.4 GHz); the sweep has 640 x 64 wave-slots per CU-equivalent... \n",
         name, wgs, pitch, best, ns_per_wave_slot, ns_per_wave_slot * 2.4);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096 * sizeof(float) + 1024);
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  std::vector<float> h(4096 + 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * static_cast<float>(i % 97);
  hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int wgs : {256, 512, 768}) {
    run<0>("reads + packed FMAs", in, out, wgs, iters, 48);
    run<1>("reads only", in, out, wgs, iters, 48);
    run<2>("packed FMAs only", in, out, wgs, iters, 48);
    run<3>("reads + scalar FMAs", in, out, wgs, iters, 48);
  }
  run<0>("reads + packed FMAs, pitch 45", in, out, 768, iters, 45);
  run<1>("reads only, pitch 45", in, out, 768, iters, 45);
  return 0;
}
