// Probe (GPU box): does MIOpen's fusion API build convolution + bias + ReLU plans for the frame's fp32 problems on gfx950,
// which kernels do they run (look at it under rocprofv3 --kernel-trace --stats) and how long do they take?
//   hipcc --offload-arch=gfx950 -O2 tools/miopen_fusion_probe.cpp -lMIOpen -o /tmp/miopen_fusion_probe && /tmp/miopen_fusion_probe
// Result of round 3 is recorded in profiles/r03_sweep_experiment_log.md ("epilogues inside MIOpen").
#include <hip/hip_runtime.h>
#include <miopen/miopen.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK_HIP(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Problem { int cin, h, w, cout, k, stride; };

static const char* status_name(miopenStatus_t s) {
  switch (s) {
    case miopenStatusSuccess: return "success";
    case miopenStatusNotInitialized: return "not initialized";
    case miopenStatusInvalidValue: return "invalid value";
    case miopenStatusBadParm: return "bad parameter";
    case miopenStatusAllocFailed: return "alloc failed";
    case miopenStatusInternalError: return "internal error";
    case miopenStatusNotImplemented: return "not implemented";
    case miopenStatusUnknownError: return "unknown error";
    case miopenStatusUnsupportedOp: return "unsupported op";
    default: return "other";
  }
}

int main() {
  const Problem problems[] = {{16, 12, 20, 16, 3, 1},    // small: checked against a CPU loop
                              {128, 32, 40, 128, 3, 1}, {64, 64, 80, 64, 3, 1}, {64, 64, 80, 64, 5, 1}, {32, 128, 160, 32, 5, 1},
                              {96, 128, 160, 32, 5, 1}, {32, 256, 320, 32, 5, 1}, {64, 64, 80, 128, 3, 2}, {512, 8, 10, 512, 3, 1}};
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  miopenHandle_t handle;
  if (miopenCreateWithStream(&handle, stream) != miopenStatusSuccess) { printf("no MIOpen handle\n"); return 1; }
  for (const Problem& p : problems) {
    const int pad = p.k / 2, ho = (p.h + 2 * pad - p.k) / p.stride + 1, wo = (p.w + 2 * pad - p.k) / p.stride + 1;
    miopenTensorDescriptor_t xd, wd, yd, bd;
    miopenConvolutionDescriptor_t cd;
    miopenCreateTensorDescriptor(&xd); miopenCreateTensorDescriptor(&wd); miopenCreateTensorDescriptor(&yd); miopenCreateTensorDescriptor(&bd);
    miopenSet4dTensorDescriptor(xd, miopenFloat, 1, p.cin, p.h, p.w);
    miopenSet4dTensorDescriptor(wd, miopenFloat, p.cout, p.cin, p.k, p.k);
    miopenSet4dTensorDescriptor(yd, miopenFloat, 1, p.cout, ho, wo);
    miopenSet4dTensorDescriptor(bd, miopenFloat, 1, p.cout, 1, 1);
    miopenCreateConvolutionDescriptor(&cd);
    miopenInitConvolutionDescriptor(cd, miopenConvolution, pad, pad, p.stride, p.stride, 1, 1);
    const size_t nx = static_cast<size_t>(p.cin) * p.h * p.w, nw = static_cast<size_t>(p.cout) * p.cin * p.k * p.k, ny = static_cast<size_t>(p.cout) * ho * wo;
    std::vector<float> hx(nx), hw(nw), hb(p.cout), hy(ny);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (static_cast<float>(s >> 8) / 8388608.0f) - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.1f;
    for (auto& v : hb) v = rnd();
    float *dx, *dw, *db, *dy;
    CHECK_HIP(hipMalloc(&dx, nx * 4)); CHECK_HIP(hipMalloc(&dw, nw * 4)); CHECK_HIP(hipMalloc(&db, p.cout * 4)); CHECK_HIP(hipMalloc(&dy, ny * 4));
    CHECK_HIP(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(db, hb.data(), p.cout * 4, hipMemcpyHostToDevice));

    miopenFusionPlanDescriptor_t plan;
    miopenFusionOpDescriptor_t conv_op, bias_op, act_op;
    miopenCreateFusionPlan(&plan, miopenVerticalFusion, xd);
    miopenStatus_t st = miopenCreateOpConvForward(plan, &conv_op, cd, wd);
    if (st == miopenStatusSuccess) st = miopenCreateOpBiasForward(plan, &bias_op, bd);
    if (st == miopenStatusSuccess) st = miopenCreateOpActivationForward(plan, &act_op, miopenActivationRELU);
    if (st == miopenStatusSuccess) st = miopenCompileFusionPlan(handle, plan);
    printf("conv %dx%d s%d  %d -> %d  %dx%d: fusion plan conv+bias+relu: %s", p.k, p.k, p.stride, p.cin, p.cout, p.h, p.w, status_name(st));
    if (st == miopenStatusSuccess) {
      miopenOperatorArgs_t args;
      miopenCreateOperatorArgs(&args);
      const float one = 1.0f, zero = 0.0f;
      miopenSetOpArgsConvForward(args, conv_op, &one, &zero, dw);
      miopenSetOpArgsBiasForward(args, bias_op, &one, &zero, db);
      miopenSetOpArgsActivForward(args, act_op, &one, &zero, 0.0, 0.0, 0.0);
      st = miopenExecuteFusionPlan(handle, plan, xd, dx, yd, dy, args);
      CHECK_HIP(hipStreamSynchronize(stream));
      printf(", execute: %s", status_name(st));
      if (st == miopenStatusSuccess) {
        hipEvent_t e0, e1;
        CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
        CHECK_HIP(hipEventRecord(e0, stream));
        for (int i = 0; i < 50; ++i) miopenExecuteFusionPlan(handle, plan, xd, dx, yd, dy, args);
        CHECK_HIP(hipEventRecord(e1, stream));
        CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        printf(", %.2f us per call", ms * 1e3f / 50);
        if (nx < 8192) {
          CHECK_HIP(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
          double worst = 0.0;
          for (int co = 0; co < p.cout; ++co)
            for (int y = 0; y < ho; ++y)
              for (int x = 0; x < wo; ++x) {
                double acc = hb[co];
                for (int ci = 0; ci < p.cin; ++ci)
                  for (int ky = 0; ky < p.k; ++ky)
                    for (int kx = 0; kx < p.k; ++kx) {
                      const int yy = y * p.stride + ky - pad, xx = x * p.stride + kx - pad;
                      if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w)
                        acc += static_cast<double>(hx[(static_cast<size_t>(ci) * p.h + yy) * p.w + xx]) * hw[((static_cast<size_t>(co) * p.cin + ci) * p.k + ky) * p.k + kx];
                    }
                const double ref = acc > 0.0 ? acc : 0.0;
                worst = std::fmax(worst, std::fabs(ref - hy[(static_cast<size_t>(co) * ho + y) * wo + x]));
              }
          printf(", max |diff vs CPU loop| %.2e", worst);
        }
      }
    }
    printf("\n");
    fflush(stdout);
    CHECK_HIP(hipFree(dx)); CHECK_HIP(hipFree(dw)); CHECK_HIP(hipFree(db)); CHECK_HIP(hipFree(dy));
  }
  return 0;
}
