// Dense convolution with its bias add and ReLU inside MIOpen's own kernel (fusion plan: convolution + bias [+ activation]).
//
// (Round 3; since round 6 no layer of a default engine takes this path: the dense, bottleneck and 1x1 layers have their own kernels.)
// A convolution that stays on MIOpen; this entry only changes WHERE its epilogue runs: for the problems MIOpen
// solves with its fp32 Winograd kernel the fused plan is the same kernel with the epilogue in its store path, i.e. one launch
// instead of convolution + dvmvs_bias_act_fwd.  For other problems (those MIOpen's search gives to a GEMM or implicit-GEMM
// solver) the fused plan is slower than the two launches; the caller decides per problem by timing both at warm-up
// (dvmvs/engine.py, FusedConv2d), and a plan MIOpen cannot build is reported as DVMVS_EUNSUPPORTED, never emulated.
//
// Plans, descriptors and the MIOpen handle are cached per device; the call itself only binds pointers and launches, so it can
// be recorded into a hipGraph once the plan exists (the first call of a problem compiles: make it before capturing).
#include <miopen/miopen.h>

#include <array>
#include <map>
#include <mutex>

#include "dvmvs_device.h"

namespace {

struct FusedConvolution {
  miopenTensorDescriptor_t x = nullptr, w = nullptr, y = nullptr, bias = nullptr;
  miopenConvolutionDescriptor_t conv = nullptr;
  miopenFusionPlanDescriptor_t plan = nullptr;
  miopenFusionOpDescriptor_t conv_op = nullptr, bias_op = nullptr, act_op = nullptr;
  miopenOperatorArgs_t args = nullptr;
  bool usable = false;
};

using ProblemKey = std::array<long long, 13>;

std::mutex g_mutex;
std::map<int, miopenHandle_t> g_handles;            // one per device; the stream is set per call
std::map<ProblemKey, FusedConvolution> g_problems;

bool ok(miopenStatus_t s) { return s == miopenStatusSuccess; }

// a problem without a usable plan keeps no MIOpen objects (the map entry only remembers the answer)
void release(FusedConvolution& p) {
  if (p.plan) miopenDestroyFusionPlan(p.plan);
  if (p.args) miopenDestroyOperatorArgs(p.args);
  if (p.conv) miopenDestroyConvolutionDescriptor(p.conv);
  if (p.x) miopenDestroyTensorDescriptor(p.x);
  if (p.w) miopenDestroyTensorDescriptor(p.w);
  if (p.y) miopenDestroyTensorDescriptor(p.y);
  if (p.bias) miopenDestroyTensorDescriptor(p.bias);
  p = FusedConvolution();
}

// conv + bias + activation; for "no activation" first the two-operator plan, then a pass-through activation
bool build_plan(miopenHandle_t handle, FusedConvolution& p, int activation, bool pass_through) {
  if (!ok(miopenCreateFusionPlan(&p.plan, miopenVerticalFusion, p.x))) return false;
  bool good = ok(miopenCreateOpConvForward(p.plan, &p.conv_op, p.conv, p.w)) && ok(miopenCreateOpBiasForward(p.plan, &p.bias_op, p.bias));
  p.act_op = nullptr;
  if (good && (activation == 1 || pass_through))
    good = ok(miopenCreateOpActivationForward(p.plan, &p.act_op, activation == 1 ? miopenActivationRELU : miopenActivationPASTHRU));
  good = good && ok(miopenCompileFusionPlan(handle, p.plan));
  if (!good) {
    miopenDestroyFusionPlan(p.plan);
    p.plan = nullptr;
  }
  return good;
}

}  // namespace

extern "C" int dvmvs_conv_bias_act_fwd(const float* x, const float* weight, const float* bias, float* out, long long out_batch_stride,
                                       int B, int Cin, int H, int W, int Cout, int K, int stride, int padding, int activation,
                                       dvmvs_stream_t stream) {
  if (!x || !weight || !bias || !out) return DVMVS_EINVAL;
  if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || K <= 0 || stride <= 0 || padding < 0) return DVMVS_EINVAL;
  if (activation < 0 || activation > 1) return DVMVS_EINVAL;
  const int Ho = (H + 2 * padding - K) / stride + 1, Wo = (W + 2 * padding - K) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return DVMVS_EINVAL;
  const long long dense = static_cast<long long>(Cout) * Ho * Wo;
  if (out_batch_stride == 0) out_batch_stride = dense;
  if (out_batch_stride < dense) return DVMVS_EINVAL;
  if (B == 1) out_batch_stride = dense;          // no second batch item: every destination is a packed tensor
  int device = 0;
  DVMVS_RETURN_IF_HIP(hipGetDevice(&device));

  std::lock_guard<std::mutex> lock(g_mutex);
  miopenHandle_t& handle = g_handles[device];
  if (!handle && !ok(miopenCreateWithStream(&handle, static_cast<hipStream_t>(stream)))) {
    handle = nullptr;
    return DVMVS_ELIBRARY;
  }
  const ProblemKey key = {device, B, Cin, H, W, Cout, K, stride, padding, activation, out_batch_stride, 0, 0};
  auto found = g_problems.find(key);
  if (found == g_problems.end()) {
    FusedConvolution p;
    const int ydims[4] = {B, Cout, Ho, Wo};
    const int ystrides[4] = {static_cast<int>(out_batch_stride), Ho * Wo, Wo, 1};
    bool good = out_batch_stride < (1LL << 31) && ok(miopenCreateTensorDescriptor(&p.x)) && ok(miopenCreateTensorDescriptor(&p.w)) &&
                ok(miopenCreateTensorDescriptor(&p.y)) && ok(miopenCreateTensorDescriptor(&p.bias)) &&
                ok(miopenCreateConvolutionDescriptor(&p.conv)) && ok(miopenSet4dTensorDescriptor(p.x, miopenFloat, B, Cin, H, W)) &&
                ok(miopenSet4dTensorDescriptor(p.w, miopenFloat, Cout, Cin, K, K)) &&
                ok(miopenSetTensorDescriptor(p.y, miopenFloat, 4, ydims, ystrides)) &&
                ok(miopenSet4dTensorDescriptor(p.bias, miopenFloat, 1, Cout, 1, 1)) &&
                ok(miopenInitConvolutionDescriptor(p.conv, miopenConvolution, padding, padding, stride, stride, 1, 1)) &&
                ok(miopenCreateOperatorArgs(&p.args));
    if (good) {
      if (!ok(miopenSetStream(handle, static_cast<hipStream_t>(stream)))) {
        release(p);
        return DVMVS_ELIBRARY;
      }
      good = build_plan(handle, p, activation, false) || (activation == 0 && build_plan(handle, p, activation, true));
    }
    if (!good) release(p);
    p.usable = good;
    found = g_problems.emplace(key, p).first;      // failures are remembered too: the answer does not change
  }
  FusedConvolution& p = found->second;
  if (!p.usable) return DVMVS_EUNSUPPORTED;
  const float one = 1.0f, zero = 0.0f;
  if (!ok(miopenSetStream(handle, static_cast<hipStream_t>(stream)))) return DVMVS_ELIBRARY;
  if (!ok(miopenSetOpArgsConvForward(p.args, p.conv_op, &one, &zero, weight))) return DVMVS_ELIBRARY;
  if (!ok(miopenSetOpArgsBiasForward(p.args, p.bias_op, &one, &zero, bias))) return DVMVS_ELIBRARY;
  if (p.act_op && !ok(miopenSetOpArgsActivForward(p.args, p.act_op, &one, &zero, 0.0, 0.0, 0.0))) return DVMVS_ELIBRARY;
  if (!ok(miopenExecuteFusionPlan(handle, p.plan, p.x, x, p.y, out, p.args))) return DVMVS_ELIBRARY;
  return dvmvs::launch_status();
}
