// Per-step classification statistics kept on the device.
//
// The reference's log_stats (model_cross.py:243-255 -> utils.py:18-62) builds six torchmetrics objects and reads six
// scalars back with .item() every training / validation step, plus an AUROC: seven host synchronisations behind a step
// that takes tens of milliseconds.  Lightning then logs the batch-size-weighted epoch mean of those per-step values.
// Here one single-block launch per step folds the step into a small state vector in HBM; the host reads it once per epoch.
//
// state (double[XVIT_METRIC_STATE]):  [0..3] pooled tn, fp, fn, tp;  [4] samples;  [5] steps;
//                                     [6..12] sum over steps of batch_size * {acc, prec, rec, spec, f1, npv, auroc}
#include "xvit_common.h"

namespace xvit {

constexpr int METRIC_MAX_B = 8192;

__global__ __launch_bounds__(256) void binary_metrics_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels, int B,
                                                             double* __restrict__ state) {
  __shared__ float prob[METRIC_MAX_B];
  __shared__ unsigned char lab[METRIC_MAX_B];
  __shared__ unsigned int cnt[4];            // tn, fp, fn, tp of this step
  __shared__ unsigned long long pairs2;      // 2 * #(p_pos > p_neg) + #(p_pos == p_neg)
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0u;
  if (threadIdx.x == 0) pairs2 = 0ull;
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float l0 = logits[b * ld], l1 = logits[b * ld + 1];
    const bool y = labels[b] != 0, pred = l1 > l0;            // argmax; a tie is class 0 (first maximum)
    const float m = fmaxf(l0, l1), e0 = expf(l0 - m), e1 = expf(l1 - m);
    prob[b] = e1 / (e0 + e1);                                 // softmax(logits)[1], as the reference feeds to auroc
    lab[b] = y ? 1 : 0;
    atomicAdd(&cnt[y ? (pred ? 3 : 2) : (pred ? 1 : 0)], 1u);   // tn, fp, fn, tp
  }
  __syncthreads();
  // exact ROC area: pairs (positive i, negative j)
  unsigned long long mine = 0ull;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    if (!lab[i]) continue;
    const float pi = prob[i];
    for (int j = 0; j < B; ++j)
      if (!lab[j]) mine += pi > prob[j] ? 2ull : (pi == prob[j] ? 1ull : 0ull);
  }
  atomicAdd(&pairs2, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tn = cnt[0], fp = cnt[1], fn = cnt[2], tp = cnt[3], n = (double)B;
    auto div = [](double a, double b) { return b > 0.0 ? a / b : 0.0; };   // torchmetrics' _safe_divide
    const double npos = tp + fn, nneg = tn + fp;
    const double m[7] = {div(tp + tn, n), div(tp, tp + fp), div(tp, tp + fn), div(tn, tn + fp), div(2.0 * tp, 2.0 * tp + fp + fn), div(tn, tn + fn),
                         (npos > 0.0 && nneg > 0.0) ? 0.5 * (double)pairs2 / (npos * nneg) : 0.0};
    state[0] += tn; state[1] += fp; state[2] += fn; state[3] += tp;
    state[4] += n;  state[5] += 1.0;
    for (int k = 0; k < 7; ++k) state[6 + k] += n * m[k];
  }
}

}  // namespace xvit

extern "C" int xvit_binary_metrics_step(const float* logits, int64_t ld, const int64_t* labels, int B, int C, double* state, xvit_stream_t stream) {
  XVIT_REQUIRE(logits && labels && state, "xvit_binary_metrics_step: null pointer");
  XVIT_REQUIRE(C == 2 && ld >= 2, "xvit_binary_metrics_step: binary classification only (C = %d, ld = %lld)", C, (long long)ld);
  XVIT_REQUIRE(B > 0 && B <= xvit::METRIC_MAX_B, "xvit_binary_metrics_step: batch %d outside 1..%d", B, xvit::METRIC_MAX_B);
  hipLaunchKernelGGL(xvit::binary_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, B, state);
  return xvit::check_launch("xvit_binary_metrics_step");
}
