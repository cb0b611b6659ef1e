// Fused joint multi-task loss, forward + gradient in one launch.
//
// Reference being restated: /root/reference/training/losses.py
//   FocalLoss.forward          :15-38   alpha_t (1 - p_t)^gamma * CE, mean over the batch
//   OrdinalBCELoss.forward     :48-72   BCE-with-logits against (target > k), mean over thresholds then batch
//   UncertaintyLoss.forward    :80-101  0.5 ((y - mu)^2 exp(-s) + s), mean
//   KANRegressionLoss.forward  :109-114 MSE
//   JointLoss.forward          :139-181 total = cls + lambda*ord (stage>=2) + mu*unc (stage>=3) + nu*kan (stage>=4)
// One workgroup walks the batch (B is a few hundred rows, 4+3+1+1+1 values per row); the per-row gradients of the
// TOTAL loss w.r.t. every head output are written in the same pass, so backward is a single scale by the upstream
// gradient instead of ~25 elementwise/reduction launches.
#include "common.h"

namespace {

constexpr int MAXC = 16;

struct LossArgs {
  const float* cls; const float* ord; const float* mu; const float* lv; const float* kan;
  const long long* cls_t; const void* sev_t; int sev_i64; const float* alpha;
  float* d_cls; float* d_ord; float* d_mu; float* d_lv; float* d_kan;
  float* out;      // [5]: cls, ord, unc, kan, total
  int B, C;
  float lambda_ord, mu_unc, nu_kan, gamma;
};

__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = wave_sum64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ __launch_bounds__(256) void joint_loss_kernel(const LossArgs a) {
  __shared__ float s_red[4];
  const float invB = 1.f / a.B;
  float l_cls = 0.f, l_ord = 0.f, l_unc = 0.f, l_kan = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    // an out-of-range class label would index z[] / alpha[] out of bounds: clamp it and poison the losses with NaN so
    // that the caller sees it (torch's cross_entropy raises a device assert in that case)
    const long long t_raw = a.cls_t[b];
    const bool t_ok = t_raw >= 0 && t_raw < a.C;
    const int t = t_ok ? (int)t_raw : 0;
    if (!t_ok) l_cls = __builtin_nanf("");
    // severity as float, like the reference's .float() casts (:89-90, :110-111); int64 labels are converted here, not by a copy launch
    const float y = a.sev_i64 ? (float)((const long long*)a.sev_t)[b] : ((const float*)a.sev_t)[b];
    // ---- focal cross-entropy (losses.py:15-38) ----
    float z[MAXC];
    float zmax = -INFINITY;
    for (int j = 0; j < a.C; ++j) { z[j] = a.cls[(size_t)b * a.C + j]; zmax = fmaxf(zmax, z[j]); }
    float se = 0.f;
    for (int j = 0; j < a.C; ++j) se += __expf(z[j] - zmax);
    const float lse = zmax + __logf(se);
    const float logpt = z[t] - lse;
    const float pt = __expf(logpt);
    const float al = a.alpha ? a.alpha[t] : 1.f;
    const float om = 1.f - pt;
    const float fg = __powf(fmaxf(om, 0.f), a.gamma);                // (1 - p_t)^gamma
    l_cls += al * fg * (-logpt);
    // d/dz_j = alpha [gamma p_t (1-p_t)^(gamma-1) log p_t - (1-p_t)^gamma] (delta_jt - p_j) / B
    const float fgm1 = a.gamma == 0.f ? 0.f : a.gamma * __powf(fmaxf(om, 1e-30f), a.gamma - 1.f);
    const float coef = al * (fgm1 * pt * logpt - fg) * invB;
    for (int j = 0; j < a.C; ++j) {
      const float pj = __expf(z[j] - lse);
      a.d_cls[(size_t)b * a.C + j] = coef * ((j == t ? 1.f : 0.f) - pj);
    }
    // ---- ordinal BCE (losses.py:48-72) ----
    if (a.ord) {
      const int K1 = a.C - 1;
      const float w = a.lambda_ord * invB / K1;
      float acc = 0.f;
      for (int k = 0; k < K1; ++k) {
        const float x = a.ord[(size_t)b * K1 + k];
        const float yt = y > (float)k ? 1.f : 0.f;                  // (targets > k).float(), losses.py:55-56
        acc += fmaxf(x, 0.f) - x * yt + __logf(1.f + __expf(-fabsf(x)));      // stable BCE-with-logits
        a.d_ord[(size_t)b * K1 + k] = w * (1.f / (1.f + __expf(-x)) - yt);
      }
      l_ord += acc / K1;
    }
    // ---- heteroscedastic regression (losses.py:80-101) ----
    if (a.mu) {
      const float m = a.mu[b], s = a.lv[b];
      const float prec = __expf(-s), r = y - m;
      l_unc += 0.5f * (r * r * prec + s);
      a.d_mu[b] = -a.mu_unc * invB * r * prec;
      a.d_lv[b] = a.mu_unc * invB * 0.5f * (1.f - r * r * prec);
    }
    // ---- KAN severity regression (losses.py:109-114) ----
    if (a.kan) {
      const float r = a.kan[b] - y;
      l_kan += r * r;
      a.d_kan[b] = a.nu_kan * invB * 2.f * r;
    }
  }
  const float s_cls = block_sum(l_cls, s_red) * invB;
  const float s_ord = block_sum(l_ord, s_red) * invB;
  const float s_unc = block_sum(l_unc, s_red) * invB;
  const float s_kan = block_sum(l_kan, s_red) * invB;
  if (threadIdx.x == 0) {
    a.out[0] = s_cls; a.out[1] = s_ord; a.out[2] = s_unc; a.out[3] = s_kan;
    a.out[4] = s_cls + (a.ord ? a.lambda_ord * s_ord : 0.f) + (a.mu ? a.mu_unc * s_unc : 0.f) + (a.kan ? a.nu_kan * s_kan : 0.f);
  }
}

// g[i] *= *scale for up to 5 small buffers (the backward of the loss: chain with the upstream gradient)
struct ScaleArgs { float* p[5]; int n[5]; const float* scale; };
__global__ __launch_bounds__(256) void scale_buffers_kernel(const ScaleArgs a) {
  const float s = *a.scale;
  for (int k = 0; k < 5; ++k)
    if (a.p[k])
      for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n[k]; i += gridDim.x * 256) a.p[k][i] *= s;
}

}  // namespace

// Inactive heads: pass NULL for (ord, d_ord) / (mu, lv, d_mu, d_lv) / (kan, d_kan) -- that is the curriculum gate.
// d_* receive d(total)/d(output) for an upstream gradient of 1.  losses_out: [cls, ord, unc, kan, total].
extern "C" int rovit_joint_loss(const float* cls_logits, const float* ordinal_logits, const float* mu, const float* log_var,
                                const float* kan_severity, const long long* class_targets, const void* severity_targets,
                                int severity_is_int64, const float* focal_alpha, float* d_cls, float* d_ord, float* d_mu, float* d_lv, float* d_kan,
                                float* losses_out, int batch, int num_classes, float lambda_ord, float mu_unc, float nu_kan,
                                float focal_gamma, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(cls_logits && class_targets && severity_targets && d_cls && losses_out, ROVIT_ERR_NULL, "joint_loss: null pointer");
  ROVIT_CHECK_ARG(batch > 0 && num_classes >= 2 && num_classes <= MAXC, ROVIT_ERR_SHAPE, "joint_loss: bad batch/classes");
  ROVIT_CHECK_ARG((!ordinal_logits || d_ord) && (!mu || (log_var && d_mu && d_lv)) && (!kan_severity || d_kan), ROVIT_ERR_NULL,
                  "joint_loss: gradient buffer missing for an active head");
  LossArgs a{cls_logits, ordinal_logits, mu, log_var, kan_severity, class_targets, severity_targets, severity_is_int64 ? 1 : 0, focal_alpha,
             d_cls, d_ord, d_mu, d_lv, d_kan, losses_out, batch, num_classes, lambda_ord, mu_unc, nu_kan, focal_gamma};
  hipLaunchKernelGGL(joint_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("joint_loss_kernel");
  return ROVIT_OK;
}

extern "C" int rovit_scale_buffers(float* const* bufs, const int* counts, int n, const float* scale, rovit_stream_t stream) {
  ROVIT_CHECK_ARG(bufs && counts && scale && n >= 1 && n <= 5, ROVIT_ERR_SHAPE, "scale_buffers: 1..5 buffers");
  ScaleArgs a{};
  int mx = 0;
  for (int i = 0; i < n; ++i) { a.p[i] = bufs[i]; a.n[i] = counts[i]; mx = counts[i] > mx ? counts[i] : mx; }
  a.scale = scale;
  hipLaunchKernelGGL(scale_buffers_kernel, dim3((mx + 255) / 256 > 0 ? (mx + 255) / 256 : 1), dim3(256), 0, (hipStream_t)stream, a);
  ROVIT_CHECK_LAUNCH("scale_buffers_kernel");
  return ROVIT_OK;
}
