// Device helpers shared by the KAN / head kernels (kan_heads.hip, head_phase.hip): the truncated cubic B-spline basis of
// /root/reference/models/kan.py:8-44 in closed form and the activations KANSeverityModule applies (:138-149).
#pragma once
#include "common.h"

namespace {

constexpr int KAN_MAX_KNOTS = 64;

struct Basis4 {
  int j;        // knot interval, -1 when every basis value is zero (beyond the truncation / saturated)
  float v[4];   // value of basis j-m, m = 0..3
};

// knots: LDS or global pointer to nk fp32 knots (uniform by construction, but the STORED values are used
// for the interval search and for h, as the reference does: kan.py:24,33-38).
template <bool DERIV>
__device__ __forceinline__ Basis4 kan_basis(float xn, const float* knots, int nk, float inv_h0, float* dv) {
  Basis4 r;
  const int nb = nk - 4;
  const float t0 = knots[0], tl = knots[nk - 1];
  float xc = fminf(fmaxf(xn, t0), tl);                         // kan.py:16
  int j = (int)floorf((xc - t0) * inv_h0);
  j = j < 0 ? 0 : (j > nk - 1 ? nk - 1 : j);
  while (j > 0 && xc < knots[j]) --j;                          // exact search on the stored knots
  while (j < nk - 1 && xc >= knots[j + 1]) ++j;
  if (j >= nb) {                                               // truncation: SURVEY.md 0.2
    r.j = -1; r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.f;
    if (DERIV) dv[0] = dv[1] = dv[2] = dv[3] = 0.f;
    return r;
  }
  const float tj = knots[j];
  const float h = knots[j + 1] - tj;
  const float u = (xc - tj) / h;
  const float u2 = u * u, u3 = u2 * u, om = 1.f - u;
  r.j = j;
  r.v[0] = u3 * (1.f / 6.f);
  r.v[1] = (-3.f * u3 + 3.f * u2 + 3.f * u + 1.f) * (1.f / 6.f);
  r.v[2] = (3.f * u3 - 6.f * u2 + 4.f) * (1.f / 6.f);
  r.v[3] = om * om * om * (1.f / 6.f);
  if (DERIV) {
    const float ih = 1.f / h;
    dv[0] = 0.5f * u2 * ih;
    dv[1] = (-9.f * u2 + 6.f * u + 3.f) * (1.f / 6.f) * ih;
    dv[2] = (9.f * u2 - 12.f * u) * (1.f / 6.f) * ih;
    dv[3] = -0.5f * om * om * ih;
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
    if (j - m < 0) { r.v[m] = 0.f; if (DERIV) dv[m] = 0.f; }   // left edge loses terms
  return r;
}

__device__ __forceinline__ float act_apply(float z, int act) {
  if (act == ROVIT_ACT_RELU) return fmaxf(z, 0.f);
  if (act == ROVIT_ACT_SIGMOID3) return 3.f / (1.f + __expf(-z));
  return z;
}
// d(act)/dz expressed through the post-activation value y
__device__ __forceinline__ float act_grad(float g, float y, int act) {
  if (act == ROVIT_ACT_RELU) return y > 0.f ? g : 0.f;
  if (act == ROVIT_ACT_SIGMOID3) return g * y * (1.f - y * (1.f / 3.f));
  return g;
}

}  // namespace
