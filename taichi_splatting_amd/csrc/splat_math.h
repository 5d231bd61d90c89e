// splat_math.h — per-element math of the splatting hot path, shared by every gfx950 kernel.
//
// Everything here is a template on the scalar type (float for the product path, double for the
// gradcheck-style tests the reference runs in f64) and is marked MS_HD so that the *same* source
// can also be compiled by g++ into a host-only test shim (tests/hostmath) that validates the
// hand-derived backward chains against the torch oracle on a machine without a GPU.  The reference
// obtains these derivatives from Taichi autodiff (perspective/projection.py:177,
// indexed_spherical_harmonics.py:158); here they are written out by hand.
//
// Reference semantics followed (file:line relative to /root/reference/taichi_splatting):
//   project_gaussian           taichi_lib/generic.py:96-158, perspective/projection.py:33-81
//   eig / ellipse_bounds       taichi_lib/generic.py:217-237
//   gaussian_pdf[_with_grad]   taichi_lib/generic.py:311-336
//   gaussian_pdf_antialias*    taichi_lib/generic.py:341-404
//   obb_grid_query/tile_ranges taichi_lib/grid_query.py:10-91
//   rsh_cart_0..3              indexed_spherical_harmonics.py:38-106
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MS_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MS_HD inline
#endif

namespace ms {

template <typename T> MS_HD T t_min(T a, T b) { return a < b ? a : b; }
template <typename T> MS_HD T t_max(T a, T b) { return a > b ? a : b; }
template <typename T> MS_HD T t_clamp(T x, T lo, T hi) { return t_min(t_max(x, lo), hi); }
MS_HD float t_sqrt(float x) { return sqrtf(x); }
MS_HD double t_sqrt(double x) { return sqrt(x); }
MS_HD float t_exp(float x) { return expf(x); }
MS_HD double t_exp(double x) { return exp(x); }
MS_HD float t_log(float x) { return logf(x); }
MS_HD double t_log(double x) { return log(x); }
MS_HD float t_abs(float x) { return fabsf(x); }
MS_HD double t_abs(double x) { return fabs(x); }
MS_HD float t_floor(float x) { return floorf(x); }
MS_HD double t_floor(double x) { return floor(x); }
MS_HD float t_ceil(float x) { return ceilf(x); }
MS_HD double t_ceil(double x) { return ceil(x); }

// ------------------------------------------------------------------------------------------------
// Camera: 3x4 world->camera matrix (row major) + [fx, fy, cx, cy]
// ------------------------------------------------------------------------------------------------
template <typename T> struct Camera {
  T t[3][4];
  T fx, fy, cx, cy;
};

template <typename T> struct ProjParams {
  T width, height;      // image size as scalars of type T
  T near_plane, far_plane;
  T blur_cov, clamp_margin, alpha_threshold;
};

// Intermediate values of the forward projection that the backward pass re-uses.
template <typename T> struct ProjState {
  T qh[4], qn;          // normalised quaternion (xyzw) and the norm
  T s[3];               // exp(log_scale)
  T pc[3];              // point in camera
  T uv[2], tclamp[2];
  bool clamp_pass[2];   // clamp inactive (gradient flows through t)
  T j00, j11, j02, j12;
  T R[3][3];            // rotation of qh
  T JW[2][3];
  T M[2][3];
  T a, b, c;            // covariance (with blur)
  T tr, gap, sg, l1, l2;
  T v[2], vn;
  T sigma[2], axis[2], alpha;
};

template <typename T>
MS_HD void quat_to_mat(const T q[4], T R[3][3]) {
  // taichi_lib/generic.py:408-416 (xyzw)
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T x2 = x * x, y2 = y * y, z2 = z * z;
  R[0][0] = 1 - 2 * y2 - 2 * z2; R[0][1] = 2 * x * y - 2 * w * z; R[0][2] = 2 * x * z + 2 * w * y;
  R[1][0] = 2 * x * y + 2 * w * z; R[1][1] = 1 - 2 * x2 - 2 * z2; R[1][2] = 2 * y * z - 2 * w * x;
  R[2][0] = 2 * x * z - 2 * w * y; R[2][1] = 2 * y * z + 2 * w * x; R[2][2] = 1 - 2 * x2 - 2 * y2;
}

// Forward projection of one gaussian.  Returns in_view (perspective/projection.py:62-71).
template <typename T>
MS_HD bool project_forward(const T p[3], const T ls[3], const T q[4], T alpha_logit,
                           const Camera<T>& cam, const ProjParams<T>& pp, ProjState<T>& st) {
  st.qn = t_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) st.qh[i] = q[i] / st.qn;
  for (int i = 0; i < 3; ++i) st.s[i] = t_exp(ls[i]);

  for (int i = 0; i < 3; ++i)
    st.pc[i] = cam.t[i][0] * p[0] + cam.t[i][1] * p[1] + cam.t[i][2] * p[2] + cam.t[i][3];

  const T z = st.pc[2];
  st.uv[0] = (cam.fx * st.pc[0]) / z + cam.cx;
  st.uv[1] = (cam.fy * st.pc[1]) / z + cam.cy;

  const T lo0 = -pp.width * pp.clamp_margin, hi0 = (pp.width - 1) * (1 + pp.clamp_margin);
  const T lo1 = -pp.height * pp.clamp_margin, hi1 = (pp.height - 1) * (1 + pp.clamp_margin);
  st.tclamp[0] = t_clamp(st.uv[0], lo0, hi0);
  st.tclamp[1] = t_clamp(st.uv[1], lo1, hi1);
  st.clamp_pass[0] = (st.uv[0] >= lo0) && (st.uv[0] <= hi0);
  st.clamp_pass[1] = (st.uv[1] >= lo1) && (st.uv[1] <= hi1);

  st.j00 = cam.fx / z;
  st.j11 = cam.fy / z;
  st.j02 = -(st.tclamp[0] - cam.cx) / z;
  st.j12 = -(st.tclamp[1] - cam.cy) / z;

  quat_to_mat(st.qh, st.R);

  for (int k = 0; k < 3; ++k) {
    st.JW[0][k] = st.j00 * cam.t[0][k] + st.j02 * cam.t[2][k];
    st.JW[1][k] = st.j11 * cam.t[1][k] + st.j12 * cam.t[2][k];
  }
  for (int r = 0; r < 2; ++r)
    for (int j = 0; j < 3; ++j)
      st.M[r][j] = (st.JW[r][0] * st.R[0][j] + st.JW[r][1] * st.R[1][j] + st.JW[r][2] * st.R[2][j]) * st.s[j];

  st.a = st.M[0][0] * st.M[0][0] + st.M[0][1] * st.M[0][1] + st.M[0][2] * st.M[0][2] + pp.blur_cov;
  st.b = st.M[0][0] * st.M[1][0] + st.M[0][1] * st.M[1][1] + st.M[0][2] * st.M[1][2];
  st.c = st.M[1][0] * st.M[1][0] + st.M[1][1] * st.M[1][1] + st.M[1][2] * st.M[1][2] + pp.blur_cov;

  // eig: taichi_lib/generic.py:217-230 — the same eigen-pair, evaluated without the two cancellations of the
  // reference's formula chain (round 5; the backward has used this form since round 4).  The reference takes
  // sqrt(tr^2 - 4 det) and normalises (a - l2, b): for a >= c that vector adds two positive numbers, for a < c it is
  // (a - c + sg) / 2 with sg ~ c - a, which in float32 loses the small component of a nearly vertical axis (torch_lib's
  // own float32 arithmetic: axis off by 1e-2, the rebuilt covariance by 2.6e-3 on the test protocol).  Here a - c comes
  // from the factors of M (the blur term drops out exactly), l1 - l2 = hypot(a - c, 2 b), and the eigenvector from
  // whichever of (a - l2, b) and +-(b, l1 - a) — the same direction, first component >= 0 — adds two positive
  // numbers.  float64 agrees with the reference fixtures to 1e-12.  a < c with b == 0 exactly gives the axis (0, 1)
  // where the reference divides 0 by 0.
  st.tr = st.a + st.c;
  T amc = T(0);                                   // a - c
  for (int j = 0; j < 3; ++j) amc += (st.M[0][j] - st.M[1][j]) * (st.M[0][j] + st.M[1][j]);
  st.sg = t_sqrt(amc * amc + 4 * st.b * st.b);
  st.gap = st.sg * st.sg;
  st.l1 = (st.tr + st.sg) * T(0.5);
  st.l2 = (st.tr - st.sg) * T(0.5);
  st.sigma[0] = t_sqrt(st.l1);
  st.sigma[1] = t_sqrt(st.l2);
  if (amc >= 0) { st.v[0] = (amc + st.sg) * T(0.5); st.v[1] = st.b; }                                  // (a - l2, b)
  else { st.v[0] = st.b < 0 ? -st.b : st.b; st.v[1] = (st.b < 0 ? T(-0.5) : T(0.5)) * (st.sg - amc); }  // +-(b, l1 - a)
  st.vn = t_sqrt(st.v[0] * st.v[0] + st.v[1] * st.v[1]);
  st.axis[0] = st.v[0] / st.vn;
  st.axis[1] = st.v[1] / st.vn;

  st.alpha = T(1) / (T(1) + t_exp(-alpha_logit));

  // culling: perspective/projection.py:60-68.  alpha < threshold => log < 0 => sqrt NaN => all
  // comparisons false => culled.
  const T gs = t_sqrt(2 * t_log(st.alpha / pp.alpha_threshold));
  const T sx = st.sigma[0] * gs, sy = st.sigma[1] * gs;
  const T v1x = st.axis[0] * sx, v1y = st.axis[1] * sx;
  const T v2x = -st.axis[1] * sy, v2y = st.axis[0] * sy;
  const T ex = t_sqrt(v1x * v1x + v2x * v2x), ey = t_sqrt(v1y * v1y + v2y * v2y);
  const T lower_x = st.uv[0] - ex, upper_x = st.uv[0] + ex;
  const T lower_y = st.uv[1] - ey, upper_y = st.uv[1] + ey;

  return (z > pp.near_plane) && (z < pp.far_plane) && (upper_x > 0) && (upper_y > 0) &&
         (lower_x < pp.width) && (lower_y < pp.height);
}

// Gradients of one projected gaussian.  cam_grad (16 values: 12 of T[3][4] row-major, then
// fx, fy, cx, cy) is *accumulated into*.
template <typename T>
MS_HD void project_backward(const T p[3], const Camera<T>& cam, const ProjState<T>& st,
                            const T g_point[7], T g_depth,
                            T d_pos[3], T d_ls[3], T d_q[4], T& d_alpha_logit, T cam_grad[16],
                            const T* g_cov = nullptr) {
  // g_cov (optional): dL/d(a, b, c) of the blurred 2D covariance, ADDED to what g_point's (axis, sigma) entries give.
  // The fused per-gaussian pass (gaussian_bwd.hip) passes the rasterizer's gradient this way — the plain gaussian pdf
  // depends on (axis, sigma) only through the covariance, so its gradient never needs the eigen-decomposition's
  // derivative at all (no division by l1 - l2 anywhere).
  const T z = st.pc[2];
  const T g_mean[2] = {g_point[0], g_point[1]};
  const T g_axis[2] = {g_point[2], g_point[3]};
  const T g_sigma[2] = {g_point[4], g_point[5]};

  // 1. alpha = sigmoid(alpha_logit)
  d_alpha_logit = g_point[6] * st.alpha * (1 - st.alpha);

  // 2 + 3. (sigma, axis) = eigen-pair of the covariance [[a, b], [b, c]] (taichi_lib/generic.py:217-230), as ONE
  // closed form.  With u = axis, w = perp(u), l1 - l2 = sg, first-order perturbation of a symmetric 2x2 matrix gives
  //   d l1 = u^T dS u,   d l2 = w^T dS w,   d u = (w^T dS u) / (l1 - l2) w
  // hence  dL/dS = gl1 u u^T + gl2 w w^T + kappa sym(w u^T),  kappa = <g_axis, w> / sg.
  // The reference differentiates its formula chain (normalise((a - l2, b)), sqrt(tr^2 - 4 det)) step by step; in
  // float32 that chain divides by |(a - l2, b)| and by sqrt(gap), both of which CANCEL to a few bits for a nearly
  // isotropic covariance (a ~ c, b ~ 0: rows off by 1e-3 .. 1e+3 of the largest gradient, in torch_lib's own float32
  // arithmetic just as here until round 3).  Evaluated as below only ONE ill-conditioned quantity is left, kappa,
  // and its ingredients are formed without cancellation: a - c from the factors of M (the blur term drops out
  // exactly), sg = hypot(a - c, 2 b) instead of sqrt(tr^2 - 4 det), and the eigenvector from whichever of its two
  // equivalent expressions adds two positive numbers.  Same function, same derivative (float64 agrees with the
  // fixtures to 1e-12); what is left in float32 is the conditioning of the problem itself, eps * a / sg.
  const T dl1 = g_sigma[0] / (2 * st.sigma[0]);
  const T dl2 = g_sigma[1] / (2 * st.sigma[1]);
  // (project_forward already holds the pair in this form: sg = hypot(a - c, 2 b), axis from the non-cancelling vector)
  const T sg = st.sg, un = st.vn;
  T u0 = st.axis[0], u1 = st.axis[1];
  T da, db, dc;
  if (st.gap > 0 && un > 0) {
    const T kappa = (g_axis[1] * u0 - g_axis[0] * u1) / sg;      // <g_axis, w> / (l1 - l2), w = (-u1, u0)
    const T uu = u0 * u0, ww = u1 * u1, uw = u0 * u1;
    da = dl1 * uu + dl2 * ww - kappa * uw;
    dc = dl1 * ww + dl2 * uu + kappa * uw;
    db = 2 * uw * (dl1 - dl2) + kappa * (uu - ww);
  } else {
    // gap <= 0 (exactly isotropic): the reference's sqrt(max(gap, 0)) passes no gradient to gap and its axis is
    // normalise((a - tr / 2, b)); keep that branch as the reference has it
    const T dot_ax = st.axis[0] * g_axis[0] + st.axis[1] * g_axis[1];
    const T dvx = (g_axis[0] - st.axis[0] * dot_ax) / st.vn;
    const T dvy = (g_axis[1] - st.axis[1] * dot_ax) / st.vn;
    const T dtr = (dl1 + (dl2 - dvx)) * T(0.5);
    da = dvx + dtr; db = dvy; dc = dtr;
  }
  if (g_cov) { da += g_cov[0]; db += g_cov[1]; dc += g_cov[2]; }

  // 4. cov = M M^T (+ blur)
  T dM[2][3];
  for (int j = 0; j < 3; ++j) {
    dM[0][j] = 2 * da * st.M[0][j] + db * st.M[1][j];
    dM[1][j] = 2 * dc * st.M[1][j] + db * st.M[0][j];
  }

  // 5. M = JW . (R diag(s));  JW = J . W
  T dRS[3][3], dJW[2][3];
  for (int k = 0; k < 3; ++k) {
    for (int j = 0; j < 3; ++j) dRS[k][j] = st.JW[0][k] * dM[0][j] + st.JW[1][k] * dM[1][j];
    for (int r = 0; r < 2; ++r)
      dJW[r][k] = dM[r][0] * st.R[k][0] * st.s[0] + dM[r][1] * st.R[k][1] * st.s[1] + dM[r][2] * st.R[k][2] * st.s[2];
  }
  T dj00 = 0, dj02 = 0, dj11 = 0, dj12 = 0;
  for (int k = 0; k < 3; ++k) {
    dj00 += dJW[0][k] * cam.t[0][k];
    dj02 += dJW[0][k] * cam.t[2][k];
    dj11 += dJW[1][k] * cam.t[1][k];
    dj12 += dJW[1][k] * cam.t[2][k];
    cam_grad[0 * 4 + k] += st.j00 * dJW[0][k];
    cam_grad[1 * 4 + k] += st.j11 * dJW[1][k];
    cam_grad[2 * 4 + k] += st.j02 * dJW[0][k] + st.j12 * dJW[1][k];
  }

  // 6. RS = R diag(s); R = quat_to_mat(qh); qh = q / |q|
  T dR[3][3];
  for (int j = 0; j < 3; ++j) {
    const T ds = dRS[0][j] * st.R[0][j] + dRS[1][j] * st.R[1][j] + dRS[2][j] * st.R[2][j];
    d_ls[j] = ds * st.s[j];
    for (int k = 0; k < 3; ++k) dR[k][j] = dRS[k][j] * st.s[j];
  }
  const T x = st.qh[0], y = st.qh[1], zq = st.qh[2], w = st.qh[3];
  T dqh[4];
  dqh[0] = 2 * (y * dR[0][1] + zq * dR[0][2] + y * dR[1][0] - 2 * x * dR[1][1] - w * dR[1][2] + zq * dR[2][0] + w * dR[2][1] - 2 * x * dR[2][2]);
  dqh[1] = 2 * (-2 * y * dR[0][0] + x * dR[0][1] + w * dR[0][2] + x * dR[1][0] + zq * dR[1][2] - w * dR[2][0] + zq * dR[2][1] - 2 * y * dR[2][2]);
  dqh[2] = 2 * (-2 * zq * dR[0][0] - w * dR[0][1] + x * dR[0][2] + w * dR[1][0] - 2 * zq * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
  dqh[3] = 2 * (-zq * dR[0][1] + y * dR[0][2] + zq * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
  const T qdot = st.qh[0] * dqh[0] + st.qh[1] * dqh[1] + st.qh[2] * dqh[2] + st.qh[3] * dqh[3];
  for (int i = 0; i < 4; ++i) d_q[i] = (dqh[i] - st.qh[i] * qdot) / st.qn;

  // 7. jacobian entries
  T dz = g_depth;
  T dfx = dj00 / z, dfy = dj11 / z;
  dz -= (dj00 * st.j00 + dj11 * st.j11 + dj02 * st.j02 + dj12 * st.j12) / z;
  const T dtx = -dj02 / z, dty = -dj12 / z;
  T dcx = dj02 / z, dcy = dj12 / z;

  // 8. uv = f * pc.xy / z + c ; t = clamp(uv)
  const T duvx = g_mean[0] + (st.clamp_pass[0] ? dtx : T(0));
  const T duvy = g_mean[1] + (st.clamp_pass[1] ? dty : T(0));
  dfx += duvx * st.pc[0] / z;
  dfy += duvy * st.pc[1] / z;
  dcx += duvx;
  dcy += duvy;
  T dpc[3];
  dpc[0] = duvx * cam.fx / z;
  dpc[1] = duvy * cam.fy / z;
  dz -= (duvx * cam.fx * st.pc[0] + duvy * cam.fy * st.pc[1]) / (z * z);
  dpc[2] = dz;

  // 9. pc = T [p; 1]
  for (int k = 0; k < 3; ++k)
    d_pos[k] = dpc[0] * cam.t[0][k] + dpc[1] * cam.t[1][k] + dpc[2] * cam.t[2][k];
  for (int i = 0; i < 3; ++i) {
    for (int k = 0; k < 3; ++k) cam_grad[i * 4 + k] += dpc[i] * p[k];
    cam_grad[i * 4 + 3] += dpc[i];
  }
  cam_grad[12] += dfx;
  cam_grad[13] += dfy;
  cam_grad[14] += dcx;
  cam_grad[15] += dcy;
}

// ------------------------------------------------------------------------------------------------
// Spherical harmonics (real, cartesian, degree <= 3)
// ------------------------------------------------------------------------------------------------
#define MS_SH_C0 0.282094791773878
#define MS_SH_C1 0.48860251190292
#define MS_SH_C2A 1.09254843059208
#define MS_SH_C2B 0.94617469575756
#define MS_SH_C2C 0.31539156525252
#define MS_SH_C2D 0.54627421529604
#define MS_SH_C3A 0.590043589926644
#define MS_SH_C3B 2.89061144264055
#define MS_SH_C3C 0.304697199642977
#define MS_SH_C3D 1.24392110863372
#define MS_SH_C3E 0.497568443453487
#define MS_SH_C3F 1.44530572132028

// Y[0 .. (DEG+1)^2) at unit direction (x, y, z): indexed_spherical_harmonics.py:38-106
template <typename T, int DEG>
MS_HD void sh_basis(T x, T y, T z, T* Y) {
  Y[0] = T(MS_SH_C0);
  if (DEG >= 1) {
    Y[1] = T(-MS_SH_C1) * y;
    Y[2] = T(MS_SH_C1) * z;
    Y[3] = T(-MS_SH_C1) * x;
  }
  if (DEG >= 2) {
    const T x2 = x * x, y2 = y * y, z2 = z * z;
    Y[4] = T(MS_SH_C2A) * (x * y);
    Y[5] = T(-MS_SH_C2A) * (y * z);
    Y[6] = T(MS_SH_C2B) * z2 - T(MS_SH_C2C);
    Y[7] = T(-MS_SH_C2A) * (x * z);
    Y[8] = T(MS_SH_C2D) * x2 - T(MS_SH_C2D) * y2;
    if (DEG >= 3) {
      Y[9] = T(-MS_SH_C3A) * y * (T(3) * x2 - y2);
      Y[10] = T(MS_SH_C3B) * (x * y) * z;
      Y[11] = T(MS_SH_C3C) * y * (T(1.5) - T(7.5) * z2);
      Y[12] = T(MS_SH_C3D) * z * (T(1.5) * z2 - T(0.5)) - T(MS_SH_C3E) * z;
      Y[13] = T(MS_SH_C3C) * x * (T(1.5) - T(7.5) * z2);
      Y[14] = T(MS_SH_C3F) * z * (x2 - y2);
      Y[15] = T(-MS_SH_C3A) * x * (x2 - T(3) * y2);
    }
  }
}

// d_dir += sum_d c[d] * grad Y_d(x, y, z)  (polynomial gradient, dir treated as free in R^3)
template <typename T, int DEG>
MS_HD void sh_basis_grad_dot(T x, T y, T z, const T* c, T g[3]) {
  g[0] = g[1] = g[2] = T(0);
  if (DEG >= 1) {
    g[1] += c[1] * T(-MS_SH_C1);
    g[2] += c[2] * T(MS_SH_C1);
    g[0] += c[3] * T(-MS_SH_C1);
  }
  if (DEG >= 2) {
    g[0] += c[4] * T(MS_SH_C2A) * y;
    g[1] += c[4] * T(MS_SH_C2A) * x;
    g[1] += c[5] * T(-MS_SH_C2A) * z;
    g[2] += c[5] * T(-MS_SH_C2A) * y;
    g[2] += c[6] * T(2 * MS_SH_C2B) * z;
    g[0] += c[7] * T(-MS_SH_C2A) * z;
    g[2] += c[7] * T(-MS_SH_C2A) * x;
    g[0] += c[8] * T(2 * MS_SH_C2D) * x;
    g[1] += c[8] * T(-2 * MS_SH_C2D) * y;
  }
  if (DEG >= 3) {
    const T x2 = x * x, y2 = y * y, z2 = z * z;
    g[0] += c[9] * T(-MS_SH_C3A) * T(6) * x * y;
    g[1] += c[9] * T(-MS_SH_C3A) * (T(3) * x2 - T(3) * y2);
    g[0] += c[10] * T(MS_SH_C3B) * y * z;
    g[1] += c[10] * T(MS_SH_C3B) * x * z;
    g[2] += c[10] * T(MS_SH_C3B) * x * y;
    g[1] += c[11] * T(MS_SH_C3C) * (T(1.5) - T(7.5) * z2);
    g[2] += c[11] * T(MS_SH_C3C) * T(-15) * y * z;
    g[2] += c[12] * (T(MS_SH_C3D) * (T(4.5) * z2 - T(0.5)) - T(MS_SH_C3E));
    g[0] += c[13] * T(MS_SH_C3C) * (T(1.5) - T(7.5) * z2);
    g[2] += c[13] * T(MS_SH_C3C) * T(-15) * x * z;
    g[0] += c[14] * T(MS_SH_C3F) * T(2) * x * z;
    g[1] += c[14] * T(-MS_SH_C3F) * T(2) * y * z;
    g[2] += c[14] * T(MS_SH_C3F) * (x2 - y2);
    g[0] += c[15] * T(-MS_SH_C3A) * (T(3) * x2 - T(3) * y2);
    g[1] += c[15] * T(MS_SH_C3A) * T(6) * x * y;
  }
}

// ------------------------------------------------------------------------------------------------
// 2D gaussian pdfs
// ------------------------------------------------------------------------------------------------
template <typename T>
MS_HD T gaussian_pdf(T px, T py, const T* g /*mean2 axis2 sigma2*/) {
  const T dx = px - g[0], dy = py - g[1];
  const T tx = (dx * g[2] + dy * g[3]) / g[4];
  const T ty = (dx * -g[3] + dy * g[2]) / g[5];
  return t_exp(T(-0.5) * (tx * tx + ty * ty));
}

// returns p; fills dmean[2], daxis[2], dsigma[2]  (taichi_lib/generic.py:321-336)
template <typename T>
MS_HD T gaussian_pdf_with_grad(T px, T py, const T* g, T dmean[2], T daxis[2], T dsigma[2]) {
  const T dx = px - g[0], dy = py - g[1];
  const T ax = g[2], ay = g[3], sx = g[4], sy = g[5];
  const T tx = (dx * ax + dy * ay) / sx;
  const T ty = (dx * -ay + dy * ax) / sy;
  const T tx2 = tx * tx, ty2 = ty * ty;
  const T p = t_exp(T(-0.5) * (tx2 + ty2));
  dsigma[0] = tx2 * p / sx;
  dsigma[1] = ty2 * p / sy;
  const T tx_s = tx / sx, ty_s = ty / sy;
  // perp(v) = (-v.y, v.x)
  daxis[0] = p * (tx_s * -dx + ty_s * -dy);
  daxis[1] = p * (tx_s * -dy + ty_s * dx);
  dmean[0] = p * (tx_s * ax + ty_s * -ay);
  dmean[1] = p * (tx_s * ay + ty_s * ax);
  return p;
}

template <typename T>
MS_HD T s_sig(T x, T sigma) {
  const T z = x / sigma;
  return T(1) / (T(1) + t_exp(T(-1.6) * z - T(0.07) * z * z * z));
}

template <typename T>
MS_HD T gaussian_pdf_antialias(T px, T py, const T* g) {
  const T dx = px - g[0], dy = py - g[1];
  const T sx = g[4], sy = g[5];
  const T tx = dx * g[2] + dy * g[3];
  const T ty = dx * -g[3] + dy * g[2];
  const T Sx1 = s_sig(tx + T(0.5), sx), Sx2 = s_sig(tx - T(0.5), sx);
  const T Sy1 = s_sig(ty + T(0.5), sy), Sy2 = s_sig(ty - T(0.5), sy);
  return T(2 * 3.14159265358979323846) * sx * (Sx1 - Sx2) * sy * (Sy1 - Sy2);
}

template <typename T>
MS_HD void s_sig_grad(T x, T sigma, T& s, T& ds_dx, T& ds_dsigma) {
  const T z = x / sigma;
  s = T(1) / (T(1) + t_exp(T(-1.6) * z - T(0.07) * z * z * z));
  const T ds = (T(1.6) + T(0.21) * z * z) * s * (T(1) - s);
  ds_dx = ds / sigma;
  ds_dsigma = ds_dx * -z;
}

// taichi_lib/generic.py:371-404
template <typename T>
MS_HD T gaussian_pdf_antialias_with_grad(T px, T py, const T* g, T dmean[2], T daxis[2], T dsigma[2]) {
  const T dx = px - g[0], dy = py - g[1];
  const T ax = g[2], ay = g[3], sx = g[4], sy = g[5];
  const T tx = dx * ax + dy * ay;
  const T ty = dx * -ay + dy * ax;

  T Sx1, dSx1, dSx1s, Sx2, dSx2, dSx2s, Sy1, dSy1, dSy1s, Sy2, dSy2, dSy2s;
  s_sig_grad(tx + T(0.5), sx, Sx1, dSx1, dSx1s);
  s_sig_grad(tx - T(0.5), sx, Sx2, dSx2, dSx2s);
  s_sig_grad(ty + T(0.5), sy, Sy1, dSy1, dSy1s);
  s_sig_grad(ty - T(0.5), sy, Sy2, dSy2, dSy2s);

  const T ix = sx * (Sx1 - Sx2);
  const T iy = sy * (Sy1 - Sy2);
  const T tau = T(2 * 3.14159265358979323846);
  const T i2d = tau * ix * iy;

  const T dSx = iy * sx * (dSx1 - dSx2);
  const T dSy = ix * sy * (dSy1 - dSy2);

  // di_dmean = tau * (dSx * -axis + dSy * -perp(axis))
  dmean[0] = tau * (dSx * -ax + dSy * ay);
  dmean[1] = tau * (dSx * -ay + dSy * -ax);
  dsigma[0] = tau * iy * (Sx1 - Sx2 + (dSx1s - dSx2s) * sx);
  dsigma[1] = tau * ix * (Sy1 - Sy2 + (dSy1s - dSy2s) * sy);
  // di_daxis = tau * (dSx * d + dSy * -perp(d)), perp(d) = (-dy, dx)
  daxis[0] = tau * (dSx * dx + dSy * dy);
  daxis[1] = tau * (dSx * dy + dSy * -dx);
  return i2d;
}

// ------------------------------------------------------------------------------------------------
// OBB-vs-tile query (taichi_lib/grid_query.py:10-91).  float only, as in the reference.
// ------------------------------------------------------------------------------------------------
struct ObbQuery {
  float inv00, inv01, inv10, inv11;    // rows: axis1 / scale.x, axis2 / scale.y
  float rel_min_x, rel_min_y;          // min_tile * tile_size - mean
  int min_tile_x, min_tile_y;
  int span_x, span_y;                  // may be <= 0
};

MS_HD ObbQuery obb_grid_query(const float* g /*7*/, int image_w, int image_h, int tile_size,
                              float alpha_threshold) {
  const float mx = g[0], my = g[1], ax = g[2], ay = g[3], sgx = g[4], sgy = g[5], alpha = g[6];
  const float gs = sqrtf(2.0f * logf(alpha / alpha_threshold));
  const float sx = sgx * gs, sy = sgy * gs;
  const float a2x = -ay, a2y = ax;

  // ellipse_bounds(mean, axis1 * scale.x, axis2 * scale.y)
  const float v1x = ax * sx, v1y = ay * sx, v2x = a2x * sy, v2y = a2y * sy;
  const float ex = sqrtf(v1x * v1x + v2x * v2x), ey = sqrtf(v1y * v1y + v2y * v2y);
  const float min_x = mx - ex, min_y = my - ey, max_x = mx + ex, max_y = my + ey;

  ObbQuery q;
  q.inv00 = ax / sx; q.inv01 = ay / sx;
  q.inv10 = a2x / sy; q.inv11 = a2y / sy;

  const float ts = (float)tile_size;
  const int max_tile_x = (image_w - 1) / tile_size, max_tile_y = (image_h - 1) / tile_size;
  // NaN bounds (alpha < threshold) convert to INT_MIN-like values on the device; guard so the
  // span is empty, matching "all comparisons false" culling.
  int lo_x = (min_x == min_x) ? (int)floorf(min_x / ts) : 0x3fffffff;
  int lo_y = (min_y == min_y) ? (int)floorf(min_y / ts) : 0x3fffffff;
  lo_x = lo_x > 0 ? lo_x : 0;
  lo_y = lo_y > 0 ? lo_y : 0;
  int hi_x = (max_x == max_x) ? (int)ceilf(max_x / ts) : 0;
  int hi_y = (max_y == max_y) ? (int)ceilf(max_y / ts) : 0;
  hi_x = hi_x > lo_x + 1 ? hi_x : lo_x + 1;
  hi_y = hi_y > lo_y + 1 ? hi_y : lo_y + 1;
  hi_x = hi_x < max_tile_x + 1 ? hi_x : max_tile_x + 1;
  hi_y = hi_y < max_tile_y + 1 ? hi_y : max_tile_y + 1;

  q.min_tile_x = lo_x; q.min_tile_y = lo_y;
  q.span_x = hi_x - lo_x; q.span_y = hi_y - lo_y;
  q.rel_min_x = (float)(lo_x * tile_size) - mx;
  q.rel_min_y = (float)(lo_y * tile_size) - my;
  return q;
}

// not separates_bbox(inv_basis, lower, lower + tile_size)
MS_HD bool obb_test_tile(const ObbQuery& q, int u, int v, int tile_size) {
  const float lx = q.rel_min_x + (float)(u * tile_size), ly = q.rel_min_y + (float)(v * tile_size);
  const float ux = lx + (float)tile_size, uy = ly + (float)tile_size;
  // corners: (lx,ly) (ux,ly) (ux,uy) (lx,uy)
  bool separates = false;
  {
    const float p0 = q.inv00 * lx + q.inv01 * ly, p1 = q.inv00 * ux + q.inv01 * ly;
    const float p2 = q.inv00 * ux + q.inv01 * uy, p3 = q.inv00 * lx + q.inv01 * uy;
    const float mn = fminf(fminf(p0, p1), fminf(p2, p3)), mxv = fmaxf(fmaxf(p0, p1), fmaxf(p2, p3));
    if (mn > 1.0f || mxv < -1.0f) separates = true;
  }
  {
    const float p0 = q.inv10 * lx + q.inv11 * ly, p1 = q.inv10 * ux + q.inv11 * ly;
    const float p2 = q.inv10 * ux + q.inv11 * uy, p3 = q.inv10 * lx + q.inv11 * uy;
    const float mn = fminf(fminf(p0, p1), fminf(p2, p3)), mxv = fmaxf(fmaxf(p0, p1), fmaxf(p2, p3));
    if (mn > 1.0f || mxv < -1.0f) separates = true;
  }
  return !separates;
}

}  // namespace ms
