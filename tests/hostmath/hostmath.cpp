// Host-only compilation of csrc/splat_math.h for the CPU test-suite (TEST INFRASTRUCTURE).
// The same templated source that the gfx950 kernels instantiate is built here with g++ so the
// hand-derived backward chains can be checked against the torch oracle without a GPU.  The product
// never loads this library.
#include <stdint.h>
#include "../../taichi_splatting_amd/csrc/splat_math.h"

using namespace ms;

static Camera<double> make_cam(const double* T, const double* P) {
  Camera<double> cam;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) cam.t[i][j] = T[i * 4 + j];
  cam.fx = P[0]; cam.fy = P[1]; cam.cx = P[2]; cam.cy = P[3];
  return cam;
}

extern "C" void hm_project_fwd(const double* pos, const double* ls, const double* rot, const double* al,
                               const double* T, const double* P, int W, int H, double near_plane, double far_plane,
                               double blur, double margin, double thr, int64_t n, double* points7, double* depth,
                               int32_t* flag) {
  Camera<double> cam = make_cam(T, P);
  ProjParams<double> pp{(double)W, (double)H, near_plane, far_plane, blur, margin, thr};
  for (int64_t i = 0; i < n; ++i) {
    ProjState<double> st;
    bool in_view = project_forward(pos + i * 3, ls + i * 3, rot + i * 4, al[i], cam, pp, st);
    double* o = points7 + i * 7;
    o[0] = st.uv[0]; o[1] = st.uv[1]; o[2] = st.axis[0]; o[3] = st.axis[1];
    o[4] = st.sigma[0]; o[5] = st.sigma[1]; o[6] = st.alpha;
    depth[i] = st.pc[2];
    flag[i] = in_view;
  }
}

// float32 instantiation of the projection forward (inputs / outputs travel as double, the arithmetic is float): the
// conditioning of the eigen-pair in the PRODUCT precision (tests/test_hostmath.py::test_projection_forward_f32_covariance)
extern "C" void hm_project_fwd_f32(const double* pos, const double* ls, const double* rot, const double* al,
                                   const double* T, const double* P, int W, int H, double near_plane, double far_plane,
                                   double blur, double margin, double thr, int64_t n, double* points7, double* depth,
                                   int32_t* flag) {
  Camera<float> cam;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) cam.t[i][j] = (float)T[i * 4 + j];
  cam.fx = (float)P[0]; cam.fy = (float)P[1]; cam.cx = (float)P[2]; cam.cy = (float)P[3];
  ProjParams<float> pp{(float)W, (float)H, (float)near_plane, (float)far_plane, (float)blur, (float)margin, (float)thr};
  for (int64_t i = 0; i < n; ++i) {
    float p[3], l[3], q[4];
    for (int k = 0; k < 3; ++k) { p[k] = (float)pos[i * 3 + k]; l[k] = (float)ls[i * 3 + k]; }
    for (int k = 0; k < 4; ++k) q[k] = (float)rot[i * 4 + k];
    ProjState<float> st;
    bool in_view = project_forward(p, l, q, (float)al[i], cam, pp, st);
    double* o = points7 + i * 7;
    o[0] = st.uv[0]; o[1] = st.uv[1]; o[2] = st.axis[0]; o[3] = st.axis[1];
    o[4] = st.sigma[0]; o[5] = st.sigma[1]; o[6] = st.alpha;
    depth[i] = st.pc[2];
    flag[i] = in_view;
  }
}

extern "C" void hm_project_bwd(const double* pos, const double* ls, const double* rot, const double* al,
                               const double* T, const double* P, int W, int H, double blur, double margin,
                               int64_t n, const double* g_points7, const double* g_depth, double* d_pos,
                               double* d_ls, double* d_rot, double* d_al, double* d_cam16) {
  Camera<double> cam = make_cam(T, P);
  ProjParams<double> pp{(double)W, (double)H, 1.0, 2.0, blur, margin, 1.0 / 255.0};
  for (int k = 0; k < 16; ++k) d_cam16[k] = 0;
  for (int64_t i = 0; i < n; ++i) {
    ProjState<double> st;
    project_forward(pos + i * 3, ls + i * 3, rot + i * 4, al[i], cam, pp, st);
    project_backward(pos + i * 3, cam, st, g_points7 + i * 7, g_depth[i], d_pos + i * 3, d_ls + i * 3,
                     d_rot + i * 4, d_al[i], d_cam16);
  }
}

// float32 instantiation of the projection backward (inputs / outputs travel as double, the arithmetic is float):
// the conditioning of the hand-derived chain in the PRODUCT precision can be measured without a GPU
// (tests/test_hostmath.py::test_projection_backward_f32_conditioning).
extern "C" void hm_project_bwd_f32(const double* pos, const double* ls, const double* rot, const double* al,
                                   const double* T, const double* P, int W, int H, double blur, double margin,
                                   int64_t n, const double* g_points7, const double* g_depth, double* d_pos,
                                   double* d_ls, double* d_rot, double* d_al, double* d_cam16) {
  Camera<float> cam;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) cam.t[i][j] = (float)T[i * 4 + j];
  cam.fx = (float)P[0]; cam.fy = (float)P[1]; cam.cx = (float)P[2]; cam.cy = (float)P[3];
  ProjParams<float> pp{(float)W, (float)H, 1.0f, 2.0f, (float)blur, (float)margin, 1.0f / 255.0f};
  float cam_grad[16];
  for (int k = 0; k < 16; ++k) cam_grad[k] = 0;
  for (int64_t i = 0; i < n; ++i) {
    float p[3], l[3], q[4], gp[7], dp_[3], dl[3], dq[4], da;
    for (int k = 0; k < 3; ++k) { p[k] = (float)pos[i * 3 + k]; l[k] = (float)ls[i * 3 + k]; }
    for (int k = 0; k < 4; ++k) q[k] = (float)rot[i * 4 + k];
    for (int k = 0; k < 7; ++k) gp[k] = (float)g_points7[i * 7 + k];
    ProjState<float> st;
    project_forward(p, l, q, (float)al[i], cam, pp, st);
    project_backward(p, cam, st, gp, (float)g_depth[i], dp_, dl, dq, da, cam_grad);
    for (int k = 0; k < 3; ++k) { d_pos[i * 3 + k] = dp_[k]; d_ls[i * 3 + k] = dl[k]; }
    for (int k = 0; k < 4; ++k) d_rot[i * 4 + k] = dq[k];
    d_al[i] = da;
  }
  for (int k = 0; k < 16; ++k) d_cam16[k] = cam_grad[k];
}

template <int DEG>
static void sh_eval(const double* params, const double* positions, const int64_t* indexes, const double* cam,
                    int64_t v, int f, const double* g_out, double* out, double* g_params, double* g_pos,
                    double* g_cam) {
  constexpr int D = (DEG + 1) * (DEG + 1);
  for (int64_t i = 0; i < v; ++i) {
    const int64_t idx = indexes[i];
    const double dx = positions[idx * 3] - cam[0], dy = positions[idx * 3 + 1] - cam[1], dz = positions[idx * 3 + 2] - cam[2];
    const double len = sqrt(dx * dx + dy * dy + dz * dz);
    const double x = dx / len, y = dy / len, z = dz / len;
    double Y[D], coef[D];
    sh_basis<double, DEG>(x, y, z, Y);
    for (int d = 0; d < D; ++d) coef[d] = 0;
    const double* p = params + idx * f * D;
    for (int c = 0; c < f; ++c) {
      double acc = 0;
      for (int d = 0; d < D; ++d) acc += Y[d] * p[c * D + d];
      const double pre = acc + 0.5;
      if (out) out[i * f + c] = pre < 0 ? 0 : (pre > 1 ? 1 : pre);
      if (g_out) {
        const double g = (pre >= 0 && pre <= 1) ? g_out[i * f + c] : 0.0;
        for (int d = 0; d < D; ++d) { g_params[(idx * f + c) * D + d] += g * Y[d]; coef[d] += g * p[c * D + d]; }
      }
    }
    if (g_out) {
      double gd[3];
      sh_basis_grad_dot<double, DEG>(x, y, z, coef, gd);
      const double dot = x * gd[0] + y * gd[1] + z * gd[2];
      const double gx = (gd[0] - x * dot) / len, gy = (gd[1] - y * dot) / len, gz = (gd[2] - z * dot) / len;
      g_pos[idx * 3] += gx; g_pos[idx * 3 + 1] += gy; g_pos[idx * 3 + 2] += gz;
      g_cam[0] -= gx; g_cam[1] -= gy; g_cam[2] -= gz;
    }
  }
}

extern "C" void hm_sh(const double* params, const double* positions, const int64_t* indexes, const double* cam,
                      int64_t v, int f, int degree, const double* g_out, double* out, double* g_params,
                      double* g_pos, double* g_cam) {
  switch (degree) {
    case 0: sh_eval<0>(params, positions, indexes, cam, v, f, g_out, out, g_params, g_pos, g_cam); break;
    case 1: sh_eval<1>(params, positions, indexes, cam, v, f, g_out, out, g_params, g_pos, g_cam); break;
    case 2: sh_eval<2>(params, positions, indexes, cam, v, f, g_out, out, g_params, g_pos, g_cam); break;
    default: sh_eval<3>(params, positions, indexes, cam, v, f, g_out, out, g_params, g_pos, g_cam); break;
  }
}

// pdf + gradients at n (pixel, gaussian) pairs: g6 = mean2 axis2 sigma2
extern "C" void hm_pdf(const double* pix, const double* g6, int64_t n, int antialias, double* p, double* dmean,
                       double* daxis, double* dsigma, double* p_plain) {
  for (int64_t i = 0; i < n; ++i) {
    const double* g = g6 + i * 6;
    if (antialias) {
      p[i] = gaussian_pdf_antialias_with_grad(pix[i * 2], pix[i * 2 + 1], g, dmean + i * 2, daxis + i * 2, dsigma + i * 2);
      p_plain[i] = gaussian_pdf_antialias(pix[i * 2], pix[i * 2 + 1], g);
    } else {
      p[i] = gaussian_pdf_with_grad(pix[i * 2], pix[i * 2 + 1], g, dmean + i * 2, daxis + i * 2, dsigma + i * 2);
      p_plain[i] = gaussian_pdf(pix[i * 2], pix[i * 2 + 1], g);
    }
  }
}

// overlap counts of n gaussians (float, as on the device)
extern "C" void hm_tile_count(const float* points7, int64_t n, int W, int H, int tile, float thr, int32_t* counts,
                              int32_t* spans /*n x 4: min_x min_y span_x span_y*/) {
  for (int64_t i = 0; i < n; ++i) {
    ObbQuery q = obb_grid_query(points7 + i * 7, W, H, tile, thr);
    int c = 0;
    for (int tv = 0; tv < q.span_y; ++tv)
      for (int tu = 0; tu < q.span_x; ++tu)
        if (obb_test_tile(q, tu, tv, tile)) ++c;
    counts[i] = c;
    spans[i * 4 + 0] = q.min_tile_x; spans[i * 4 + 1] = q.min_tile_y;
    spans[i * 4 + 2] = q.span_x; spans[i * 4 + 3] = q.span_y;
  }
}
