// projection.hip — perspective projection of 3D gaussians (forward, compaction, backward).
//
// One thread per gaussian, pure HBM streaming: 44 B in / 36 B out per gaussian forward.  The
// camera (20 scalars) is read through the scalar cache by every thread instead of being
// replicated per point as the reference does (perspective/projection.py:215-216, 64 B/point).
// culling decisions (in_view) are float comparisons shared with the oracle: no FMA contraction
#pragma clang fp contract(off)
#include "common.h"
#include "frame_internal.h"

namespace ms {

template <typename T>
__device__ __forceinline__ void load_camera(const T* __restrict__ Tcw, const T* __restrict__ proj,
                                            Camera<T>& cam) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) cam.t[i][j] = Tcw[i * 4 + j];
  cam.fx = proj[0]; cam.fy = proj[1]; cam.cx = proj[2]; cam.cy = proj[3];
}

template <typename T>
__global__ void __launch_bounds__(256)
project_fwd_kernel(const T* __restrict__ position, const T* __restrict__ log_scaling,
                   const T* __restrict__ rotation, const T* __restrict__ alpha_logit,
                   const T* __restrict__ Tcw, const T* __restrict__ proj, ProjParams<T> pp,
                   int64_t n, T* __restrict__ out_points, T* __restrict__ out_depth,
                   int32_t* __restrict__ out_flag, float* __restrict__ splat_rows = nullptr,
                   const T* __restrict__ colours_in = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;

  Camera<T> cam;
  load_camera(Tcw, proj, cam);

  const T p[3] = {position[i * 3 + 0], position[i * 3 + 1], position[i * 3 + 2]};
  const T ls[3] = {log_scaling[i * 3 + 0], log_scaling[i * 3 + 1], log_scaling[i * 3 + 2]};
  const T q[4] = {rotation[i * 4 + 0], rotation[i * 4 + 1], rotation[i * 4 + 2], rotation[i * 4 + 3]};

  ProjState<T> st;
  const bool in_view = project_forward(p, ls, q, alpha_logit[i], cam, pp, st);

  T* o = out_points + i * 7;
  o[0] = st.uv[0]; o[1] = st.uv[1];
  o[2] = st.axis[0]; o[3] = st.axis[1];
  o[4] = st.sigma[0]; o[5] = st.sigma[1];
  o[6] = st.alpha;
  out_depth[i] = in_view ? st.pc[2] : T(0);
  if (out_flag) out_flag[i] = in_view ? 1 : 0;
  // frame executor (float32 RGB): the gaussian's splat row (common.h) — the same seven values and the depth; the
  // colour words are the SH kernel's, or copied here when the features ARE the colours (colours_in)
  if constexpr (sizeof(T) == 4) {
    if (splat_rows) {
      float4* row = reinterpret_cast<float4*>(splat_rows + i * SPLAT_ROW);
      row[0] = float4{st.uv[0], st.uv[1], st.axis[0], st.axis[1]};
      row[1] = float4{st.sigma[0], st.sigma[1], st.alpha, in_view ? st.pc[2] : 0.0f};
      if (colours_in) row[2] = float4{colours_in[i * 3 + 0], colours_in[i * 3 + 1], colours_in[i * 3 + 2], 0.0f};
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
project_gather_kernel(const T* __restrict__ points, const T* __restrict__ depth,
                      const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, int64_t n,
                      T near_plane, T far_plane, T* __restrict__ out_points, T* __restrict__ out_depth,
                      T* __restrict__ out_ndc, int64_t* __restrict__ out_indexes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!flag[i]) return;
  const int64_t o = scan[i];
#pragma unroll
  for (int k = 0; k < 7; ++k) out_points[o * 7 + k] = points[i * 7 + k];
  const T d = depth[i];
  out_depth[o] = d;
  if (out_ndc) out_ndc[o] = T(1) - (T(1) / d - T(1) / far_plane) / (T(1) / near_plane - T(1) / far_plane);
  out_indexes[o] = i;
}

template <typename T>
__global__ void __launch_bounds__(256)
project_bwd_kernel(const T* __restrict__ position, const T* __restrict__ log_scaling,
                   const T* __restrict__ rotation, const T* __restrict__ alpha_logit,
                   const T* __restrict__ Tcw, const T* __restrict__ proj, ProjParams<T> pp,
                   const int64_t* __restrict__ indexes, int64_t v,
                   const T* __restrict__ g_points, const T* __restrict__ g_depth,
                   T* __restrict__ d_position, T* __restrict__ d_log_scaling,
                   T* __restrict__ d_rotation, T* __restrict__ d_alpha_logit,
                   T* __restrict__ d_camera) {
  T cam_grad[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) cam_grad[k] = T(0);
  __shared__ T s_cam[4 * 16];

  // grid-stride: with camera gradients the launch is capped at a few thousand blocks so that the 16
  // whole-launch sums cost one atomic per value per BLOCK
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < v; i += (int64_t)gridDim.x * blockDim.x) {
    Camera<T> cam;
    load_camera(Tcw, proj, cam);
    const int64_t idx = indexes[i];
    const T p[3] = {position[idx * 3 + 0], position[idx * 3 + 1], position[idx * 3 + 2]};
    const T ls[3] = {log_scaling[idx * 3 + 0], log_scaling[idx * 3 + 1], log_scaling[idx * 3 + 2]};
    const T q[4] = {rotation[idx * 4 + 0], rotation[idx * 4 + 1], rotation[idx * 4 + 2], rotation[idx * 4 + 3]};

    ProjState<T> st;
    project_forward(p, ls, q, alpha_logit[idx], cam, pp, st);

    T gp[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) gp[k] = g_points[i * 7 + k];

    T dp[3], dls[3], dq[4], dal;
    project_backward(p, cam, st, gp, g_depth ? g_depth[i] : T(0), dp, dls, dq, dal, cam_grad);

#pragma unroll
    for (int k = 0; k < 3; ++k) d_position[idx * 3 + k] = dp[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) d_log_scaling[idx * 3 + k] = dls[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) d_rotation[idx * 4 + k] = dq[k];
    d_alpha_logit[idx] = dal;
  }

  if (d_camera) block_sum_commit<T, 16>(cam_grad, d_camera, s_cam);
}

}  // namespace ms

using namespace ms;

template <typename T>
static ProjParams<T> make_params(int w, int h, double near_plane, double far_plane, double blur,
                                 double clamp_margin, double alpha_threshold) {
  ProjParams<T> pp;
  pp.width = (T)w; pp.height = (T)h;
  pp.near_plane = (T)near_plane; pp.far_plane = (T)far_plane;
  pp.blur_cov = (T)blur; pp.clamp_margin = (T)clamp_margin; pp.alpha_threshold = (T)alpha_threshold;
  return pp;
}

extern "C" int ms_project_fwd(const void* position, const void* log_scaling, const void* rotation,
                              const void* alpha_logit, const void* T_camera_world,
                              const void* projection, int image_w, int image_h, double near_plane,
                              double far_plane, double blur_cov, double clamp_margin,
                              double alpha_threshold, int64_t n, void* out_points7, void* out_depth,
                              int32_t* out_flag, int dtype, void* stream) {
  return project_fwd_launch(position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_w, image_h,
                            near_plane, far_plane, blur_cov, clamp_margin, alpha_threshold, n, out_points7, out_depth,
                            out_flag, dtype, stream, nullptr, nullptr);
}

// ms_project_fwd; splat_rows (float32 only): also fills words 0..7 of every gaussian's splat row, and words 8..11 from
// colours_in (n x 3) when given
int ms::project_fwd_launch(const void* position, const void* log_scaling, const void* rotation,
                           const void* alpha_logit, const void* T_camera_world,
                           const void* projection, int image_w, int image_h, double near_plane,
                           double far_plane, double blur_cov, double clamp_margin,
                           double alpha_threshold, int64_t n, void* out_points7, void* out_depth,
                           int32_t* out_flag, int dtype, void* stream, float* splat_rows, const void* colours_in) {
  MS_CHECK_ARG(n >= 0, "n < 0");
  MS_CHECK_ARG(dtype == MS_F32 || dtype == MS_F64, "dtype must be MS_F32 or MS_F64");
  MS_CHECK_ARG(!splat_rows || dtype == MS_F32, "splat rows are float32");
  if (n == 0) return 0;
  MS_CHECK_ARG(position && log_scaling && rotation && alpha_logit && T_camera_world && projection,
               "null input");
  MS_CHECK_ARG(out_points7 && out_depth, "null output");   // out_flag may be NULL (depth > 0 <=> in view)
  const dim3 block(256), grid((unsigned)div_up(n, 256));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MS_F32) {
    project_fwd_kernel<float><<<grid, block, 0, s>>>(
        (const float*)position, (const float*)log_scaling, (const float*)rotation,
        (const float*)alpha_logit, (const float*)T_camera_world, (const float*)projection,
        make_params<float>(image_w, image_h, near_plane, far_plane, blur_cov, clamp_margin, alpha_threshold),
        n, (float*)out_points7, (float*)out_depth, out_flag, splat_rows, (const float*)colours_in);
  } else {
    project_fwd_kernel<double><<<grid, block, 0, s>>>(
        (const double*)position, (const double*)log_scaling, (const double*)rotation,
        (const double*)alpha_logit, (const double*)T_camera_world, (const double*)projection,
        make_params<double>(image_w, image_h, near_plane, far_plane, blur_cov, clamp_margin, alpha_threshold),
        n, (double*)out_points7, (double*)out_depth, out_flag);
  }
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_project_gather(const void* points7, const void* depth, const int32_t* flag,
                                 const int32_t* scan, int64_t n, double near_plane, double far_plane,
                                 void* out_points7, void* out_depth, void* out_ndc_depth,
                                 int64_t* out_indexes, int dtype, void* stream) {
  MS_CHECK_ARG(n >= 0, "n < 0");
  MS_CHECK_ARG(dtype == MS_F32 || dtype == MS_F64, "dtype must be MS_F32 or MS_F64");
  if (n == 0) return 0;
  MS_CHECK_ARG(points7 && depth && flag && scan, "null input");
  MS_CHECK_ARG(out_points7 && out_depth && out_indexes, "null output");
  const dim3 block(256), grid((unsigned)div_up(n, 256));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MS_F32)
    project_gather_kernel<float><<<grid, block, 0, s>>>(
        (const float*)points7, (const float*)depth, flag, scan, n, (float)near_plane, (float)far_plane,
        (float*)out_points7, (float*)out_depth, (float*)out_ndc_depth, out_indexes);
  else
    project_gather_kernel<double><<<grid, block, 0, s>>>(
        (const double*)points7, (const double*)depth, flag, scan, n, near_plane, far_plane,
        (double*)out_points7, (double*)out_depth, (double*)out_ndc_depth, out_indexes);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_project_bwd(const void* position, const void* log_scaling, const void* rotation,
                              const void* alpha_logit, const void* T_camera_world,
                              const void* projection, int image_w, int image_h, double blur_cov,
                              double clamp_margin, const int64_t* indexes, int64_t v,
                              const void* grad_points7, const void* grad_depth, void* grad_position,
                              void* grad_log_scaling, void* grad_rotation, void* grad_alpha_logit,
                              void* grad_camera, int dtype, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(dtype == MS_F32 || dtype == MS_F64, "dtype must be MS_F32 or MS_F64");
  if (v == 0) return 0;
  MS_CHECK_ARG(position && log_scaling && rotation && alpha_logit && T_camera_world && projection && indexes,
               "null input");
  MS_CHECK_ARG(grad_points7 != nullptr, "null incoming gradient");   // grad_depth may be NULL (= zeros)
  MS_CHECK_ARG(grad_position && grad_log_scaling && grad_rotation && grad_alpha_logit, "null output");
  int64_t blocks = div_up(v, 256);
  if (grad_camera && blocks > 2048) blocks = 2048;       // bounded atomic count for the camera sums
  const dim3 block(256), grid((unsigned)blocks);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MS_F32)
    project_bwd_kernel<float><<<grid, block, 0, s>>>(
        (const float*)position, (const float*)log_scaling, (const float*)rotation,
        (const float*)alpha_logit, (const float*)T_camera_world, (const float*)projection,
        make_params<float>(image_w, image_h, 1.0, 2.0, blur_cov, clamp_margin, 1.0 / 255.0), indexes, v,
        (const float*)grad_points7, (const float*)grad_depth, (float*)grad_position,
        (float*)grad_log_scaling, (float*)grad_rotation, (float*)grad_alpha_logit, (float*)grad_camera);
  else
    project_bwd_kernel<double><<<grid, block, 0, s>>>(
        (const double*)position, (const double*)log_scaling, (const double*)rotation,
        (const double*)alpha_logit, (const double*)T_camera_world, (const double*)projection,
        make_params<double>(image_w, image_h, 1.0, 2.0, blur_cov, clamp_margin, 1.0 / 255.0), indexes, v,
        (const double*)grad_points7, (const double*)grad_depth, (double*)grad_position,
        (double*)grad_log_scaling, (double*)grad_rotation, (double*)grad_alpha_logit, (double*)grad_camera);
  MS_CHECK_LAUNCH();
  return 0;
}


// ---- camera position --------------------------------------------------------------------------------
// Translation column of inverse(T_camera_world) (reference perspective/params.py:62-65 computes
// torch.inverse(T)[0:3, 3]).  One thread, Gauss-Jordan with partial pivoting in double: replaces the ~10
// launch-latency-bound kernels of a device-side LU (rocSOLVER) with one 3 us launch per frame.
namespace ms {
template <typename T>
__global__ void camera_position_kernel(const T* __restrict__ m, T* __restrict__ out) {
  camera_position_solve(m, out);
}
}  // namespace ms

extern "C" int ms_camera_position(const void* t_camera_world, void* out_position3, int dtype, void* stream) {
  MS_CHECK_ARG(t_camera_world && out_position3, "null pointer");
  MS_CHECK_ARG(dtype == MS_F32 || dtype == MS_F64, "dtype must be MS_F32 or MS_F64");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MS_F32) ms::camera_position_kernel<float><<<1, 1, 0, s>>>((const float*)t_camera_world, (float*)out_position3);
  else ms::camera_position_kernel<double><<<1, 1, 0, s>>>((const double*)t_camera_world, (double*)out_position3);
  MS_CHECK_LAUNCH();
  return 0;
}
