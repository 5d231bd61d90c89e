// sh.hip — view-dependent colour from real spherical harmonics (degree 0..3), forward + backward.
//
// One thread per visible gaussian; params rows are (F, D) contiguous (D = (deg+1)^2), i.e. 192 B
// per gaussian for RGB degree 3: a pure HBM stream, gathered through the int64 index list.
#include <stdlib.h>

#include "common.h"
#include "frame_internal.h"

#ifndef MS_SH_ROWS_BLOCKS
#define MS_SH_ROWS_BLOCKS 16384
#endif

namespace ms {

constexpr int SH_MAX_F = 4;

template <typename T, int DEG>
__global__ void __launch_bounds__(256)
sh_fwd_kernel(const T* __restrict__ params, const T* __restrict__ positions,
              const int64_t* __restrict__ indexes, const T* __restrict__ cam_pos, int64_t v, int f,
              T* __restrict__ out, const T* __restrict__ cull_depth = nullptr, float* __restrict__ splat_rows = nullptr) {
  constexpr int D = (DEG + 1) * (DEG + 1);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v) return;
  // frame executor, float32 RGB: the colour also goes into the gaussian's splat row (raster_common.h)
  float* row = nullptr;
  if constexpr (sizeof(T) == 4) { if (splat_rows) row = splat_rows + i * SPLAT_ROW + SPLAT_ROW_COLOUR; }
  // frame executor: indexes == NULL is the identity list over ALL gaussians; culled ones (depth <= 0) get zeros
  // without touching their parameter row
  if (cull_depth && !(cull_depth[i] > T(0))) {
    for (int c = 0; c < f; ++c) out[i * f + c] = T(0);
    if (row) *reinterpret_cast<float4*>(row) = float4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  const int64_t idx = indexes ? indexes[i] : i;

  const T dx = positions[idx * 3 + 0] - cam_pos[0];
  const T dy = positions[idx * 3 + 1] - cam_pos[1];
  const T dz = positions[idx * 3 + 2] - cam_pos[2];
  const T len = t_sqrt(dx * dx + dy * dy + dz * dz);
  T Y[D];
  sh_basis<T, DEG>(dx / len, dy / len, dz / len, Y);

  const T* p = params + idx * (int64_t)f * D;
  for (int c = 0; c < f; ++c) {
    T acc = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) acc += Y[d] * p[c * D + d];
    const T colour = t_clamp(acc + T(0.5), T(0), T(1));
    out[i * f + c] = colour;
    if (row) row[c] = (float)colour;
  }
  if (row) row[3] = 0.0f;
}

// Frame executor, float32 RGB degree 3 (the headline configuration): the parameter rows of a wave's 64 consecutive
// gaussians are ONE contiguous 12 KB block, read as twelve fully coalesced 128-bit loads per lane (1 KB per wave
// instruction) instead of 64 lanes each walking its own 192-byte row — that walk lives off L1 / L2 hits between the
// lane's four visits of a 64-byte line and streamed 4.6-5.0 TB/s where the chip reads 6.2-7.1 (tools/ubench_stream.hip).
// A lane then holds piece q = 64 r + lane of the block = coefficients 4k..4k+3 of channel c of gaussian j (q = 12 j + k):
// it takes the four basis values of gaussian j from LDS (written by lane j a moment ago), forms a partial dot
// product, and the four pieces of a (gaussian, channel) — an aligned quad of lanes — are summed with two quad_perm DPP
// adds; the quad's first lane clamps and stores (16 consecutive floats per store instruction).  Rows of culled
// gaussians are not read.  The summation order differs from sh_fwd_kernel's (4 x 4 instead of 16 in a row): colours
// agree to rounding, which is what both are held to against the oracle.
template <bool ROWS>      // ROWS: the colours also go into the gaussians' splat rows (a template so that the default
__global__ void __launch_bounds__(256)      // instantiation keeps its 158 VGPRs = three waves per SIMD; with the row path: 173)
sh_fwd_rows_deg3_kernel(const float* __restrict__ params, const float* __restrict__ positions,
                        const float* __restrict__ cam_pos, const float* __restrict__ cull_depth, int64_t n,
                        float* __restrict__ out, float* __restrict__ splat_rows) {
  constexpr int D = 16, PIECES = 12, YS = 20;      // YS: 16-byte aligned rows, 4 lanes apart never on the same banks
  typedef float vec4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float s_Y[4][64 * YS];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const float cx = cam_pos[0], cy = cam_pos[1], cz = cam_pos[2];
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64; base < n; base += (int64_t)gridDim.x * 256) {
    const int count = (n - base) < 64 ? (int)(n - base) : 64;
    const int64_t i = base + lane;
    const bool vis = lane < count && (!cull_depth || cull_depth[i] > 0.0f);
    const unsigned long long visible = __ballot(vis);
    // the block's pieces first: twelve independent loads in flight while the basis is evaluated
    const vec4* src = reinterpret_cast<const vec4*>(params + base * (3 * D));
    vec4 piece[PIECES];
#pragma unroll
    for (int r = 0; r < PIECES; ++r) {
      const int q = r * 64 + lane, j = q / PIECES;
      const bool on = j < count && ((visible >> j) & 1ull);
      piece[r] = on ? __builtin_nontemporal_load(src + q) : vec4{0.f, 0.f, 0.f, 0.f};
    }
    if (vis) {
      const float dx = positions[i * 3 + 0] - cx, dy = positions[i * 3 + 1] - cy, dz = positions[i * 3 + 2] - cz;
      const float len = t_sqrt(dx * dx + dy * dy + dz * dz);
      float Y[D];
      sh_basis<float, 3>(dx / len, dy / len, dz / len, Y);
#pragma unroll
      for (int d = 0; d < D; d += 4)
        *reinterpret_cast<vec4*>(&s_Y[wave][lane * YS + d]) = vec4{Y[d], Y[d + 1], Y[d + 2], Y[d + 3]};
    }
    // LDS traffic stays inside the wave; the fences keep the compiler from forwarding per-thread values
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* dst = out + base * 3;
#pragma unroll
    for (int r = 0; r < PIECES; ++r) {
      const int q = r * 64 + lane, j = q / PIECES, k = q - j * PIECES;
      const bool on = j < count && ((visible >> j) & 1ull);
      const int d0 = (k & 3) * 4;                       // k = 4 c + (piece of the channel)
      const vec4 y = on ? *reinterpret_cast<const vec4*>(&s_Y[wave][j * YS + d0]) : vec4{0.f, 0.f, 0.f, 0.f};
      float acc = piece[r].x * y.x + piece[r].y * y.y + piece[r].z * y.z + piece[r].w * y.w;
      acc += dpp_f32<0xB1>(0.f, acc);                   // quad_perm:[1,0,3,2]
      acc += dpp_f32<0x4E>(0.f, acc);                   // quad_perm:[2,3,0,1]
      // q / 4 = 3 j + c: the quad's first lane writes colour c of gaussian j (0 for a culled one, sh_fwd_kernel's)
      if ((lane & 3) == 0 && j < count) {
        const float colour = on ? t_clamp(acc + 0.5f, 0.0f, 1.0f) : 0.0f;
        dst[q >> 2] = colour;
        // splat rows: the colour goes through the four spare words of the gaussian's LDS row (k >> 2 = channel) so that
        // its lane stores ONE 16-byte piece behind the barrier below (three scattered dwords per row cost the frame more
        // than the raster kernels gain)
        if constexpr (ROWS) s_Y[wave][j * YS + D + (k >> 2)] = colour;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (ROWS && lane < count) {
      vec4 c = *reinterpret_cast<const vec4*>(&s_Y[wave][lane * YS + D]);
      c.w = 0.0f;
      *reinterpret_cast<vec4*>(splat_rows + i * SPLAT_ROW + SPLAT_ROW_COLOUR) = c;
    }
  }
}

template <typename T, int DEG>
__global__ void __launch_bounds__(256)
sh_bwd_kernel(const T* __restrict__ params, const T* __restrict__ positions,
              const int64_t* __restrict__ indexes, const T* __restrict__ cam_pos, int64_t v, int f,
              const T* __restrict__ g_out, T* __restrict__ g_params, T* __restrict__ g_positions,
              T* __restrict__ g_cam) {
  constexpr int D = (DEG + 1) * (DEG + 1);
  T dcam[3] = {T(0), T(0), T(0)};
  __shared__ T s_cam[4 * 3];
  // grid-stride (the launch is capped when the camera gradient is wanted: one atomic per value per block)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < v; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = indexes[i];
    const T dx = positions[idx * 3 + 0] - cam_pos[0];
    const T dy = positions[idx * 3 + 1] - cam_pos[1];
    const T dz = positions[idx * 3 + 2] - cam_pos[2];
    const T len = t_sqrt(dx * dx + dy * dy + dz * dz);
    const T x = dx / len, y = dy / len, z = dz / len;
    T Y[D];
    sh_basis<T, DEG>(x, y, z, Y);

    const T* p = params + idx * (int64_t)f * D;
    T coef[D];   // sum_c g_c m_c P[c, d]: the weights of grad Y_d in d_dir
#pragma unroll
    for (int d = 0; d < D; ++d) coef[d] = T(0);

    for (int c = 0; c < f; ++c) {
      T acc = T(0);
#pragma unroll
      for (int d = 0; d < D; ++d) acc += Y[d] * p[c * D + d];
      const T pre = acc + T(0.5);
      // the clamp passes the gradient strictly inside (0, 1): the convention of Taichi's min / max autodiff
      // (indexed_spherical_harmonics.py:133,158) and the only one sh_bwd_params_kernel can apply, which sees the
      // clamped output; both kernels must agree or d(params) and d(direction) of one point would disagree
      const T g = (pre > T(0) && pre < T(1)) ? g_out[i * f + c] : T(0);
      if (g_params) {
        T* gp = g_params + (idx * (int64_t)f + c) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) atomic_add_noret(gp + d, g * Y[d]);
      }
#pragma unroll
      for (int d = 0; d < D; ++d) coef[d] += g * p[c * D + d];
    }

    if (g_positions || g_cam) {
      T gd[3];
      sh_basis_grad_dot<T, DEG>(x, y, z, coef, gd);
      // dir = d / |d|: dp = (I - dir dir^T) gd / |d|
      const T dot = x * gd[0] + y * gd[1] + z * gd[2];
      const T gx = (gd[0] - x * dot) / len, gy = (gd[1] - y * dot) / len, gz = (gd[2] - z * dot) / len;
      if (g_positions) {
        atomic_add_noret(g_positions + idx * 3 + 0, gx);
        atomic_add_noret(g_positions + idx * 3 + 1, gy);
        atomic_add_noret(g_positions + idx * 3 + 2, gz);
      }
      dcam[0] -= gx; dcam[1] -= gy; dcam[2] -= gz;
    }
  }
  if (g_cam) block_sum_commit<T, 3>(dcam, g_cam, s_cam);
}


// Fast parameter-gradient path: d params[idx, c, d] = g_out[i, c] * [0 < out[i, c] < 1] * Y_d(dir).
// Phase 1 (lane = gaussian) evaluates the basis and the masked colour gradient into LDS; phase 2
// (lanes = the F*D coefficients of ONE gaussian row) streams the rows out with full-line stores,
// so the 4*F*D bytes per gaussian (192 B for RGB degree 3) leave the CU coalesced instead of as
// 48 scattered 4-byte atomics per thread.  UNIQUE (indexes come from the projection compaction, no
// repeats) uses plain stores; otherwise the same rows are accumulated with atomics.
template <typename T, int DEG, bool UNIQUE>
__global__ void __launch_bounds__(256)
sh_bwd_params_kernel(const T* __restrict__ positions, const int64_t* __restrict__ indexes,
                     const T* __restrict__ cam_pos, int64_t v, int f, const T* __restrict__ out,
                     const T* __restrict__ g_out, T* __restrict__ g_params) {
  constexpr int D = (DEG + 1) * (DEG + 1);
  constexpr int YS = D + 1;                 // padded row stride: conflict-free phase-1 writes
  __shared__ T s_Y[4][64 * YS];
  __shared__ T s_g[4][64 * SH_MAX_F];
  __shared__ int64_t s_idx[4][64];

  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64;
  if (base >= v) return;
  const int64_t i = base + lane;
  const int count = (v - base) < 64 ? (int)(v - base) : 64;

  if (lane < count) {
    const int64_t idx = indexes[i];
    const T dx = positions[idx * 3 + 0] - cam_pos[0];
    const T dy = positions[idx * 3 + 1] - cam_pos[1];
    const T dz = positions[idx * 3 + 2] - cam_pos[2];
    const T len = t_sqrt(dx * dx + dy * dy + dz * dz);
    T Y[D];
    sh_basis<T, DEG>(dx / len, dy / len, dz / len, Y);
#pragma unroll
    for (int d = 0; d < D; ++d) s_Y[wave][lane * YS + d] = Y[d];
    for (int c = 0; c < f; ++c) {
      const T o = out[i * f + c];
      s_g[wave][lane * SH_MAX_F + c] = (o > T(0) && o < T(1)) ? g_out[i * f + c] : T(0);
    }
    s_idx[wave][lane] = idx;
  }
  __builtin_amdgcn_wave_barrier();   // LDS traffic stays inside the wave: no block barrier needed

  const int row = f * D;
  if (UNIQUE && D % 4 == 0 && f == 3 && sizeof(T) == 4 && (reinterpret_cast<uintptr_t>(g_params) & 15) == 0) {
    // RGB, degree 1 / 3, float: the wave's count x 3 x D gradient values leave as 128-bit stores, four consecutive
    // coefficients of one (gaussian, channel) per lane — 1 KB per store instruction when the rows are adjacent
    // (every gaussian visible), 16-byte pieces of the right rows otherwise.  The row-at-a-time loop below issues
    // one 192-byte store instruction per gaussian.
    constexpr int PIECES = 3 * D / 4;                       // 128-bit pieces per gaussian row
    for (int q = lane; q < count * PIECES; q += 64) {
      const int j = q / PIECES, k = q - j * PIECES;
      const int c = (4 * k) / D, d0 = 4 * k - c * D;
      const T g = s_g[wave][j * SH_MAX_F + c];
      const T* y = &s_Y[wave][j * YS + d0];
      float4 val = make_float4((float)(g * y[0]), (float)(g * y[1]), (float)(g * y[2]), (float)(g * y[3]));
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(g_params) + s_idx[wave][j] * row + 4 * k) = val;
    }
    return;
  }
  for (int j = 0; j < count; ++j) {
    T* dst = g_params + s_idx[wave][j] * row;
    for (int e = lane; e < row; e += 64) {
      const T val = s_g[wave][j * SH_MAX_F + e / D] * s_Y[wave][j * YS + e % D];
      if (UNIQUE) dst[e] = val;
      else atomic_add_noret(dst + e, val);
    }
  }
}

template <typename T>
static int launch_sh_fwd(const void* params, const void* positions, const int64_t* indexes,
                         const void* cam, int64_t v, int f, int degree, void* out, hipStream_t s) {
  const dim3 block(256), grid((unsigned)div_up(v, 256));
#define MS_SH_FWD(DEG) \
  sh_fwd_kernel<T, DEG><<<grid, block, 0, s>>>((const T*)params, (const T*)positions, indexes, (const T*)cam, v, f, (T*)out)
  switch (degree) {
    case 0: MS_SH_FWD(0); break;
    case 1: MS_SH_FWD(1); break;
    case 2: MS_SH_FWD(2); break;
    default: MS_SH_FWD(3); break;
  }
#undef MS_SH_FWD
  return 0;
}

template <typename T>
static int launch_sh_bwd(const void* params, const void* positions, const int64_t* indexes,
                         const void* cam, int64_t v, int f, int degree, const void* out, const void* g_out,
                         void* g_params, void* g_positions, void* g_cam, int unique, hipStream_t s) {
  const dim3 block(256), grid((unsigned)div_up(v, 256));
  const bool fast_params = out && g_params && f <= SH_MAX_F;
  if (fast_params) {
#define MS_SH_BWD_P(DEG)                                                                                  \
  do {                                                                                                    \
    if (unique) sh_bwd_params_kernel<T, DEG, true><<<grid, block, 0, s>>>((const T*)positions, indexes,   \
        (const T*)cam, v, f, (const T*)out, (const T*)g_out, (T*)g_params);                               \
    else sh_bwd_params_kernel<T, DEG, false><<<grid, block, 0, s>>>((const T*)positions, indexes,         \
        (const T*)cam, v, f, (const T*)out, (const T*)g_out, (T*)g_params);                               \
  } while (0)
    switch (degree) {
      case 0: MS_SH_BWD_P(0); break;
      case 1: MS_SH_BWD_P(1); break;
      case 2: MS_SH_BWD_P(2); break;
      default: MS_SH_BWD_P(3); break;
    }
#undef MS_SH_BWD_P
    if (!g_positions && !g_cam) return 0;
    g_params = nullptr;        // the direction gradients below: a second, atomics-free-for-params pass
  }
  const dim3 grid_dir((unsigned)(g_cam ? (div_up(v, 256) < 2048 ? div_up(v, 256) : 2048) : div_up(v, 256)));
#define MS_SH_BWD(DEG)                                                                              \
  sh_bwd_kernel<T, DEG><<<grid_dir, block, 0, s>>>((const T*)params, (const T*)positions, indexes,      \
                                                (const T*)cam, v, f, (const T*)g_out, (T*)g_params, \
                                                (T*)g_positions, (T*)g_cam)
  switch (degree) {
    case 0: MS_SH_BWD(0); break;
    case 1: MS_SH_BWD(1); break;
    case 2: MS_SH_BWD(2); break;
    default: MS_SH_BWD(3); break;
  }
#undef MS_SH_BWD
  return 0;
}

template <typename T>
static void launch_sh_fwd_inplace(const void* params, const void* positions, const void* depth, const void* cam,
                                  int64_t n, int f, int degree, void* out, hipStream_t s, float* splat_rows) {
  const dim3 block(256), grid((unsigned)div_up(n, 256));
#define MS_SH_FWD(DEG) \
  sh_fwd_kernel<T, DEG><<<grid, block, 0, s>>>((const T*)params, (const T*)positions, nullptr, (const T*)cam, n, f, (T*)out, (const T*)depth, splat_rows)
  switch (degree) {
    case 0: MS_SH_FWD(0); break;
    case 1: MS_SH_FWD(1); break;
    case 2: MS_SH_FWD(2); break;
    default: MS_SH_FWD(3); break;
  }
#undef MS_SH_FWD
}

int sh_fwd_inplace_launch(const void* params, const void* positions, const void* depth, const void* cam_pos,
                          int64_t n, int f, int degree, void* out, int dtype, hipStream_t s, float* splat_rows) {
  if (n == 0) return 0;
  if (splat_rows && !(dtype == MS_F32 && f == 3)) { set_error("sh_fwd_inplace_launch: splat rows are float32 RGB"); return MS_ERR_BAD_ARG; }
  static const bool rows_off = [] { const char* e = getenv("MS_SH_FWD"); return e && e[0] == 'w'; }();   // "walk": the per-lane row walk
  if (dtype == MS_F32 && f == 3 && degree == 3 && !rows_off && (reinterpret_cast<uintptr_t>(params) & 15) == 0) {
    int64_t blocks = div_up(n, 256);
    if (blocks > MS_SH_ROWS_BLOCKS) blocks = MS_SH_ROWS_BLOCKS;       // grid-stride: a resident grid streams best
    if (splat_rows)
      sh_fwd_rows_deg3_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const float*)params, (const float*)positions,
                                                                                (const float*)cam_pos, (const float*)depth, n,
                                                                                (float*)out, splat_rows);
    else
      sh_fwd_rows_deg3_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const float*)params, (const float*)positions,
                                                                                 (const float*)cam_pos, (const float*)depth, n,
                                                                                 (float*)out, nullptr);
    return 0;
  }
  if (dtype == MS_F32) launch_sh_fwd_inplace<float>(params, positions, depth, cam_pos, n, f, degree, out, s, splat_rows);
  else launch_sh_fwd_inplace<double>(params, positions, depth, cam_pos, n, f, degree, out, s, nullptr);
  return 0;
}

}  // namespace ms

using namespace ms;

extern "C" int ms_sh_fwd(const void* params, const void* positions, const int64_t* indexes,
                         const void* camera_pos, int64_t v, int f, int degree, void* out, int dtype,
                         void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(degree >= 0 && degree <= 3, "degree must be in [0, 3]");
  MS_CHECK_ARG(f >= 1, "f < 1");
  MS_CHECK_ARG(dtype == MS_F32 || dtype == MS_F64, "dtype must be MS_F32 or MS_F64");
  if (v == 0) return 0;
  MS_CHECK_ARG(params && positions && indexes && camera_pos && out, "null pointer");
  if (dtype == MS_F32) launch_sh_fwd<float>(params, positions, indexes, camera_pos, v, f, degree, out, (hipStream_t)stream);
  else launch_sh_fwd<double>(params, positions, indexes, camera_pos, v, f, degree, out, (hipStream_t)stream);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_sh_bwd(const void* params, const void* positions, const int64_t* indexes,
                         const void* camera_pos, int64_t v, int f, int degree, const void* out,
                         const void* grad_out, void* grad_params, void* grad_positions,
                         void* grad_camera_pos, int unique_indexes, int dtype, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(degree >= 0 && degree <= 3, "degree must be in [0, 3]");
  MS_CHECK_ARG(f >= 1, "f < 1");
  MS_CHECK_ARG(dtype == MS_F32 || dtype == MS_F64, "dtype must be MS_F32 or MS_F64");
  if (v == 0) return 0;
  MS_CHECK_ARG(params && positions && indexes && camera_pos && grad_out, "null pointer");
  if (dtype == MS_F32) launch_sh_bwd<float>(params, positions, indexes, camera_pos, v, f, degree, out, grad_out, grad_params, grad_positions, grad_camera_pos, unique_indexes, (hipStream_t)stream);
  else launch_sh_bwd<double>(params, positions, indexes, camera_pos, v, f, degree, out, grad_out, grad_params, grad_positions, grad_camera_pos, unique_indexes, (hipStream_t)stream);
  MS_CHECK_LAUNCH();
  return 0;
}
