// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on access patterns with KNOWN byte counts (VERDICT round 3, item 6;
// MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads on gfx950).  Each kernel runs twice under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// (tools/pmc_frame.sh); tools/pmc_frame_report.py divides the counters by the bytes printed here.
//   stream128 / stream32 : N x 16 B / 4 B coalesced reads of a 1.0 GB buffer (read once, larger than every cache)
//   gather28             : the mapper's and rasterizer's pattern — rows of 28 B (7 floats, 28 B stride) fetched through a
//                          random permutation (4 B index, coalesced): 28 B x rows useful, but a row touches 1 or 2
//                          64 B sectors / 128 B lines
//   gather32             : the same rows padded to 32 B (aligned: never straddles a 64 B sector)
//   write128             : N x 16 B coalesced stores
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
typedef float vec4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) calib_stream128(const vec4* __restrict__ a, size_t n, float* out) {
  vec4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) *out = 1.0f;
}
__global__ void __launch_bounds__(256) calib_stream32(const float* __restrict__ a, size_t n, float* out) {
  float acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i];
  if (acc == 123.456f) *out = 1.0f;
}
template <int STRIDE>      // floats per row: 7 (28 B rows) or 8 (32 B aligned rows); 7 floats are read either way
__global__ void __launch_bounds__(256) calib_gather(const float* __restrict__ rows, const int32_t* __restrict__ order,
                                                    size_t n, float* out) {
  float acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float* r = rows + (size_t)order[i] * STRIDE;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += r[k];
  }
  if (acc == 123.456f) *out = 1.0f;
}
__global__ void __launch_bounds__(256) calib_write128(vec4* __restrict__ a, size_t n) {
  const vec4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = v;
}

int main() {
  const size_t bytes = 1000ull << 20;
  const size_t rows = 6000000;
  float *a, *r7, *r8, *o;
  int32_t* order;
  hipMalloc(&a, bytes); hipMalloc(&r7, rows * 28); hipMalloc(&r8, rows * 32); hipMalloc(&o, 4); hipMalloc(&order, rows * 4);
  hipMemset(a, 0, bytes); hipMemset(r7, 0, rows * 28); hipMemset(r8, 0, rows * 32);
  std::vector<int32_t> perm(rows);
  std::iota(perm.begin(), perm.end(), 0);
  std::mt19937 rng(0);
  std::shuffle(perm.begin(), perm.end(), rng);
  hipMemcpy(order, perm.data(), rows * 4, hipMemcpyHostToDevice);
  // lines a 28 B row at offset 28 i touches
  size_t sect64 = 0, line128 = 0;
  for (size_t i = 0; i < rows; ++i) {
    const size_t b = 28 * i, e = b + 27;
    sect64 += e / 64 - b / 64 + 1;
    line128 += e / 128 - b / 128 + 1;
  }
  const dim3 g(8192), t(256);
  for (int rep = 0; rep < 3; ++rep) {
    calib_stream128<<<g, t>>>((const vec4*)a, bytes / 16, o);
    calib_stream32<<<g, t>>>(a, bytes / 4, o);
    calib_gather<7><<<g, t>>>(r7, order, rows, o);
    calib_gather<8><<<g, t>>>(r8, order, rows, o);
    calib_write128<<<g, t>>>((vec4*)a, bytes / 16);
  }
  hipDeviceSynchronize();
  printf("{\"calib_stream128\": {\"read_bytes\": %zu}, \"calib_stream32\": {\"read_bytes\": %zu}, "
         "\"calib_gather<7>\": {\"read_bytes\": %zu, \"index_bytes\": %zu, \"row_bytes_as_64B_sectors\": %zu, \"row_bytes_as_128B_lines\": %zu}, "
         "\"calib_gather<8>\": {\"read_bytes\": %zu, \"index_bytes\": %zu, \"row_bytes_as_64B_sectors\": %zu, \"row_bytes_as_128B_lines\": %zu}, "
         "\"calib_write128\": {\"write_bytes\": %zu}}\n",
         bytes, bytes, rows * 28 + rows * 4, rows * 4, sect64 * 64, line128 * 128,
         rows * 28 + rows * 4, rows * 4, rows * 64, rows * 128, bytes);
  return 0;
}
