// tile_sort.hip — per-tile depth sort of the frame executor's direct-order mapper.
//
// The reference sorts ONE array of 64 bit keys tile << 32 | depth bits (mapper/tile_mapper.py:36-42, 148-170: a full
// radix sort over K overlaps); the round-2/3 frame executor pre-sorted the gaussians by depth and sorted the overlaps by
// tile only.  Here the overlaps are emitted in storage order, grouped by tile with a stable sort on the tile bits alone
// (scan_sort.hip), and every tile's run — a few hundred to a few thousand entries that one workgroup holds in LDS — is
// put into (depth key, point index) order by this kernel.  The order that results is the reference's: equal depth
// keys keep ascending point indices, which is what a stable sort of the storage-order emission gives.
//
// One workgroup per tile.  Runs up to 256 * R entries: a bucket sort in LDS — min / max of the run's keys, a MONOTONE
// map of the key onto as many buckets as there are entries (by float value for normal positive float keys, so that
// uniformly spread depths fill the buckets evenly; by integer value otherwise: 16 bit keys, zeros, denormals, sign
// bits), LDS atomics for the slots, a scan for the bucket offsets, then every entry counts the entries of its bucket
// that precede it in (key, index) order.  Anything else — longer runs, or runs whose keys pile up in few buckets (sum
// of squared bucket sizes above TS_COST_LIMIT per entry) — goes through an LSD radix sort of the workgroup on global
// memory that skips the digits that do not vary: slower per entry but bounded, so no input makes this kernel quadratic.
#include "common.h"
#include "frame_internal.h"

namespace ms {

constexpr int TS_THREADS = 256;
constexpr int TS_WAVES = TS_THREADS / 64;
constexpr int TS_COST_LIMIT = 64;          // bucket path: at most this many comparisons per entry on average
// run-length classes: 256 * R entries held in LDS by one workgroup (tuned on the bench scene)
#ifndef TS_SMALL_R
#define TS_SMALL_R 4
#endif
#ifndef TS_MID_R
#define TS_MID_R 10                         // 0: no middle class
#endif
#ifndef TS_LONG_R
#define TS_LONG_R 20
#endif

constexpr int TS_PER_WAVE = 256;           // bounded path: entries a wave ranks per round of the workgroup
struct TsShared {
  uint32_t red[2 * TS_WAVES];
  uint32_t hist[256];
};
// a run the per-tile kernels decline (keys piled up in few buckets) is left to the long-run kernel through this word
// in the first scratch slot of the run; no key is all ones (tile ids are below 2^31)
constexpr uint64_t TS_DECLINED = ~0ull;
constexpr int TS_REPORT_ABOVE = 16384;      // runs longer than this are reported through ms_frame_inputs.longest_run_host

__device__ __forceinline__ uint32_t ts_wave_min(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o); v = w < v ? w : v; }
  return v;
}
__device__ __forceinline__ uint32_t ts_wave_max(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t w = (uint32_t)__shfl_xor((int)v, o); v = w > v ? w : v; }
  return v;
}
__device__ __forceinline__ uint32_t ts_wave_or(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o);
  return v;
}
__device__ __forceinline__ uint32_t ts_wave_and(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v &= (uint32_t)__shfl_xor((int)v, o);
  return v;
}
__device__ __forceinline__ uint32_t ts_wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
  return v;
}
// inclusive prefix sum over the wave
__device__ __forceinline__ uint32_t ts_wave_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t w = (uint32_t)__shfl_up((int)v, o); if (lane >= o) v += w; }
  return v;
}

// exclusive prefix sum of one value per thread over the workgroup; *total = the sum (uses red[0 .. TS_WAVES))
__device__ __forceinline__ uint32_t ts_block_exclusive(uint32_t v, uint32_t* red, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = ts_wave_scan(v);
  __syncthreads();
  if (lane == 63) red[wave] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < TS_WAVES; ++w) { const uint32_t c = red[w]; if (w < wave) before += c; all += c; }
  *total = all;
  return before + inc - v;
}

// ---- bounded path: LSD radix sort of one run by the whole workgroup, on global memory ------------------------------
// srt[b .. b + n): tile << 32 | depth key; o2p[b .. b + n): point indices (ascending); alt: scratch of the same extent.
// wcnt: TS_WAVES * 256 words of LDS (per-wave digit counters, then write positions)
__device__ void tile_radix_sort_global(uint64_t* srt, int32_t* o2p, uint64_t* alt,
                                       int64_t b, int n, TsShared& sh, uint32_t* wcnt) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t kor = 0u, kand = 0xffffffffu;
  for (int i = t; i < n; i += TS_THREADS) {
    const uint32_t k = (uint32_t)srt[b + i];
    srt[b + i] = ((uint64_t)k << 32) | (uint32_t)o2p[b + i];
    kor |= k;
    kand &= k;
  }
  kor = ts_wave_or(kor);
  kand = ts_wave_and(kand);
  __syncthreads();
  if (lane == 0) { sh.red[wave] = kor; sh.red[TS_WAVES + wave] = kand; }
  __syncthreads();
  kor = 0u; kand = 0xffffffffu;
#pragma unroll
  for (int w = 0; w < TS_WAVES; ++w) { kor |= sh.red[w]; kand &= sh.red[TS_WAVES + w]; }
  const uint32_t vary = kor ^ kand;                    // bits that differ somewhere in the run

  uint64_t* src = srt;
  uint64_t* dst = alt;
  for (int pass = 0; pass < 4; ++pass) {
    if (((vary >> (8 * pass)) & 0xffu) == 0u) continue;         // same digit everywhere: identity permutation
    const int shift = 32 + 8 * pass;
    __syncthreads();
    sh.hist[t] = 0u;
    __syncthreads();
    for (int i = t; i < n; i += TS_THREADS) atomicAdd(&sh.hist[(uint32_t)(src[b + i] >> shift) & 0xffu], 1u);
    __syncthreads();
    uint32_t total;
    const uint32_t base = ts_block_exclusive(sh.hist[t], sh.red, &total);
    __syncthreads();
    sh.hist[t] = base;                                           // running write position of digit t
    __syncthreads();
    // 1024 entries per round of the workgroup: every wave ranks its 256 entries (four groups of 64, in order) against
    // its OWN digit counters — no workgroup barrier inside — then one thread per digit turns the four waves' counts
    // into write positions, and the entries go out.  Stable: waves, groups and lanes are taken in entry order.
    for (int c = 0; c < n; c += TS_WAVES * TS_PER_WAVE) {
      for (int j = t; j < TS_WAVES * 256; j += TS_THREADS) wcnt[j] = 0u;
      __syncthreads();
      uint64_t p[TS_PER_WAVE / 64];
      uint32_t rank[TS_PER_WAVE / 64];
#pragma unroll
      for (int g = 0; g < TS_PER_WAVE / 64; ++g) {
        const int i = c + wave * TS_PER_WAVE + g * 64 + lane;
        const bool valid = i < n;
        p[g] = valid ? src[b + i] : 0ull;
        const uint32_t d = (uint32_t)(p[g] >> shift) & 0xffu;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
          const bool one = (d >> bit) & 1u;
          const uint64_t bal = __ballot(one);
          peers &= one ? bal : ~bal;
        }
        const uint64_t below = peers & ((1ull << lane) - 1ull);
        uint32_t prev = 0u;
        if (valid) prev = wcnt[wave * 256 + d];
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0ull) wcnt[wave * 256 + d] = prev + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[g] = prev + (uint32_t)__popcll(below);
      }
      __syncthreads();
      {
        uint32_t at = sh.hist[t];                                 // thread t = digit t
#pragma unroll
        for (int w = 0; w < TS_WAVES; ++w) { const uint32_t cw = wcnt[w * 256 + t]; wcnt[w * 256 + t] = at; at += cw; }
        sh.hist[t] = at;
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < TS_PER_WAVE / 64; ++g) {
        const int i = c + wave * TS_PER_WAVE + g * 64 + lane;
        if (i < n) dst[b + wcnt[wave * 256 + ((uint32_t)(p[g] >> shift) & 0xffu)] + rank[g]] = p[g];
      }
      __syncthreads();
    }
    uint64_t* const swap = src; src = dst; dst = swap;
  }
  __syncthreads();
  if (src == srt && vary == 0u) return;                          // nothing moved
  for (int i = t; i < n; i += TS_THREADS) o2p[b + i] = (int32_t)(uint32_t)src[b + i];
}

// ---- LDS bucket path --------------------------------------------------------------------------------------------
struct BucketMap {
  uint32_t kmin, top;       // smallest key; last bucket
  float base, scale;
  int by_value;
  __device__ __forceinline__ uint32_t operator()(uint32_t k) const {
    const float x = by_value ? __uint_as_float(k) - base : (float)(k - kmin);
    const uint32_t bkt = (uint32_t)(x * scale);
    return bkt < top ? bkt : top;
  }
};

// false: declined (nothing written) — the keys pile up in few buckets and the run belongs to the bounded path
template <int R>
__device__ __forceinline__ bool tile_bucket_sort(uint64_t* __restrict__ srt, int32_t* __restrict__ o2p,
                                                 int64_t b, int n, uint64_t* pairs, uint32_t* cnt, TsShared& sh) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t key[R], id[R];
  uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = t + TS_THREADS * r;
    key[r] = 0u; id[r] = 0u;
    if (i < n) {
      key[r] = (uint32_t)srt[b + i];
      id[r] = (uint32_t)o2p[b + i];
      kmin = key[r] < kmin ? key[r] : kmin;
      kmax = key[r] > kmax ? key[r] : kmax;
    }
  }
  kmin = ts_wave_min(kmin);
  kmax = ts_wave_max(kmax);
  if (lane == 0) { sh.red[wave] = kmin; sh.red[TS_WAVES + wave] = kmax; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < TS_WAVES; ++w) {
    kmin = sh.red[w] < kmin ? sh.red[w] : kmin;
    kmax = sh.red[TS_WAVES + w] > kmax ? sh.red[TS_WAVES + w] : kmax;
  }
  if (kmin == kmax) return true;           // one depth key: the run is in point order already

  const uint32_t nb = (uint32_t)n;
  BucketMap bucket;
  bucket.kmin = kmin;
  bucket.top = nb - 1u;
  bucket.by_value = kmin >= 0x00800000u && kmax < 0x7f800000u;      // normal positive floats only
  bucket.base = __uint_as_float(kmin);
  bucket.scale = (float)nb / (__uint_as_float(kmax) - bucket.base);
  if (bucket.by_value && !(bucket.scale < 3.0e38f)) bucket.by_value = 0;
  if (!bucket.by_value) bucket.scale = (float)nb / (float)(kmax - kmin);

  for (uint32_t j = t; j < nb; j += TS_THREADS) cnt[j] = 0u;
  __syncthreads();
  uint32_t bkt[R], slot[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    bkt[r] = 0u; slot[r] = 0u;
    if (t + TS_THREADS * r < n) {
      bkt[r] = bucket(key[r]);
      slot[r] = atomicAdd(&cnt[bkt[r]], 1u);
    }
  }
  __syncthreads();

  // bucket offsets: thread t owns buckets [t * per, (t + 1) * per)
  const uint32_t per = (nb + TS_THREADS - 1u) / TS_THREADS;
  const uint32_t j0 = t * per, j1 = j0 + per < nb ? j0 + per : nb;
  uint32_t mine = 0u, squares = 0u;
  for (uint32_t j = j0; j < j1; ++j) { const uint32_t c = cnt[j]; mine += c; squares += c * c; }
  // exclusive scan of the per-thread sums and the total of the squares, through one pair of barriers
  const uint32_t inc = ts_wave_scan(mine);
  squares = ts_wave_sum(squares);
  if (lane == 63) { sh.red[wave] = inc; sh.red[TS_WAVES + wave] = squares; }
  __syncthreads();
  uint32_t run = inc - mine, cost = 0u;
#pragma unroll
  for (int w = 0; w < TS_WAVES; ++w) {
    if (w < wave) run += sh.red[w];
    cost += sh.red[TS_WAVES + w];
  }
  if (cost > (uint32_t)TS_COST_LIMIT * nb) return false;  // keys pile up (nothing written yet)
  for (uint32_t j = j0; j < j1; ++j) { const uint32_t c = cnt[j]; cnt[j] = run; run += c; }
  __syncthreads();

#pragma unroll
  for (int r = 0; r < R; ++r)
    if (t + TS_THREADS * r < n) pairs[cnt[bkt[r]] + slot[r]] = ((uint64_t)key[r] << 32) | id[r];
  __syncthreads();

#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = t + TS_THREADS * r;
    if (i < n) {
      const uint64_t p = pairs[i];
      const uint32_t g = bucket((uint32_t)(p >> 32));
      const uint32_t first = cnt[g], last = g + 1u < nb ? cnt[g + 1u] : nb;
      uint32_t rank = first;
      for (uint32_t m = first; m < last; ++m) rank += pairs[m] < p ? 1u : 0u;
      o2p[b + rank] = (int32_t)(uint32_t)p;
    }
  }
  return true;
}

// One workgroup per tile (runs of lo < n <= hi <= 256 * R entries; the others belong to the kernel below).  Leaves
// alt[b] = TS_DECLINED for a run it does not sort, 0 otherwise.
template <int R>
__global__ void __launch_bounds__(TS_THREADS, R <= 4 ? 8 : 1)
tile_depth_sort_kernel(const int32_t* __restrict__ ranges, uint64_t* __restrict__ srt, int32_t* __restrict__ o2p,
                       uint64_t* __restrict__ alt, int lo, int hi, int32_t* __restrict__ run_stats) {
  constexpr int CAP = TS_THREADS * R;
  __shared__ uint64_t pairs[CAP];          // key << 32 | point index, grouped by bucket
  __shared__ uint32_t cnt[CAP];            // bucket sizes, then bucket offsets
  __shared__ TsShared sh;
  const int64_t tile = blockIdx.x;
  const int64_t b = ranges[2 * tile];
  const int n = ranges[2 * tile + 1] - (int32_t)b;
  if (n <= lo || n > hi) return;           // lo >= 1: a single entry is sorted
  const bool sorted = tile_bucket_sort<R>(srt, o2p, b, n, pairs, cnt, sh);
  if (threadIdx.x == 0) {
    alt[b] = sorted ? 0ull : TS_DECLINED;
    if (!sorted && run_stats) atomicOr(run_stats + 2, 1);          // tells the long-run kernel to look for the marks
  }
}

// The long runs (n > lo) and the declined ones: 60 KB of LDS per workgroup, so the grid is a few workgroups per CU and
// each takes a contiguous share of the tiles — one thread looks at one tile's run, the workgroup then sorts the ones
// that are its business: in LDS up to 256 * R entries, by the bounded path beyond (or when declined).
template <int R>
__global__ void __launch_bounds__(TS_THREADS)
tile_depth_sort_long_kernel(const int32_t* __restrict__ ranges, int64_t num_tiles, uint64_t* __restrict__ srt,
                            int32_t* __restrict__ o2p, uint64_t* __restrict__ alt, int lo, int32_t* __restrict__ run_stats,
                            int32_t* __restrict__ run_host) {
  constexpr int CAP = TS_THREADS * R;
  static_assert(CAP >= TS_WAVES * 256, "the bucket counters double as the bounded path's digit counters");
  __shared__ uint64_t pairs[CAP];
  __shared__ uint32_t cnt[CAP];
  __shared__ TsShared sh;
  __shared__ int32_t todo[TS_THREADS];
  __shared__ int32_t todo_count;
  const int64_t share = (num_tiles + gridDim.x - 1) / gridDim.x;
  const int64_t first = (int64_t)blockIdx.x * share;
  const int64_t last = first + share < num_tiles ? first + share : num_tiles;
  const bool any_declined = run_stats ? run_stats[2] != 0 : true;   // without the flag word every mark is read
  for (int64_t base = first; base < last; base += TS_THREADS) {
    __syncthreads();
    if (threadIdx.x == 0) todo_count = 0;
    __syncthreads();
    const int64_t mine = base + threadIdx.x;
    if (mine < last) {
      const int64_t mb = ranges[2 * mine];
      const int mn = ranges[2 * mine + 1] - (int32_t)mb;
      // bit 30 of the entry: declined by a per-tile kernel
      if (mn > lo) {
        todo[atomicAdd(&todo_count, 1)] = (int32_t)(mine - base);
        // a giant run (one workgroup, ~12 ns per entry) is reported to the host: any of them, if there are several
        if (mn > TS_REPORT_ABOVE && run_host) *run_host = mn;
      }
      else if (any_declined && mn > 1 && alt[mb] == TS_DECLINED) todo[atomicAdd(&todo_count, 1)] = (int32_t)(mine - base) | (1 << 30);
    }
    __syncthreads();
    const int count = todo_count;
    for (int j = 0; j < count; ++j) {
      const int entry = todo[j];
      const int64_t tile = base + (entry & 0xffff);
      const int64_t b = ranges[2 * tile];
      const int n = ranges[2 * tile + 1] - (int32_t)b;
      bool sorted = false;
      if (n <= CAP && !(entry >> 30)) sorted = tile_bucket_sort<R>(srt, o2p, b, n, pairs, cnt, sh);
      if (!sorted) {
        __syncthreads();
        tile_radix_sort_global(srt, o2p, alt, b, n, sh, cnt);
      }
      __syncthreads();                     // done with the shared arrays before the next run
    }
  }
}

void tile_depth_sort_launch(const int32_t* tile_ranges, int64_t num_tiles, uint64_t* sorted_keys, int32_t* overlap_to_point,
                            uint64_t* scratch, hipStream_t s, int32_t* run_stats, int32_t* run_host) {
  if (num_tiles <= 0) return;
  const dim3 per_tile((unsigned)num_tiles), block(TS_THREADS);
  int covered = TS_THREADS * TS_SMALL_R;
  tile_depth_sort_kernel<TS_SMALL_R><<<per_tile, block, 0, s>>>(tile_ranges, sorted_keys, overlap_to_point, scratch, 1, covered,
                                                                run_stats);
#if TS_MID_R > 0
  tile_depth_sort_kernel<TS_MID_R><<<per_tile, block, 0, s>>>(tile_ranges, sorted_keys, overlap_to_point, scratch, covered,
                                                             TS_THREADS * TS_MID_R, run_stats);
  covered = TS_THREADS * TS_MID_R;
#endif
  const int64_t few = 2 * 256;             // two workgroups of the long-run kernel fit a CU
  tile_depth_sort_long_kernel<TS_LONG_R><<<dim3((unsigned)(num_tiles < few ? num_tiles : few)), block, 0, s>>>(
      tile_ranges, num_tiles, sorted_keys, overlap_to_point, scratch, covered, run_stats, run_host);
}

}  // namespace ms

using namespace ms;

extern "C" int ms_tile_depth_sort(const int32_t* tile_ranges, int64_t num_tiles, uint64_t* sorted_keys,
                                  int32_t* overlap_to_point, uint64_t* scratch, void* stream) {
  MS_CHECK_ARG(num_tiles >= 0 && num_tiles < (1ll << 31), "num_tiles out of range");
  if (num_tiles == 0) return 0;
  MS_CHECK_ARG(tile_ranges && sorted_keys && overlap_to_point && scratch, "null pointer");
  tile_depth_sort_launch(tile_ranges, num_tiles, sorted_keys, overlap_to_point, scratch, (hipStream_t)stream);
  MS_CHECK_LAUNCH();
  return 0;
}
