// scan_sort.hip — integer primitives of the tile mapper, hand-written for wave64:
//   * exclusive prefix sum of int32 (replaces cub::DeviceScan::ExclusiveSum, full_cumsum.cu:17-47)
//   * stable LSD radix sort of (key, int32 value) pairs, 8-bit digits, u32/u64 keys
//     (replaces cub::DeviceRadixSort::SortPairs, radix_sort_pairs.cu:8-70)
//   * segmented pair sort (cub::DeviceSegmentedSort::SortPairs, segmented_sort_pairs.cu:9-73;
//     not on the render path — kept for API parity)
//
// Radix pass = upsweep (per-block digit histogram) -> one workgroup per digit scans its row of the digit-major
// histogram (the 256 digit totals are scanned by every downsweep block itself) -> downsweep (stable in-block
// ranking with wave64 ballots, then scatter): three launches.  Block item order is
// wave-major, round-major, lane-minor, so loads are fully coalesced and stability only needs
// (earlier waves) + (earlier rounds of this wave) + (lower lanes of this round).
#include "common.h"
#include "frame_internal.h"

namespace ms {

// ------------------------------------------------------------------------------------------------
// exclusive scan
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// exclusive scan of one int per thread across a 256-thread block; returns the exclusive prefix,
// *block_total receives the block sum (valid in all threads).
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds /*>= 4 ints*/, int* block_total) {
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  int wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    const int s = lds[w];
    if (w < wave) wave_off += s;
    total += s;
  }
  __syncthreads();
  *block_total = total;
  return wave_off + incl - v;
}

// a thread's SCAN_ITEMS consecutive ints: four 128-bit loads when the tile is complete and aligned
__device__ __forceinline__ void load_items(const int32_t* __restrict__ in, int64_t base, int64_t n, int (&vals)[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n && (reinterpret_cast<uintptr_t>(in + base) & 15) == 0) {
    const int4* p = reinterpret_cast<const int4*>(in + base);
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS / 4; ++q) {
      const int4 v = p[q];
      vals[4 * q] = v.x; vals[4 * q + 1] = v.y; vals[4 * q + 2] = v.z; vals[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) vals[k] = base + k < n ? in[base + k] : 0;
  }
}

__device__ __forceinline__ void store_items(int32_t* __restrict__ out, int64_t base, int64_t n, const int (&vals)[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n && (reinterpret_cast<uintptr_t>(out + base) & 15) == 0) {
    int4* p = reinterpret_cast<int4*>(out + base);
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS / 4; ++q) p[q] = make_int4(vals[4 * q], vals[4 * q + 1], vals[4 * q + 2], vals[4 * q + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) out[base + k] = vals[k];
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_block_sums_kernel(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ block_sums) {
  __shared__ int lds[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int items[SCAN_ITEMS];
  load_items(in, base, n, items);
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) s += items[k];
  int total;
  block_exclusive_scan(s, lds, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of block_sums in place; writes the grand total
__global__ void __launch_bounds__(SCAN_THREADS)
scan_spine_kernel(int32_t* __restrict__ block_sums, int64_t num_blocks, int32_t* __restrict__ total_out,
                  int32_t* __restrict__ total_host, int32_t* __restrict__ total_copy = nullptr) {
  __shared__ int lds[4];
  int carry = 0;
  for (int64_t base = 0; base < num_blocks; base += SCAN_THREADS) {
    const int64_t i = base + threadIdx.x;
    const int v = i < num_blocks ? block_sums[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, lds, &total);
    if (i < num_blocks) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) {
    *total_out = carry;
    if (total_copy) *total_copy = carry;
    if (total_host) *total_host = carry;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_downsweep_kernel(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ block_offsets,
                      int32_t* __restrict__ out) {
  __shared__ int lds[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int vals[SCAN_ITEMS];
  load_items(in, base, n, vals);
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) s += vals[k];
  int total;
  int prefix = block_exclusive_scan(s, lds, &total) + block_offsets[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {       // vals[k] <- exclusive prefix of item k
    const int v = vals[k];
    vals[k] = prefix;
    prefix += v;
  }
  store_items(out, base, n, vals);
}

// Two-launch variant for up to SCAN_SELF_MAX_BLOCKS tiles: every block sums the totals of the tiles before
// it itself (<= 32 KB of L2-resident reads) instead of waiting for a single-workgroup spine launch; the last
// block also writes the grand total.  Saves one dependent launch (~5 us) per scan: the frame runs eight.
constexpr int64_t SCAN_SELF_MAX_BLOCKS = 8192;

__global__ void __launch_bounds__(SCAN_THREADS)
scan_downsweep_self_kernel(const int32_t* __restrict__ in, int64_t n, const int32_t* __restrict__ block_sums,
                           int32_t* __restrict__ out, int32_t* __restrict__ total_host,
                           int32_t* __restrict__ total_copy = nullptr) {
  __shared__ int lds[4];
  int before = 0;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += SCAN_THREADS) before += block_sums[j];
  int block_prefix;
  block_exclusive_scan(before, lds, &block_prefix);

  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int vals[SCAN_ITEMS];
  load_items(in, base, n, vals);
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) s += vals[k];
  int total;
  int prefix = block_exclusive_scan(s, lds, &total) + block_prefix;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {       // vals[k] <- exclusive prefix of item k
    const int v = vals[k];
    vals[k] = prefix;
    prefix += v;
  }
  store_items(out, base, n, vals);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    out[n] = block_prefix + total;
    if (total_copy) *total_copy = block_prefix + total;         // frame executor: counters[0]
    if (total_host) *total_host = block_prefix + total;
  }
}

static size_t scan_tmp_bytes(int64_t n) { return align_up((size_t)div_up(n > 0 ? n : 1, SCAN_TILE) * sizeof(int32_t), 256); }

// out has n + 1 entries (out[n] = total).  in and out may NOT alias unless identical ranges are
// intended (in == out is allowed: every block reads its tile before writing it).
static int exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* total_host, void* tmp,
                              hipStream_t s, int32_t* total_copy = nullptr) {
  const int64_t blocks = div_up(n, SCAN_TILE);
  int32_t* block_sums = (int32_t*)tmp;
  scan_block_sums_kernel<<<dim3((unsigned)blocks), dim3(SCAN_THREADS), 0, s>>>(in, n, block_sums);
  if (blocks <= SCAN_SELF_MAX_BLOCKS) {
    scan_downsweep_self_kernel<<<dim3((unsigned)blocks), dim3(SCAN_THREADS), 0, s>>>(in, n, block_sums, out, total_host, total_copy);
    return 0;
  }
  scan_spine_kernel<<<dim3(1), dim3(SCAN_THREADS), 0, s>>>(block_sums, blocks, out + n, total_host, total_copy);
  scan_downsweep_kernel<<<dim3((unsigned)blocks), dim3(SCAN_THREADS), 0, s>>>(in, n, block_sums, out);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// radix sort
// ------------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ROUNDS = 16;                          // items per thread
constexpr int RS_WAVE_ITEMS = RS_ROUNDS * 64;          // 1024
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;        // 4096 items per block
constexpr int RS_RADIX = 256;

template <typename KeyT>
__device__ __forceinline__ unsigned key_digit(KeyT k, int shift, unsigned mask) {
  return (unsigned)(k >> shift) & mask;
}

// Where a pass reads its (key, value) pairs from.  PlainPairs: two arrays.  DepthPairs: the first pass of the mapper's
// depth pre-sort makes its keys on the fly — float bits of the (ndc) depth or the 16-bit quantisation, exactly
// depth_sort_key() of mapper.hip's depth_keys_kernel — and the value is the item's index: no key / value arrays are
// written or read before the first scatter (ms_depth_argsort).
template <typename KeyT> struct PlainPairs {
  const KeyT* keys;
  const int32_t* vals;
  __device__ __forceinline__ KeyT key(int64_t i) const { return keys[i]; }
  __device__ __forceinline__ int32_t val(int64_t i) const { return vals[i]; }
};
template <typename T> struct DepthPairs {
  const T* depth;
  int depth16;
  double near_plane, far_plane;
  int cull;          // frame executor: depth <= 0 marks a culled gaussian, which sorts last and is skipped downstream
  __device__ __forceinline__ uint32_t key(int64_t i) const {
    const T d = depth[i];
    if (cull && !(d > T(0))) return CULLED_DEPTH_KEY;
    return depth_sort_key(d, depth16, near_plane, far_plane);
  }
  __device__ __forceinline__ int32_t val(int64_t i) const { return (int32_t)i; }
};

// VARY: additionally the bitwise OR and AND of the block's keys (CULLED_DEPTH_KEY rows left out) go to blk_or / blk_and:
// the adaptive depth sort skips the passes whose digit is the same in every key (radix_sort_depth_adaptive)
template <typename KeyT, typename Source, bool VARY = false>
__device__ __forceinline__ void upsweep_block(const Source src, int64_t n, int shift, unsigned mask,
                                              int32_t* __restrict__ hist, int64_t num_blocks,
                                              uint32_t* __restrict__ blk_or = nullptr, uint32_t* __restrict__ blk_and = nullptr) {
  // RS_COPIES counter sets per wave (lane & 3 picks one): the last pass of the depth sort sees a handful of distinct
  // digits (sign + high exponent bits), and 64 lanes adding to one LDS word are served one after the other
  // (25 us against 11 us for the other passes over the same 6 M keys with a single set)
  constexpr int RS_COPIES = 4;
  __shared__ unsigned cnt[RS_WAVES][RS_COPIES][RS_RADIX];
  __shared__ unsigned s_or, s_and;
  for (int i = threadIdx.x; i < RS_WAVES * RS_COPIES * RS_RADIX; i += RS_THREADS) (&cnt[0][0][0])[i] = 0;
  if (VARY && threadIdx.x == 0) { s_or = 0u; s_and = ~0u; }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = lane_id();
  unsigned* mine = cnt[wave][lane & (RS_COPIES - 1)];
  const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * RS_WAVE_ITEMS + lane;
  unsigned v_or = 0u, v_and = ~0u;
  if ((int64_t)(blockIdx.x + 1) * RS_TILE <= n) {        // every item of the block exists: no bounds checks
    KeyT k[RS_ROUNDS];
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; ++j) k[j] = src.key(base + j * 64);
#pragma unroll
    for (int j = 0; j < RS_ROUNDS; ++j) {
      atomicAdd(&mine[key_digit(k[j], shift, mask)], 1u);
      if (VARY && (uint32_t)k[j] != 0xffffffffu) { v_or |= (uint32_t)k[j]; v_and &= (uint32_t)k[j]; }
    }
  } else {
#pragma unroll 4
    for (int j = 0; j < RS_ROUNDS; ++j) {
      const int64_t i = base + j * 64;
      if (i < n) {
        const KeyT key = src.key(i);
        atomicAdd(&mine[key_digit(key, shift, mask)], 1u);
        if (VARY && (uint32_t)key != 0xffffffffu) { v_or |= (uint32_t)key; v_and &= (uint32_t)key; }
      }
    }
  }
  if (VARY) {
    // one LDS atomic per wave, not per thread (256 atomics on one word are served one after the other: +18 us)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { v_or |= __shfl_xor(v_or, off); v_and &= __shfl_xor(v_and, off); }
    if (lane == 0) { atomicOr(&s_or, v_or); atomicAnd(&s_and, v_and); }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RS_RADIX; d += RS_THREADS) {
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w)
#pragma unroll
      for (int c = 0; c < RS_COPIES; ++c) t += cnt[w][c][d];
    hist[(int64_t)d * num_blocks + blockIdx.x] = (int32_t)t;
  }
  if (VARY && threadIdx.x == 0) { blk_or[blockIdx.x] = s_or; blk_and[blockIdx.x] = s_and; }
}

template <typename KeyT, typename Source>
__global__ void __launch_bounds__(RS_THREADS)
radix_upsweep_kernel(const Source src, int64_t n, const int32_t* __restrict__ n_dev, int shift, unsigned mask,
                     int32_t* __restrict__ hist, int64_t num_blocks) {
  if (n_dev) n = *n_dev;        // live count on the device; the grid covers the capacity (idle blocks write zeros)
  upsweep_block<KeyT, Source>(src, n, shift, mask, hist, num_blocks);
}

// One workgroup per digit: exclusive scan of that digit's per-block counts (row d of the digit-major histogram)
// in place, and the digit's total.  Together with the 256-entry scan of the totals that every downsweep block does
// for itself, this replaces the general two-launch scan of the whole 256 x blocks table per radix pass.
__device__ __forceinline__ void row_scan_body(int32_t* __restrict__ hist, int64_t num_blocks, int32_t* __restrict__ digit_totals) {
  __shared__ int lds[4];
  int32_t* row = hist + (int64_t)blockIdx.x * num_blocks;
  int carry = 0;
  // SCAN_TILE entries per round, 16 consecutive ones per thread: a row of config D's tile sort (3100 blocks) is one
  // round with two barriers instead of thirteen rounds of one entry per thread (11.6 -> ~6 us per launch)
  for (int64_t base = 0; base < num_blocks; base += SCAN_TILE) {
    const int64_t mine = base + (int64_t)threadIdx.x * SCAN_ITEMS;
    int vals[SCAN_ITEMS];
    load_items(row, mine, num_blocks, vals);
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += vals[k];
    int total;
    int prefix = carry + block_exclusive_scan(s, lds, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      const int v = vals[k];
      vals[k] = prefix;
      prefix += v;
    }
    store_items(row, mine, num_blocks, vals);
    carry += total;
  }
  if (threadIdx.x == 0) digit_totals[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(SCAN_THREADS)
radix_row_scan_kernel(int32_t* __restrict__ hist, int64_t num_blocks, int32_t* __restrict__ digit_totals) {
  row_scan_body(hist, num_blocks, digit_totals);
}

template <typename KeyT> struct DownsweepShared {
  unsigned cnt[RS_WAVES][RS_RADIX];
  int scan[4];
  // global start of this block's run of digit d MINUS the block-local start of the digit: an item at local position idx
  // goes to digit_shift[d] + idx (unsigned wrap-around is fine).  One array instead of two: with 8-byte keys the block's
  // LDS is 54 288 bytes instead of 55 312, i.e. three workgroups per CU instead of two (tools/kernel_resources.py)
  unsigned digit_shift[RS_RADIX];
  KeyT keys[RS_TILE];
  int32_t vals[RS_TILE];
};

// FULL: every one of the block's RS_TILE items exists (all blocks but the last): no bounds checks at all.  With
// them each of the 32 loads of a thread sat in its own EXEC-masked block behind a 64-bit compare, and the ranking and
// the scatter carried a validity flag per item.
template <typename KeyT, bool FULL, typename Source>
__device__ __forceinline__ void downsweep_block(DownsweepShared<KeyT>& sh, const Source src, KeyT* __restrict__ keys_out,
                                                int32_t* __restrict__ vals_out, int64_t n, int shift, unsigned mask,
                                                const int32_t* __restrict__ hist_scanned,
                                                const int32_t* __restrict__ digit_totals, int64_t num_blocks) {
  for (int i = threadIdx.x; i < RS_WAVES * RS_RADIX; i += RS_THREADS) (&sh.cnt[0][0])[i] = 0;
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * RS_WAVE_ITEMS + lane;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;

  KeyT k[RS_ROUNDS];
  int32_t v[RS_ROUNDS];
  unsigned rank[RS_ROUNDS];   // rank of the item among equal digits of this wave

#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const int64_t i = base + j * 64;
    const bool valid = FULL || i < n;
    k[j] = valid ? src.key(i) : (KeyT)0;
    v[j] = valid ? src.val(i) : 0;
  }

#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const int64_t i = base + j * 64;
    const bool valid = FULL || i < n;
    const unsigned d = key_digit(k[j], shift, mask);
    // match-any over the 8 digit bits: peers = lanes holding the same digit.  Per bit: the lane's bit as a 0 / ~0
    // mask (one v_bfe_i32), its ballot, and per 32-lane half  peers &= ~(ballot ^ mask)  (one v_bitop3_b32 each) —
    // four VALU instructions per bit; the 64-bit select form compiled to nine.
    const unsigned long long valid_lanes = FULL ? ~0ull : __ballot(valid);
    unsigned peers_lo = (unsigned)valid_lanes, peers_hi = (unsigned)(valid_lanes >> 32);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int m = __builtin_amdgcn_sbfe((int)d, b, 1);          // 0 or -1
      const unsigned long long bal = __ballot(m != 0);
      peers_lo &= ~((unsigned)bal ^ (unsigned)m);
      peers_hi &= ~((unsigned)(bal >> 32) ^ (unsigned)m);
    }
    const unsigned long long peers = ((unsigned long long)peers_hi << 32) | peers_lo;
    const unsigned below = (unsigned)__popcll(peers & lanes_below);
    unsigned prev = 0;
    if (valid) {
      prev = sh.cnt[wave][d];
    }
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) sh.cnt[wave][d] = prev + (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rank[j] = prev + below;
  }
  __syncthreads();

  // per digit: exclusive offsets of the waves inside the block (cnt[w][d] <- local start of wave w's
  // run of digit d), the block-local start of the digit (exclusive scan over the 256 digit totals)
  // and the global start of this block's run of the digit
  {
    const int d = threadIdx.x;                    // RS_THREADS == RS_RADIX
    unsigned total = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) total += sh.cnt[w][d];
    int block_total;
    const unsigned local = (unsigned)block_exclusive_scan((int)total, sh.scan, &block_total);
    unsigned run = local;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      const unsigned c = sh.cnt[w][d];
      sh.cnt[w][d] = run;
      run += c;
    }
    // global start of the digit = items of all smaller digits (256-entry scan, done by every block for itself)
    // + items of this digit in earlier blocks (radix_row_scan_kernel)
    int all_items;
    const unsigned digit_base = (unsigned)block_exclusive_scan(digit_totals[d], sh.scan, &all_items);
    sh.digit_shift[d] = digit_base + (unsigned)hist_scanned[(int64_t)d * num_blocks + blockIdx.x] - local;
  }
  __syncthreads();

  // reorder through LDS so the global scatter is written in digit order: consecutive threads then
  // write consecutive addresses inside each digit run (coalesced 64 B+ segments instead of one
  // transaction per lane)
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const int64_t i = base + j * 64;
    if (FULL || i < n) {
      const unsigned d = key_digit(k[j], shift, mask);
      const unsigned lp = sh.cnt[wave][d] + rank[j];
      sh.keys[lp] = k[j];
      sh.vals[lp] = v[j];
    }
  }
  __syncthreads();
  const int64_t block_base = (int64_t)blockIdx.x * RS_TILE;
  const int block_items = FULL ? RS_TILE : (int)(n - block_base);
#pragma unroll
  for (int j = 0; j < RS_ROUNDS; ++j) {
    const int idx = (int)threadIdx.x + j * RS_THREADS;
    if (FULL || idx < block_items) {
      const KeyT key = sh.keys[idx];
      const unsigned d = key_digit(key, shift, mask);
      const int64_t pos = (int64_t)(unsigned)(sh.digit_shift[d] + (unsigned)idx);
      keys_out[pos] = key;
      vals_out[pos] = sh.vals[idx];
    }
  }
}

template <typename KeyT, typename Source>
__global__ void __launch_bounds__(RS_THREADS)
radix_downsweep_kernel(const Source src,
                       KeyT* __restrict__ keys_out, int32_t* __restrict__ vals_out, int64_t n,
                       const int32_t* __restrict__ n_dev, int shift,
                       unsigned mask, const int32_t* __restrict__ hist_scanned,
                       const int32_t* __restrict__ digit_totals, int64_t num_blocks) {
  __shared__ DownsweepShared<KeyT> sh;
  if (n_dev) {
    n = *n_dev;
    if ((int64_t)blockIdx.x * RS_TILE >= n) return;
  }
  if ((int64_t)(blockIdx.x + 1) * RS_TILE <= n)
    downsweep_block<KeyT, true>(sh, src, keys_out, vals_out, n, shift, mask, hist_scanned, digit_totals, num_blocks);
  else
    downsweep_block<KeyT, false>(sh, src, keys_out, vals_out, n, shift, mask, hist_scanned, digit_totals, num_blocks);
}

template <typename KeyT>
__global__ void __launch_bounds__(256)
copy_pairs_kernel(const KeyT* __restrict__ ki, const int32_t* __restrict__ vi, KeyT* __restrict__ ko,
                  int32_t* __restrict__ vo, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ko[i] = ki[i]; vo[i] = vi[i]; }
}

struct SortTmp {
  size_t hist_off, scan_off, keys_off, vals_off, keys2_off, vals2_off, vary_off, total;
};

// adaptive (radix_sort_depth_adaptive): a second alternate pair of buffers and the per-block OR / AND words
static SortTmp sort_tmp_layout(int64_t n, int key_bytes, bool adaptive = false) {
  const int64_t blocks = div_up(n > 0 ? n : 1, RS_TILE);
  SortTmp t;
  size_t off = 0;
  t.hist_off = off; off += align_up((size_t)(RS_RADIX * blocks + 1) * sizeof(int32_t), 256);
  t.scan_off = off; off += align_up((size_t)RS_RADIX * sizeof(int32_t), 256);      // digit totals
  t.keys_off = off; off += align_up((size_t)n * key_bytes, 256);
  t.vals_off = off; off += align_up((size_t)n * sizeof(int32_t), 256);
  t.keys2_off = t.vals2_off = t.vary_off = off;
  if (adaptive) {
    t.keys2_off = off; off += align_up((size_t)n * key_bytes, 256);
    t.vals2_off = off; off += align_up((size_t)n * sizeof(int32_t), 256);
    t.vary_off = off; off += align_up((size_t)(64 + 2 * blocks) * sizeof(uint32_t), 256);
  }
  t.total = off;
  return t;
}

// `first` = where pass 0 reads its pairs (PlainPairs of the caller's arrays, or DepthPairs); later passes read the
// ping-pong buffers.
template <typename KeyT, typename First>
static int radix_sort_passes(const First first, KeyT* keys_out, int32_t* vals_out, int64_t n, int begin_bit, int end_bit,
                             char* tmp, hipStream_t s, const int32_t* n_dev = nullptr) {
  const int64_t blocks = div_up(n, RS_TILE);
  const SortTmp lay = sort_tmp_layout(n, sizeof(KeyT));
  int32_t* hist = (int32_t*)(tmp + lay.hist_off);
  int32_t* digit_totals = (int32_t*)(tmp + lay.scan_off);       // 256 ints
  KeyT* keys_alt = (KeyT*)(tmp + lay.keys_off);
  int32_t* vals_alt = (int32_t*)(tmp + lay.vals_off);
  const int passes = (end_bit - begin_bit + 7) / 8;
  const dim3 grid((unsigned)blocks), block(RS_THREADS);
  PlainPairs<KeyT> src{nullptr, nullptr};
  // digits of equal width (14 tile bits = 7 + 7, not 8 + 6): fewer bins -> longer contiguous runs in the scatter
  const int total_bits = end_bit - begin_bit, base_bits = total_bits / passes, wide = total_bits % passes;
  int shift = begin_bit;
  for (int p = 0; p < passes; shift += base_bits + (p < wide ? 1 : 0), ++p) {
    const int bits = base_bits + (p < wide ? 1 : 0);
    const unsigned mask = (1u << bits) - 1u;
    const bool to_out = ((passes - 1 - p) % 2) == 0;
    KeyT* dst_k = to_out ? keys_out : keys_alt;
    int32_t* dst_v = to_out ? vals_out : vals_alt;
    if (p == 0) radix_upsweep_kernel<KeyT, First><<<grid, block, 0, s>>>(first, n, n_dev, shift, mask, hist, blocks);
    else radix_upsweep_kernel<KeyT, PlainPairs<KeyT>><<<grid, block, 0, s>>>(src, n, n_dev, shift, mask, hist, blocks);
    radix_row_scan_kernel<<<dim3(RS_RADIX), dim3(SCAN_THREADS), 0, s>>>(hist, blocks, digit_totals);
    if (p == 0) radix_downsweep_kernel<KeyT, First><<<grid, block, 0, s>>>(first, dst_k, dst_v, n, n_dev, shift, mask, hist, digit_totals, blocks);
    else radix_downsweep_kernel<KeyT, PlainPairs<KeyT>><<<grid, block, 0, s>>>(src, dst_k, dst_v, n, n_dev, shift, mask, hist, digit_totals, blocks);
    src = PlainPairs<KeyT>{dst_k, dst_v};
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Depth pre-sort with the constant-digit passes left out (frame executor).
//
// ndc depth keys are float bits in (0, 1): sign and top exponent bit are zero, and a scene seen from outside its
// near range has every depth in [0.5, 1) — one exponent, nine constant leading bits.  A pass whose 8-bit digit is the
// same in EVERY key is the identity permutation.  The launch sequence stays fixed (HIP-graph capturable, no host
// read): pass 0 — which makes the keys anyway — also ORs and ANDs them per block, its row-scan launch folds those
// into `vary` = the bits that differ anywhere, and the three kernels of a later pass whose digit does not vary return
// at once.  So that the result always ends in the caller's arrays, the active passes chain through two alternate
// buffers and the LAST ACTIVE one writes to `out` (route()).  Culled rows (key 0xffffffff) are left out of OR / AND:
// the consumers skip them wherever they land, and the valid rows keep their order.  Config D: passes 0-2 run, pass 3
// (bits 24-31 = 0x3f everywhere) is skipped: 0.20 -> 0.15 ms.
struct AdaptiveBufs {
  uint32_t *k_alt[2], *k_out;
  int32_t *v_alt[2], *v_out;
  uint32_t* vary;             // [0] = bits that differ between keys; then blk_or[blocks], blk_and[blocks] from word 64 on
};

__device__ __forceinline__ void adaptive_route(uint32_t vary, int p, int passes, bool& active, int& k, bool& last) {
  unsigned m = 1u;                                   // pass 0 always runs
  for (int q = 1; q < passes; ++q)
    if ((vary >> (8 * q)) & 0xffu) m |= 1u << q;
  active = ((m >> p) & 1u) != 0u;
  k = __popc(m & ((1u << p) - 1u));                  // active passes before this one
  last = (m >> (p + 1)) == 0u;
}

template <typename First>
__global__ void __launch_bounds__(RS_THREADS)
depth_upsweep_first_kernel(const First first, AdaptiveBufs b, int64_t n, int32_t* __restrict__ hist, int64_t num_blocks) {
  upsweep_block<uint32_t, First, true>(first, n, 0, 0xffu, hist, num_blocks, b.vary + 64, b.vary + 64 + num_blocks);
}

__global__ void __launch_bounds__(RS_THREADS)
depth_upsweep_kernel(AdaptiveBufs b, int p, int passes, int64_t n, int32_t* __restrict__ hist, int64_t num_blocks) {
  bool active, last; int k;
  adaptive_route(b.vary[0], p, passes, active, k, last);
  if (!active) return;
  upsweep_block<uint32_t, PlainPairs<uint32_t>>(PlainPairs<uint32_t>{b.k_alt[(k - 1) & 1], b.v_alt[(k - 1) & 1]}, n, 8 * p, 0xffu, hist, num_blocks);
}

__global__ void __launch_bounds__(SCAN_THREADS)
depth_row_scan_kernel(int32_t* __restrict__ hist, int64_t num_blocks, int32_t* __restrict__ digit_totals, AdaptiveBufs b,
                      int p, int passes) {
  if (p > 0) {
    bool active, last; int k;
    adaptive_route(b.vary[0], p, passes, active, k, last);
    if (!active) return;
  }
  row_scan_body(hist, num_blocks, digit_totals);
  if (p == 0 && blockIdx.x == 0) {                   // fold the blocks' OR / AND words: which bits differ anywhere
    __shared__ unsigned s_or, s_and;
    if (threadIdx.x == 0) { s_or = 0u; s_and = ~0u; }
    __syncthreads();
    unsigned v_or = 0u, v_and = ~0u;
    for (int64_t i = threadIdx.x; i < num_blocks; i += blockDim.x) { v_or |= b.vary[64 + i]; v_and &= b.vary[64 + num_blocks + i]; }
    atomicOr(&s_or, v_or); atomicAnd(&s_and, v_and);
    __syncthreads();
    if (threadIdx.x == 0) b.vary[0] = s_or >= s_and && s_or != 0u ? (s_or ^ s_and) : 0u;      // no valid key: nothing varies
  }
}

template <typename First>
__global__ void __launch_bounds__(RS_THREADS)
depth_downsweep_first_kernel(const First first, AdaptiveBufs b, int passes, int64_t n,
                             const int32_t* __restrict__ hist_scanned, const int32_t* __restrict__ digit_totals, int64_t num_blocks) {
  __shared__ DownsweepShared<uint32_t> sh;
  bool active, last; int k;
  adaptive_route(b.vary[0], 0, passes, active, k, last);
  uint32_t* ko = last ? b.k_out : b.k_alt[0];
  int32_t* vo = last ? b.v_out : b.v_alt[0];
  if ((int64_t)(blockIdx.x + 1) * RS_TILE <= n) downsweep_block<uint32_t, true>(sh, first, ko, vo, n, 0, 0xffu, hist_scanned, digit_totals, num_blocks);
  else downsweep_block<uint32_t, false>(sh, first, ko, vo, n, 0, 0xffu, hist_scanned, digit_totals, num_blocks);
}

__global__ void __launch_bounds__(RS_THREADS)
depth_downsweep_kernel(AdaptiveBufs b, int p, int passes, int64_t n, const int32_t* __restrict__ hist_scanned,
                       const int32_t* __restrict__ digit_totals, int64_t num_blocks) {
  __shared__ DownsweepShared<uint32_t> sh;
  bool active, last; int k;
  adaptive_route(b.vary[0], p, passes, active, k, last);
  if (!active) return;
  uint32_t* ko = last ? b.k_out : b.k_alt[k & 1];
  int32_t* vo = last ? b.v_out : b.v_alt[k & 1];
  const PlainPairs<uint32_t> src{b.k_alt[(k - 1) & 1], b.v_alt[(k - 1) & 1]};
  if ((int64_t)(blockIdx.x + 1) * RS_TILE <= n) downsweep_block<uint32_t, true>(sh, src, ko, vo, n, 8 * p, 0xffu, hist_scanned, digit_totals, num_blocks);
  else downsweep_block<uint32_t, false>(sh, src, ko, vo, n, 8 * p, 0xffu, hist_scanned, digit_totals, num_blocks);
}

template <typename First>
static int radix_sort_depth_adaptive(const First first, uint32_t* keys_out, int32_t* vals_out, int64_t n, int end_bit,
                                     char* tmp, hipStream_t s) {
  const int64_t blocks = div_up(n, RS_TILE);
  const SortTmp lay = sort_tmp_layout(n, 4, true);
  int32_t* hist = (int32_t*)(tmp + lay.hist_off);
  int32_t* digit_totals = (int32_t*)(tmp + lay.scan_off);
  AdaptiveBufs b;
  b.k_alt[0] = (uint32_t*)(tmp + lay.keys_off); b.v_alt[0] = (int32_t*)(tmp + lay.vals_off);
  b.k_alt[1] = (uint32_t*)(tmp + lay.keys2_off); b.v_alt[1] = (int32_t*)(tmp + lay.vals2_off);
  b.k_out = keys_out; b.v_out = vals_out;
  b.vary = (uint32_t*)(tmp + lay.vary_off);
  const int passes = (end_bit + 7) / 8;
  const dim3 grid((unsigned)blocks), block(RS_THREADS);
  for (int p = 0; p < passes; ++p) {
    if (p == 0) depth_upsweep_first_kernel<First><<<grid, block, 0, s>>>(first, b, n, hist, blocks);
    else depth_upsweep_kernel<<<grid, block, 0, s>>>(b, p, passes, n, hist, blocks);
    depth_row_scan_kernel<<<dim3(RS_RADIX), dim3(SCAN_THREADS), 0, s>>>(hist, blocks, digit_totals, b, p, passes);
    if (p == 0) depth_downsweep_first_kernel<First><<<grid, block, 0, s>>>(first, b, passes, n, hist, digit_totals, blocks);
    else depth_downsweep_kernel<<<grid, block, 0, s>>>(b, p, passes, n, hist, digit_totals, blocks);
  }
  return 0;
}

template <typename KeyT>
static int radix_sort_pairs_impl(const KeyT* keys_in, const int32_t* vals_in, KeyT* keys_out,
                                 int32_t* vals_out, int64_t n, int begin_bit, int end_bit, char* tmp,
                                 hipStream_t s) {
  if ((end_bit - begin_bit + 7) / 8 == 0) {
    copy_pairs_kernel<KeyT><<<dim3((unsigned)div_up(n, 256)), dim3(256), 0, s>>>(keys_in, vals_in, keys_out, vals_out, n);
    return 0;
  }
  return radix_sort_passes<KeyT>(PlainPairs<KeyT>{keys_in, vals_in}, keys_out, vals_out, n, begin_bit, end_bit, tmp, s);
}

// ------------------------------------------------------------------------------------------------
// segmented sort (API parity only): one block per segment, stable rank sort
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
segmented_rank_sort_kernel(const int32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in,
                           int32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                           const int64_t* __restrict__ starts, const int64_t* __restrict__ ends) {
  const int64_t b = starts[blockIdx.x], e = ends[blockIdx.x];
  for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    const int32_t k = keys_in[i];
    int64_t r = 0;
    for (int64_t j = b; j < e; ++j) {
      const int32_t kj = keys_in[j];
      r += (kj < k) || (kj == k && j < i);
    }
    keys_out[b + r] = k;
    vals_out[b + r] = vals_in[i];
  }
}

template <typename KeyT>
__global__ void __launch_bounds__(256)
find_ranges_kernel(const KeyT* __restrict__ keys, int64_t k, const int32_t* __restrict__ k_dev, int shift,
                   int64_t num_tiles, int32_t* __restrict__ ranges) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k_dev) k = *k_dev;
  if (i >= k) return;
  const int64_t tile = (int64_t)(keys[i] >> shift);
  const int64_t next = (i + 1 < k) ? (int64_t)(keys[i + 1] >> shift) : -1;
  if (i == 0 && tile < num_tiles) ranges[tile * 2 + 0] = 0;
  if (tile != next) {
    if (tile < num_tiles) ranges[tile * 2 + 1] = (int32_t)(i + 1);
    if (next >= 0 && next < num_tiles) ranges[next * 2 + 0] = (int32_t)(i + 1);
  }
}

// frame executor: u32 tile ids (shift 0), four keys per thread from one 128-bit load + the key after them
__global__ void __launch_bounds__(256)
find_ranges4_kernel(const uint32_t* __restrict__ keys, int64_t capacity, const int32_t* __restrict__ k_dev,
                    int64_t num_tiles, int32_t* __restrict__ ranges) {
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t k = k_dev ? (int64_t)*k_dev : capacity;
  if (i0 >= k) return;
  uint32_t t[5];
  if (i0 + 4 < k) {
    const uint4 v = *reinterpret_cast<const uint4*>(keys + i0);
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w; t[4] = keys[i0 + 4];
  } else {
#pragma unroll
    for (int j = 0; j < 5; ++j) t[j] = i0 + j < k ? keys[i0 + j] : 0xffffffffu;      // 0xffffffff: no next key
  }
  if (i0 == 0 && (int64_t)t[0] < num_tiles) ranges[(int64_t)t[0] * 2 + 0] = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = i0 + j;
    if (i >= k) break;
    const int64_t tile = t[j], next = (i + 1 < k) ? (int64_t)t[j + 1] : -1;
    if (tile != next) {
      if (tile < num_tiles) ranges[tile * 2 + 1] = (int32_t)(i + 1);
      if (next >= 0 && next < num_tiles) ranges[next * 2 + 0] = (int32_t)(i + 1);
    }
  }
}

// frame executor, direct-order mapper: tile id = key >> 32 of u64 keys.  Two keys per thread from ONE 128-bit load (a wave
// reads 2 KB contiguous); the key after them is the neighbour lane's first (lane 63 reads its own)
__global__ void __launch_bounds__(256)
find_ranges2_u64_kernel(const uint64_t* __restrict__ keys, int64_t capacity, const int32_t* __restrict__ k_dev,
                        int64_t num_tiles, int32_t* __restrict__ ranges) {
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const int64_t k = k_dev ? (int64_t)*k_dev : capacity;
  uint32_t t0 = 0xffffffffu, t1 = 0xffffffffu;               // 0xffffffff: no key
  if (i0 + 1 < k) {
    const uint4 a = *reinterpret_cast<const uint4*>(keys + i0);
    t0 = a.y; t1 = a.w;
  } else if (i0 < k) {
    t0 = (uint32_t)(keys[i0] >> 32);
  }
  uint32_t t2 = (uint32_t)__shfl_down((int)t0, 1);
  if ((threadIdx.x & 63) == 63) t2 = i0 + 2 < k ? (uint32_t)(keys[i0 + 2] >> 32) : 0xffffffffu;
  if (i0 >= k) return;
  if (i0 == 0 && (int64_t)t0 < num_tiles) ranges[(int64_t)t0 * 2 + 0] = 0;
  const uint32_t t[3] = {t0, t1, t2};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t i = i0 + j;
    if (i >= k) break;
    const int64_t tile = t[j], next = (i + 1 < k) ? (int64_t)t[j + 1] : -1;
    if (tile != next) {
      if (tile < num_tiles) ranges[tile * 2 + 1] = (int32_t)(i + 1);
      if (next >= 0 && next < num_tiles) ranges[next * 2 + 0] = (int32_t)(i + 1);
    }
  }
}

// ---- launchers for the frame executor (frame_internal.h) ----------------------------------------------------------
size_t scan_tmp_size(int64_t n) { return scan_tmp_bytes(n); }
size_t sort_tmp_size(int64_t n, int key_bytes, bool adaptive) { return sort_tmp_layout(n, key_bytes, adaptive).total; }

void exclusive_scan_launch(const int32_t* in, int64_t n, int32_t* out, int32_t* total_host, void* tmp, hipStream_t s,
                           int32_t* total_copy) {
  exclusive_scan_i32(in, n, out, total_host, tmp, s, total_copy);
}

void depth_argsort_launch(const void* depth, int64_t n, int depth16, double ndc_near, double ndc_far, int dtype,
                          int cull, uint32_t* out_sorted_keys, int32_t* out_order, char* tmp, hipStream_t s) {
  const int end_bit = depth16 ? 16 : 32;
  if (dtype == MS_F32)
    radix_sort_depth_adaptive(DepthPairs<float>{(const float*)depth, depth16, ndc_near, ndc_far, cull}, out_sorted_keys, out_order, n, end_bit, tmp, s);
  else
    radix_sort_depth_adaptive(DepthPairs<double>{(const double*)depth, depth16, ndc_near, ndc_far, cull}, out_sorted_keys, out_order, n, end_bit, tmp, s);
}

void sort_pairs_u32_dev_launch(const uint32_t* keys_in, const int32_t* vals_in, uint32_t* keys_out, int32_t* vals_out,
                               int64_t capacity, const int32_t* n_dev, int end_bit, char* tmp, hipStream_t s) {
  radix_sort_passes<uint32_t>(PlainPairs<uint32_t>{keys_in, vals_in}, keys_out, vals_out, capacity, 0, end_bit, tmp, s, n_dev);
}

int find_ranges_dev_launch(const uint32_t* sorted_keys, int64_t capacity, const int32_t* k_dev, int64_t num_tiles,
                           int32_t* out_ranges, hipStream_t s, bool zeroed) {
  if (num_tiles > 0 && !zeroed) MS_CHECK_HIP(hipMemsetAsync(out_ranges, 0, (size_t)num_tiles * 2 * sizeof(int32_t), s));
  if (capacity > 0)
    find_ranges4_kernel<<<dim3((unsigned)div_up(capacity, 1024)), dim3(256), 0, s>>>(sorted_keys, capacity, k_dev, num_tiles, out_ranges);
  return 0;
}

void sort_pairs_u64_dev_launch(const uint64_t* keys_in, const int32_t* vals_in, uint64_t* keys_out, int32_t* vals_out,
                               int64_t capacity, const int32_t* n_dev, int begin_bit, int end_bit, char* tmp, hipStream_t s) {
  radix_sort_passes<uint64_t>(PlainPairs<uint64_t>{keys_in, vals_in}, keys_out, vals_out, capacity, begin_bit, end_bit, tmp, s, n_dev);
}

int find_ranges_u64_dev_launch(const uint64_t* sorted_keys, int64_t capacity, const int32_t* k_dev, int64_t num_tiles,
                               int32_t* out_ranges, hipStream_t s) {
  if (capacity > 0)
    find_ranges2_u64_kernel<<<dim3((unsigned)div_up(capacity, 512)), dim3(256), 0, s>>>(sorted_keys, capacity, k_dev, num_tiles, out_ranges);
  return 0;
}

}  // namespace ms

using namespace ms;

extern "C" int ms_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* total_host,
                                     void* tmp, size_t* tmp_bytes, void* stream) {
  MS_CHECK_ARG(n >= 0, "n < 0");
  MS_CHECK_ARG(tmp_bytes != nullptr, "tmp_bytes is null");
  const size_t need = scan_tmp_bytes(n);
  if (tmp == nullptr) { *tmp_bytes = need; return 0; }
  if (*tmp_bytes < need) { set_error("ms_exclusive_scan_i32: tmp too small (%zu < %zu)", *tmp_bytes, need); return MS_ERR_TMP_TOO_SMALL; }
  MS_CHECK_ARG(out != nullptr, "out is null");
  MS_CHECK_ARG(n == 0 || in != nullptr, "in is null");
  if (n == 0) {
    MS_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), (hipStream_t)stream));
    if (total_host) *total_host = 0;
    return 0;
  }
  exclusive_scan_i32(in, n, out, total_host, tmp, (hipStream_t)stream);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_radix_sort_pairs(const void* keys_in, const int32_t* values_in, void* keys_out,
                                   int32_t* values_out, int64_t n, int key_bytes, int begin_bit,
                                   int end_bit, void* tmp, size_t* tmp_bytes, void* stream) {
  MS_CHECK_ARG(n >= 0, "n < 0");
  MS_CHECK_ARG(key_bytes == 4 || key_bytes == 8, "key_bytes must be 4 or 8");
  MS_CHECK_ARG(begin_bit >= 0 && end_bit >= begin_bit && end_bit <= key_bytes * 8, "bad bit range");
  MS_CHECK_ARG(tmp_bytes != nullptr, "tmp_bytes is null");
  const size_t need = sort_tmp_layout(n, key_bytes).total;
  if (tmp == nullptr) { *tmp_bytes = need; return 0; }
  if (*tmp_bytes < need) { set_error("ms_radix_sort_pairs: tmp too small (%zu < %zu)", *tmp_bytes, need); return MS_ERR_TMP_TOO_SMALL; }
  if (n == 0) return 0;
  MS_CHECK_ARG(keys_in && values_in && keys_out && values_out, "null pointer");
  if (key_bytes == 4)
    radix_sort_pairs_impl<uint32_t>((const uint32_t*)keys_in, values_in, (uint32_t*)keys_out, values_out, n, begin_bit, end_bit, (char*)tmp, (hipStream_t)stream);
  else
    radix_sort_pairs_impl<uint64_t>((const uint64_t*)keys_in, values_in, (uint64_t*)keys_out, values_out, n, begin_bit, end_bit, (char*)tmp, (hipStream_t)stream);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_depth_argsort(const void* depth, int64_t v, int depth16, double ndc_near, double ndc_far,
                               int dtype, uint32_t* out_sorted_keys, int32_t* out_order, void* tmp,
                               size_t* tmp_bytes, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(dtype == MS_F32 || dtype == MS_F64, "dtype must be MS_F32 or MS_F64");
  MS_CHECK_ARG(tmp_bytes != nullptr, "tmp_bytes is null");
  const size_t need = sort_tmp_layout(v, 4).total;
  if (tmp == nullptr) { *tmp_bytes = need; return 0; }
  if (*tmp_bytes < need) { set_error("ms_depth_argsort: tmp too small (%zu < %zu)", *tmp_bytes, need); return MS_ERR_TMP_TOO_SMALL; }
  if (v == 0) return 0;
  MS_CHECK_ARG(depth && out_sorted_keys && out_order, "null pointer");
  const int end_bit = depth16 ? 16 : 32;
  if (dtype == MS_F32)
    radix_sort_passes<uint32_t>(DepthPairs<float>{(const float*)depth, depth16, ndc_near, ndc_far, 0}, out_sorted_keys, out_order, v, 0, end_bit, (char*)tmp, (hipStream_t)stream);
  else
    radix_sort_passes<uint32_t>(DepthPairs<double>{(const double*)depth, depth16, ndc_near, ndc_far, 0}, out_sorted_keys, out_order, v, 0, end_bit, (char*)tmp, (hipStream_t)stream);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_segmented_sort_pairs(const int32_t* keys_in, const int32_t* values_in, int32_t* keys_out,
                                       int32_t* values_out, int64_t n, const int64_t* start_offsets,
                                       const int64_t* end_offsets, int64_t num_segments, void* stream) {
  MS_CHECK_ARG(n >= 0 && num_segments >= 0, "negative size");
  if (n == 0 || num_segments == 0) return 0;
  MS_CHECK_ARG(keys_in && values_in && keys_out && values_out && start_offsets && end_offsets, "null pointer");
  // elements outside every segment are copied through unchanged (CUB leaves them unspecified)
  copy_pairs_kernel<int32_t><<<dim3((unsigned)div_up(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(
      keys_in, values_in, keys_out, values_out, n);
  segmented_rank_sort_kernel<<<dim3((unsigned)num_segments), dim3(256), 0, (hipStream_t)stream>>>(
      keys_in, values_in, keys_out, values_out, start_offsets, end_offsets);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_find_ranges(const void* sorted_keys, int64_t k, int key_bytes, int tile_shift,
                              int64_t num_tiles, int32_t* out_ranges, void* stream) {
  MS_CHECK_ARG(k >= 0 && num_tiles >= 0, "negative size");
  MS_CHECK_ARG(key_bytes == 4 || key_bytes == 8, "key_bytes must be 4 or 8");
  MS_CHECK_ARG(out_ranges != nullptr || num_tiles == 0, "out_ranges is null");
  if (num_tiles > 0)
    MS_CHECK_HIP(hipMemsetAsync(out_ranges, 0, (size_t)num_tiles * 2 * sizeof(int32_t), (hipStream_t)stream));
  if (k == 0) return 0;
  MS_CHECK_ARG(sorted_keys != nullptr, "sorted_keys is null");
  const dim3 block(256), grid((unsigned)div_up(k, 256));
  if (key_bytes == 4)
    find_ranges_kernel<uint32_t><<<grid, block, 0, (hipStream_t)stream>>>((const uint32_t*)sorted_keys, k, nullptr, tile_shift, num_tiles, out_ranges);
  else
    find_ranges_kernel<uint64_t><<<grid, block, 0, (hipStream_t)stream>>>((const uint64_t*)sorted_keys, k, nullptr, tile_shift, num_tiles, out_ranges);
  MS_CHECK_LAUNCH();
  return 0;
}
