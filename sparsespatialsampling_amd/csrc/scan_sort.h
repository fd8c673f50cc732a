// Device-wide exclusive scan and stable LSD radix sort of (key, value) pairs, hand-written for gfx950 (64-wide wavefronts).
// Used by the planner (csrc/plan_build.hip: Hilbert order of the cells, offsets of the packed tiles, compaction of the referenced
// rows) and by the device topology (csrc/topo_dev.hip: node numbering of a refine batch, the finished grid's renumbering).
// Both are HBM-bound integer passes far off the roofline kernels; they replace the hipCUB calls of rounds 1-2.
//
//   exclusive_scan<T>(in, out, n, tmp, stream)      out[i] = in[0] + .. + in[i-1]; in == out allowed; tmp: scan_tmp_items(n) x T
//     n <= 16 tiles of 4096: ONE workgroup walks the tiles with a carry.  Otherwise three launches: tile sums, scan of the sums
//     (that one workgroup), tiles again with their offsets -- 2 reads + 1 write of the array.
//   radix_sort_pairs(keys, keys_alt, vals, vals_alt, n, bits, hist, stream) -> which buffer holds the result
//     8-bit digits from the least significant one, ceil(bits / 8) passes, each: digit counts per tile of 2048 keys
//     (digit-major table) -> exclusive scan of the table -> scatter.  Stable: a wavefront ranks the 64 keys of a chunk among
//     themselves by matching digits with eight ballots, walks its 512 keys chunk by chunk, and the wavefronts of a tile / the
//     tiles of the array are ordered through the scanned counts.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace s3 {

constexpr int SCAN_BLOCK = 1024, SCAN_ITEMS = 4, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS, SCAN_SINGLE_TILES = 16;

inline int64_t scan_tiles(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
inline size_t scan_tmp_items(int64_t n) { return (size_t)scan_tiles(n) + 1; }

// exclusive scan of one value per thread over the 1024 threads of a workgroup; total = sum of all (sh: 17 values)
template <typename T>
__device__ __forceinline__ T block_exclusive_scan_1024(T v, T &total, T *sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T u = __shfl_up(incl, d, 64);
        if (lane >= d) incl += u;
    }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        T w = lane < SCAN_BLOCK / 64 ? sh[lane] : T(0), wi = w;
#pragma unroll
        for (int d = 1; d < SCAN_BLOCK / 64; d <<= 1) {
            const T u = __shfl_up(wi, d, 64);
            if (lane >= d) wi += u;
        }
        if (lane < SCAN_BLOCK / 64) sh[lane] = wi - w;                  // exclusive over the wavefronts
        if (lane == SCAN_BLOCK / 64 - 1) sh[SCAN_BLOCK / 64] = wi;      // total
    }
    __syncthreads();
    const T r = sh[wave] + incl - v;
    total = sh[SCAN_BLOCK / 64];
    __syncthreads();                                                   // (sh is reused by the caller's next tile)
    return r;
}

// one tile: thread t holds items t * ITEMS .. of the tile (a 16 / 32-byte run per thread: coalesced enough for a pass that is
// a rounding error next to the kernels it serves)
template <typename T>
__device__ __forceinline__ void scan_tile(const T *__restrict__ in, T *__restrict__ out, int64_t base, int64_t n, T carry, T &total,
                                          T *sh) {
    T v[SCAN_ITEMS], s = T(0);
    const int64_t at = base + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = at + i < n ? in[at + i] : T(0);
        s += v[i];
    }
    T run = carry + block_exclusive_scan_1024(s, total, sh);
    if (out != nullptr) {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            if (at + i < n) out[at + i] = run;
            run += v[i];
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_single_kernel(const T *in, T *out, int64_t n) {                   // (in == out allowed: a tile is read before it is written)
    __shared__ T sh[SCAN_BLOCK / 64 + 1];
    T carry = T(0);
    for (int64_t base = 0; base < n; base += SCAN_TILE) {
        T total;
        scan_tile<T>(in, out, base, n, carry, total, sh);
        carry += total;
    }
}

template <typename T>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_sums_kernel(const T *__restrict__ in, int64_t n, T *__restrict__ sums) {
    __shared__ T sh[SCAN_BLOCK / 64 + 1];
    T total;
    scan_tile<T>(in, nullptr, (int64_t)blockIdx.x * SCAN_TILE, n, T(0), total, sh);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

template <typename T>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_apply_kernel(const T *in, T *out, int64_t n, const T *__restrict__ offsets) {
    __shared__ T sh[SCAN_BLOCK / 64 + 1];
    T total;
    scan_tile<T>(in, out, (int64_t)blockIdx.x * SCAN_TILE, n, offsets[blockIdx.x], total, sh);
}

template <typename T>
inline hipError_t exclusive_scan(const T *in, T *out, int64_t n, T *tmp, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t tiles = scan_tiles(n);
    if (tiles <= SCAN_SINGLE_TILES) {
        scan_single_kernel<T><<<1, SCAN_BLOCK, 0, st>>>(in, out, n);
        return hipGetLastError();
    }
    scan_sums_kernel<T><<<(unsigned)tiles, SCAN_BLOCK, 0, st>>>(in, n, tmp);
    scan_single_kernel<T><<<1, SCAN_BLOCK, 0, st>>>(tmp, tmp, tiles);
    scan_apply_kernel<T><<<(unsigned)tiles, SCAN_BLOCK, 0, st>>>(in, out, n, tmp);
    return hipGetLastError();
}

// ---- radix sort -----------------------------------------------------------------------------------------------------
constexpr int SORT_BLOCK = 256, SORT_WAVES = SORT_BLOCK / 64, SORT_PER_LANE = 8, SORT_TILE = SORT_BLOCK * SORT_PER_LANE, SORT_DIGITS = 256;

inline int64_t sort_tiles(int64_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
inline size_t sort_hist_items(int64_t n) { return (size_t)SORT_DIGITS * (size_t)sort_tiles(n); }       // int32 counts, digit-major

// lanes of the wavefront whose digit equals this lane's (invalid lanes match nobody but themselves)
__device__ __forceinline__ uint64_t match_digit(uint32_t digit, bool valid) {
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const uint64_t has = __ballot((digit >> b) & 1u);
        m &= ((digit >> b) & 1u) ? has : ~has;
    }
    return valid ? m : 0ull;
}

__device__ __forceinline__ void sort_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// wavefront w of the tile owns keys w * 512 .. w * 512 + 511 of it, chunk c of the wavefront = 64 consecutive keys
__device__ __forceinline__ int64_t sort_slot(int64_t tile, int wave, int chunk, int lane) {
    return tile * SORT_TILE + (int64_t)wave * (64 * SORT_PER_LANE) + chunk * 64 + lane;
}

// counts of the tile's digits -> hist[digit * n_tiles + tile]
static __global__ void __launch_bounds__(SORT_BLOCK)
radix_hist_kernel(const uint64_t *__restrict__ keys, int64_t n, int shift, uint32_t mask, int64_t n_tiles, int32_t *__restrict__ hist) {
    __shared__ int32_t cnt[SORT_DIGITS];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < SORT_PER_LANE; ++c) {
        const int64_t i = sort_slot(blockIdx.x, wave, c, lane);
        const bool valid = i < n;
        const uint32_t digit = valid ? (uint32_t)(keys[i] >> shift) & mask : 0u;
        const uint64_t m = match_digit(digit, valid);
        if (valid && (m & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&cnt[digit], __popcll(m));     // the lowest lane of each digit
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * n_tiles + blockIdx.x] = cnt[threadIdx.x];
}

// offsets = exclusive scan of hist: where the tile's keys of each digit start in the output
static __global__ void __launch_bounds__(SORT_BLOCK)
radix_scatter_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals, int64_t n, int shift, uint32_t mask, int64_t n_tiles,
                     const int32_t *__restrict__ offsets, uint64_t *__restrict__ keys_out, int32_t *__restrict__ vals_out) {
    __shared__ int32_t run[SORT_WAVES][SORT_DIGITS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) run[w][threadIdx.x] = 0;
    __syncthreads();
    uint64_t key[SORT_PER_LANE];
    int32_t val[SORT_PER_LANE];
    // the wavefront's digit counts
#pragma unroll
    for (int c = 0; c < SORT_PER_LANE; ++c) {
        const int64_t i = sort_slot(blockIdx.x, wave, c, lane);
        const bool valid = i < n;
        key[c] = valid ? keys[i] : 0ull;
        val[c] = valid ? vals[i] : 0;
        const uint32_t digit = (uint32_t)(key[c] >> shift) & mask;
        const uint64_t m = match_digit(digit, valid);
        if (valid && (m & ((1ull << lane) - 1ull)) == 0ull) run[wave][digit] += __popcll(m);          // one lane per digit and chunk
        sort_wave_sync();
    }
    __syncthreads();
    {   // digit = threadIdx.x: start of every wavefront's keys of this digit
        int32_t at = offsets[(int64_t)threadIdx.x * n_tiles + blockIdx.x];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const int32_t c = run[w][threadIdx.x];
            run[w][threadIdx.x] = at;
            at += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < SORT_PER_LANE; ++c) {
        const int64_t i = sort_slot(blockIdx.x, wave, c, lane);
        const bool valid = i < n;
        const uint32_t digit = (uint32_t)(key[c] >> shift) & mask;
        const uint64_t m = match_digit(digit, valid);
        const uint64_t below = m & ((1ull << lane) - 1ull);
        if (valid) {
            const int64_t dst = (int64_t)run[wave][digit] + __popcll(below);
            keys_out[dst] = key[c];
            vals_out[dst] = val[c];
        }
        sort_wave_sync();                                              // every lane has read run[] before the leaders move it on
        if (valid && below == 0ull) run[wave][digit] += __popcll(m);
        sort_wave_sync();
    }
}

// sorts (keys, vals) by the low `bits` bits of the keys, ascending, stable.  keys_alt / vals_alt: buffers of the same size;
// hist: sort_hist_items(n) + scan_tmp_items(sort_hist_items(n)) int32.  *in_alt = the result is in the alt buffers.
inline hipError_t radix_sort_pairs(uint64_t *keys, uint64_t *keys_alt, int32_t *vals, int32_t *vals_alt, int64_t n, int bits,
                                   int32_t *hist, hipStream_t st, bool *in_alt) {
    *in_alt = false;
    if (n <= 0) return hipSuccess;
    const int64_t tiles = sort_tiles(n);
    const int64_t items = (int64_t)sort_hist_items(n);
    int32_t *scan_tmp = hist + items;
    bool alt = false;
    for (int shift = 0; shift < bits; shift += 8) {
        const uint64_t *kin = alt ? keys_alt : keys;
        const int32_t *vin = alt ? vals_alt : vals;
        const uint32_t mask = bits - shift >= 8 ? 255u : (1u << (bits - shift)) - 1u;        // (the last digit may be narrower)
        radix_hist_kernel<<<(unsigned)tiles, SORT_BLOCK, 0, st>>>(kin, n, shift, mask, tiles, hist);
        hipError_t e = exclusive_scan<int32_t>(hist, hist, items, scan_tmp, st);
        if (e != hipSuccess) return e;
        radix_scatter_kernel<<<(unsigned)tiles, SORT_BLOCK, 0, st>>>(kin, vin, n, shift, mask, tiles, hist, alt ? keys : keys_alt,
                                                                   alt ? vals : vals_alt);
        alt = !alt;
    }
    *in_alt = alt;
    return hipGetLastError();
}

}  // namespace s3
