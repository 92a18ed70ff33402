// k_binning.hip — scan, per-tile scatter and per-tile depth sort.  Together they replace the
// InclusiveSum -> duplicateWithKeys -> global 64-bit radix sort -> identifyTileRanges chain of
// CudaRasterizer::Rasterizer::forward (reference call site src/Trainer.cu:334-360; SURVEY.md
// Appendix A.2-A.5) and produce the identical per-tile ordered lists:
//   upstream sorts (tile << 32 | depth_bits) stably, ties keep emission order = ascending splat id;
//   here every entry gets the slot its upstream emission position would have had
//   (slot = point_offsets[i-1] + k, k-th tile of splat i in y-outer/x-inner order), entries are
//   scattered into their tile's segment in arbitrary (atomic) order, and each tile is sorted on
//   the unique 64-bit key (depth_bits << 32 | slot).  slot is monotone in splat id, so the result
//   equals the stable global sort, deterministically, without any global multi-pass sort.
// MI355X: a tile's list lives in LDS (160 KB/CU) for the whole sort; integer arithmetic only.
#include "gs_internal.h"

namespace gs {

// ---------------------------------------------------------------------------------------------
// batched inclusive scan of u32 (three phases; block = 256 threads x 16 items)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_BLOCK = WG * SCAN_ITEMS;

size_t scan_partials_count(int n, int batch) { return (size_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK) * batch; }

__device__ inline uint32_t wave_incl_scan(uint32_t x) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    return x;
}

// exclusive prefix of `x` over the 256 threads of the block; total returned through *total
__device__ inline uint32_t block_excl_scan(uint32_t x, uint32_t* total) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(x);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < w; k++) base += wsum[k];
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - x;
}

__global__ __launch_bounds__(WG) void k_scan_reduce(const uint32_t* __restrict__ in, int n, int stride, uint32_t* partials) {
    const int nb = gridDim.x, b = blockIdx.y;
    const uint32_t* src = in + (size_t)b * stride;
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) sum += src[base + k];
    uint32_t total;
    block_excl_scan(sum, &total);
    if (threadIdx.x == 0) partials[(size_t)b * nb + blockIdx.x] = total;
}

__global__ __launch_bounds__(WG) void k_scan_partials(uint32_t* partials, int nb) {
    uint32_t* p = partials + (size_t)blockIdx.x * nb;
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += WG) {
        const int idx = base + threadIdx.x;
        const uint32_t x = idx < nb ? p[idx] : 0;
        uint32_t total;
        const uint32_t ex = block_excl_scan(x, &total);
        if (idx < nb) p[idx] = carry + ex;  // exclusive prefix of block sums
        carry += total;
    }
}

__global__ __launch_bounds__(WG) void k_scan_final(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n,
                                                   int stride, const uint32_t* __restrict__ partials) {
    const int nb = gridDim.x, b = blockIdx.y;
    const uint32_t* src = in + (size_t)b * stride;
    uint32_t* dst = out + (size_t)b * stride;
    const int base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (base + k < n) ? src[base + k] : 0; sum += v[k]; }
    uint32_t run = block_excl_scan(sum, nullptr) + partials[(size_t)b * nb + blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { run += v[k]; if (base + k < n) dst[base + k] = run; }
}

int launch_scan_u32(const uint32_t* in, uint32_t* out, int n, int stride, int batch, uint32_t* partials, hipStream_t st) {
    if (n == 0 || batch == 0) return GS_OK;
    const int nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb, batch), dim3(WG), 0, st, in, n, stride, partials);
    hipLaunchKernelGGL(k_scan_partials, dim3(batch), dim3(WG), 0, st, partials, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb, batch), dim3(WG), 0, st, in, out, n, stride, (const uint32_t*)partials);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// scatter: one thread per (view, splat); entry k of splat i -> slot, tile segment position by atomic
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_scatter(Dims d, Scratch s) {
    const int i = blockIdx.x * WG + threadIdx.x;
    const int v = blockIdx.y;
    if (i >= d.P) return;
    const size_t pv = (size_t)v * d.Pa;
    const uint32_t tiles = s.tiles_touched[pv + i];
    if (i == d.P - 1) s.flags[v * 4 + 2] = s.point_offsets[pv + i];  // num_rendered of this view
    if (tiles == 0) return;
    const GeomRec* rec = s.geom + pv + i;
    const uint32_t rmin = rec->rect_min, rmax = rec->rect_max;
    const uint32_t depth = __float_as_uint(rec->depth);
    uint32_t slot = s.point_offsets[pv + i] - tiles;
    const uint32_t total = s.point_offsets[pv + d.P - 1];
    if (total > d.Rcap) {  // arena too small for this view: flag it, the host grows and replays
        if ((threadIdx.x & 63) == 0 || slot == 0) atomicOr(&s.flags[v * 4 + 0], 1u);
        return;
    }
    const uint32_t* tend = s.tile_end + (size_t)v * d.T;
    const uint32_t* tcnt = s.tile_count + (size_t)v * d.T;
    uint32_t* cur = s.tile_cursor + (size_t)v * d.T;
    uint64_t* bins = s.bins + (size_t)v * d.Rcap;
    uint32_t* ids = s.id_of_slot + (size_t)v * d.Rcap;
    const int x0 = rmin & 0xffff, y0 = rmin >> 16, x1 = rmax & 0xffff, y1 = rmax >> 16;
    for (int ty = y0; ty < y1; ty++)
        for (int tx = x0; tx < x1; tx++) {
            const int tile = ty * d.gx + tx;
            const uint32_t pos = atomicAdd(&cur[tile], 1u);
            const uint32_t start = tend[tile] - tcnt[tile];
            bins[start + pos] = ((uint64_t)depth << 32) | slot;
            ids[slot] = (uint32_t)i;
            slot++;
        }
}

int launch_scatter(const Dims& d, const Scratch& s, hipStream_t st) {
    if (d.P == 0 || d.V == 0) return GS_OK;
    hipLaunchKernelGGL(k_scatter, dim3((d.P + WG - 1) / WG, d.V), dim3(WG), 0, st, d, s);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// per-tile sort on the unique 64-bit key: counting-rank sort in LDS for n <= 512, bitonic sort in LDS for
// n <= SORT_LDS_CAP, bitonic in global scratch (the not yet used gradient-slot buffer G) for longer lists —
// the "tile-list spill path".
// ---------------------------------------------------------------------------------------------
__device__ inline void bitonic_sort(uint64_t* a, uint32_t n2) {
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (n2 >> 1); t += WG) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t l = i | j;
                const bool up = ((i & k) == 0);
                const uint64_t x = a[i], y = a[l];
                if ((x > y) == up) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(WG) void k_tile_sort(Dims d, Scratch s) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t* sk = reinterpret_cast<uint64_t*>(smem_raw);
    const int tile = blockIdx.x, v = blockIdx.y;
    if (s.flags[v * 4 + 0] & 1u) return;  // overflowed view: nothing was scattered
    const uint32_t n = s.tile_count[(size_t)v * d.T + tile];
    if (n == 0) return;
    const uint32_t start = s.tile_end[(size_t)v * d.T + tile] - n;
    uint64_t* bins = s.bins + (size_t)v * d.Rcap + start;
    const uint32_t* ids = s.id_of_slot + (size_t)v * d.Rcap;
    uint32_t* pl = s.point_list + (size_t)v * d.Rcap + start;
    uint32_t* sl = s.slot_list + (size_t)v * d.Rcap + start;
    if (threadIdx.x == 0) atomicMax(&s.flags[v * 4 + 1], n);
    constexpr uint32_t RANK_MAX = 512;
    if (n <= RANK_MAX) {
        // Short lists (the common case: a few hundred entries): counting-rank sort.  Keys are unique, so
        // rank = #keys smaller is the final position; every thread ranks its key against the whole list with
        // broadcast 16-byte LDS reads.  Two barriers in total instead of one per bitonic step.
        const uint32_t ne = (n + 1) & ~1u;
        for (uint32_t t = threadIdx.x; t < ne; t += WG) sk[t] = t < n ? bins[t] : ~0ull;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < n; t += WG) {
            const uint64_t mine = sk[t];
            uint32_t rank = 0;
            const ulonglong2* pairs = reinterpret_cast<const ulonglong2*>(sk);
            for (uint32_t j = 0; j < ne / 2; j++) {
                const ulonglong2 kk = pairs[j];
                rank += (kk.x < mine) ? 1u : 0u;
                rank += (kk.y < mine) ? 1u : 0u;
            }
            const uint32_t slot = (uint32_t)mine;
            sl[rank] = slot;
            pl[rank] = ids[slot];
        }
        return;
    }
    uint32_t n2 = 2;
    while (n2 < n) n2 <<= 1;
    uint64_t* a;
    if (n <= (uint32_t)SORT_LDS_CAP) a = sk;
    else a = reinterpret_cast<uint64_t*>(s.G + (size_t)v * d.Rcap * G_STRIDE) + 2 * (size_t)start;  // n2 < 2n entries
    for (uint32_t t = threadIdx.x; t < n2; t += WG) a[t] = t < n ? bins[t] : ~0ull;
    __syncthreads();
    bitonic_sort(a, n2);
    for (uint32_t t = threadIdx.x; t < n; t += WG) {
        const uint32_t slot = (uint32_t)a[t];
        sl[t] = slot;
        pl[t] = ids[slot];
    }
}

int launch_tile_sort(const Dims& d, const Scratch& s, hipStream_t st) {
    if (d.T == 0 || d.V == 0) return GS_OK;
    hipLaunchKernelGGL(k_tile_sort, dim3(d.T, d.V), dim3(WG), SORT_LDS_CAP * sizeof(uint64_t), st, d, s);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ranges[tile] = (first, one-past-last), (0,0) for empty tiles — upstream's identifyTileRanges output
__global__ void k_ranges(Dims d, Scratch s, uint32_t* ranges) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.T * d.V) return;
    const uint32_t n = s.tile_count[t], e = s.tile_end[t];
    ranges[2 * (size_t)t] = n ? e - n : 0;
    ranges[2 * (size_t)t + 1] = n ? e : 0;
}

int launch_ranges(const Dims& d, const Scratch& s, uint32_t* ranges, hipStream_t st) {
    const int n = d.T * d.V;
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(k_ranges, dim3((n + 255) / 256), dim3(256), 0, st, d, s, ranges);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
