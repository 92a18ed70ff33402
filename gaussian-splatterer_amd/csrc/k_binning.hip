// k_binning.hip — scans, coarse (super-tile) binning, per-tile scatter + depth sort.  Together they replace
// the InclusiveSum -> duplicateWithKeys -> global 64-bit radix sort -> identifyTileRanges chain of
// CudaRasterizer::Rasterizer::forward (reference call site src/Trainer.cu:334-360; SURVEY.md
// Appendix A.2-A.5) and produce the identical per-tile ordered lists:
//   upstream sorts (tile << 32 | depth_bits) stably, ties keep emission order = ascending splat id;
//   here every (splat, tile) entry is identified by the slot its upstream emission position would have had
//   (slot = point_offsets[i-1] + k, k-th tile of splat i in y-outer/x-inner order) and each tile is sorted
//   on the unique 64-bit key (depth_bits << 32 | slot).  slot is monotone in splat id, so the result equals
//   the stable global sort, deterministically, without any global multi-pass sort.
// MI355X: global atomics are the scarce resource (~10-20 G/s scattered, ~70 ns each on one address; they execute at
// the memory side), so this file uses NONE.  Splats are binned at 64x64-px super-tile granularity through a
// (256-splat block x super-tile) count matrix (LDS histograms in k_preprocess, column scan here, LDS cursors in
// the scatter); one workgroup per super-tile then counts and fills the segments of its 16 tiles with LDS
// atomics, and every tile sorts its own segment in LDS (160 KB LDS/CU).  Tiles are taken longest list first.
#include "gs_internal.h"

namespace gs {

// ---------------------------------------------------------------------------------------------
// batched inclusive scans of u32 (block = 256 threads x 16 items)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_BLOCK = WG * SCAN_ITEMS;

size_t scan_partials_count(int n, int batch) { return (size_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK) * batch; }

// Dynamic LDS beyond the 64 KB a kernel gets by default has to be asked for (gfx950: 160 KB per CU).  The largest request here is
// MAX_SUPER_TILES counters = 64 KB + the kernel's few static words.
int allow_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return GS_OK;
    GS_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return GS_OK;
}

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

// One workgroup scans one batch entry, SCAN_BLOCK items per trip with a running carry.  The scans of a step are short
// (super-tile counters, tile counts: 4096 items per view at 1024x1024), so a single launch per scan beats the
// three-phase form, whose three launches cost more than the work.
__device__ inline void scan_single(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n) {
    uint32_t carry = 0;
    for (int base0 = 0; base0 < n; base0 += SCAN_BLOCK) {
        const int base = base0 + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS];
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (base + k < n) ? src[base + k] : 0; sum += v[k]; }
        uint32_t total;
        uint32_t run = carry + block_excl_scan(sum, &total);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) { run += v[k]; if (base + k < n) dst[base + k] = run; }
        carry += total;
    }
}

__global__ __launch_bounds__(WG) void k_scan_single(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n, int stride) {
    scan_single(in + (size_t)blockIdx.x * stride, out + (size_t)blockIdx.x * stride, n);
}

int g_scan_single_max = 1 << 16;

int launch_scan_u32(const uint32_t* in, uint32_t* out, int n, int stride, int batch, uint32_t* partials, hipStream_t st) {
    if (n == 0 || batch == 0) return GS_OK;
    if (n <= g_scan_single_max) {
        hipLaunchKernelGGL(k_scan_single, dim3(batch), dim3(WG), 0, st, in, out, n, stride);
    } else {
        const int nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
        hipLaunchKernelGGL(k_scan_reduce, dim3(nb, batch), dim3(WG), 0, st, in, n, stride, partials);
        hipLaunchKernelGGL(k_scan_partials, dim3(batch), dim3(WG), 0, st, partials, nb);
        hipLaunchKernelGGL(k_scan_final, dim3(nb, batch), dim3(WG), 0, st, in, out, n, stride, (const uint32_t*)partials);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// per-block tile sums -> exclusive prefixes; entry count and overflow verdict of the group (every flag word is rewritten each
// step: nothing to clear beforehand)
__device__ inline void block_sums_prefix(const Dims& d, const Scratch& s, int v) {
    const int nb = (d.P + WG - 1) / WG;
    uint32_t* p = s.block_sums + (size_t)v * splat_blocks(d.Pa);
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += WG) {
        const int idx = base + threadIdx.x;
        const uint32_t x = idx < nb ? p[idx] : 0;
        uint32_t total;
        const uint32_t ex = block_excl_scan(x, &total);
        if (idx < nb) p[idx] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        s.flags[v * 4 + 2] = carry;  // num_rendered of this group
        s.flags[v * 4 + 0] = carry > d.Rcap ? 1u : 0u;  // arena too small: the later stages skip the group, the host grows and replays
        s.sort_marks[v * 2] = 0u; s.sort_marks[v * 2 + 1] = 0u;  // (flags[3], the host's hint of the last step's tile order, stays)
    }
}

// Column scan of the (splat block x super-tile) count matrix: one WAVE per super-tile walks the column over the
// blocks, replaces every count by the exclusive prefix of its column and writes the column total (the super-tile's
// candidate count).  With the scan of the totals this gives every block its private output range per super-tile:
// the scatter needs LDS cursors only.
__global__ __launch_bounds__(WG) void k_coarse_colscan(Dims d, Scratch s) {
    const int v = blockIdx.y;
    const int col_wgs = (d.NST + WG / 64 - 1) / (WG / 64);
    if ((int)blockIdx.x == col_wgs) {
        // The launch's extra workgroup: turns the per-block tile sums (k_preprocess) into exclusive prefixes and
        // publishes the group's entry count (flags[2]) and the arena-overflow bit (flags[0]).  It needs nothing of the
        // column scan, so it rides in the same launch (one dependent launch less per step).
        block_sums_prefix(d, s, v);
        return;
    }
    const int st = blockIdx.x * (WG / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (st >= d.NST) return;
    const int nb = (d.P + WG - 1) / WG;
    uint32_t* col = s.wg_hist + (size_t)v * splat_blocks(d.Pa) * d.NST + st;
    uint32_t carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 64) {
        const int b = b0 + lane;
        const uint32_t x = b < nb ? col[(size_t)b * d.NST] : 0u;
        const uint32_t inc = wave_incl_scan(x);
        if (b < nb) col[(size_t)b * d.NST] = carry + inc - x;
        carry += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) s.coarse_count[(size_t)v * d.NST + st] = carry;
}
// Column scan of the count matrix + (extra workgroup) the prefix of the block tile sums with the entry count and the
// overflow verdict.  The scan of the super-tile totals that used to be a launch of its own is done by the consumer
// (k_coarse_scatter): NST is at most MAX_SUPER_TILES, a few LDS scan trips per workgroup.
// Large models (>= COLSCAN_TWO_PASS_BLOCKS splat blocks): the one-wave-per-column walk above touches the matrix through a stride
// of NST words — one 64-byte line per lane and access, 0.23 ms for the 128 MB of four 1M-splat cameras.  Two coalesced passes
// instead: thread = column (256 consecutive super-tiles per workgroup), rows in chunks of COLSCAN_CHUNK blocks.
//   pass 1  sum of every (chunk, column)                                  -> partial[chunk][column]
//   pass 2  prefix of the chunk sums in front of the chunk, then the chunk's rows again, writing the exclusive prefix in
//           place; the last chunk also writes the column total.  The extra workgroup (block-sum prefix, entry count,
//           overflow verdict) rides in pass 2.
constexpr int COLSCAN_CHUNK = 64;
constexpr int COLSCAN_TWO_PASS_BLOCKS = 1024;
__global__ __launch_bounds__(WG) void k_colscan_partial(Dims d, Scratch s, uint32_t* __restrict__ partial, int nch) {
    const int col = blockIdx.x * WG + threadIdx.x, c = blockIdx.y, v = blockIdx.z;
    if (col >= d.NST) return;
    const int nb = (d.P + WG - 1) / WG;
    const uint32_t* m = s.wg_hist + (size_t)v * splat_blocks(d.Pa) * d.NST + col;
    uint32_t sum = 0;
    const int r1 = min(nb, (c + 1) * COLSCAN_CHUNK);
    for (int r = c * COLSCAN_CHUNK; r < r1; r++) sum += m[(size_t)r * d.NST];
    partial[((size_t)v * nch + c) * d.NST + col] = sum;
}
__global__ __launch_bounds__(WG) void k_colscan_apply(Dims d, Scratch s, const uint32_t* __restrict__ partial, int nch) {
    const int c = blockIdx.y, v = blockIdx.z;
    if (c == nch) { if (blockIdx.x == 0) block_sums_prefix(d, s, v); return; }
    const int col = blockIdx.x * WG + threadIdx.x;
    if (col >= d.NST) return;
    const int nb = (d.P + WG - 1) / WG;
    uint32_t run = 0;
    for (int k = 0; k < c; k++) run += partial[((size_t)v * nch + k) * d.NST + col];
    uint32_t* m = s.wg_hist + (size_t)v * splat_blocks(d.Pa) * d.NST + col;
    const int r1 = min(nb, (c + 1) * COLSCAN_CHUNK);
    for (int r = c * COLSCAN_CHUNK; r < r1; r++) {
        const uint32_t x = m[(size_t)r * d.NST];
        m[(size_t)r * d.NST] = run;
        run += x;
    }
    if (c == nch - 1) s.coarse_count[(size_t)v * d.NST + col] = run;
}

int launch_coarse_colscan(const Dims& d, const Scratch& s, hipStream_t st) {
    if (d.NST == 0 || d.VG == 0) return GS_OK;
    const int nb = (d.P + WG - 1) / WG;
    if (nb >= COLSCAN_TWO_PASS_BLOCKS && s.colscan_partial) {
        const int nch = (nb + COLSCAN_CHUNK - 1) / COLSCAN_CHUNK;
        hipLaunchKernelGGL(k_colscan_partial, dim3((d.NST + WG - 1) / WG, nch, d.VG), dim3(WG), 0, st, d, s, s.colscan_partial, nch);
        hipLaunchKernelGGL(k_colscan_apply, dim3((d.NST + WG - 1) / WG, nch + 1, d.VG), dim3(WG), 0, st, d, s, (const uint32_t*)s.colscan_partial, nch);
    } else {
        hipLaunchKernelGGL(k_coarse_colscan, dim3((d.NST + WG / 64 - 1) / (WG / 64) + 1, d.VG), dim3(WG), 0, st, d, s);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}
size_t colscan_partial_words(int Pa, int NST, int V) {
    const int nb = splat_blocks(Pa);
    return nb >= COLSCAN_TWO_PASS_BLOCKS ? (size_t)V * ((nb + COLSCAN_CHUNK - 1) / COLSCAN_CHUNK) * NST : 0;
}

// ---------------------------------------------------------------------------------------------
// coarse scatter: one thread per (view, splat); finishes the offsets scan, then one candidate record per super-tile
// the splat touches
// ---------------------------------------------------------------------------------------------
template <bool CUT>
__global__ __launch_bounds__(WG) void k_coarse_scatter(Dims d, Scratch s) {
    extern __shared__ uint32_t sm[];
    uint32_t* cur = sm;             // [NST] next output position of this block in every super-tile's list: starts at the block's
                                    //       first position there and is advanced by LDS atomics (one array: 64 KB at the 16384
                                    //       super-tiles of an 8192 x 8192 render, 16 KB at 2048 x 2048 — it bounds the occupancy)
    const int i = blockIdx.x * WG + threadIdx.x;
    const int v = blockIdx.y;
    const size_t pv = (size_t)v * d.Pa;
    {
        // exclusive scan of the super-tile totals (the start of every super-tile's candidate list), done by every
        // workgroup for itself in LDS; workgroup 0 also stores the inclusive scan for the per-tile kernels
        const uint32_t* ccnt = s.coarse_count + (size_t)v * d.NST;
        const uint32_t* row = s.wg_hist + ((size_t)v * splat_blocks(d.Pa) + blockIdx.x) * d.NST;
        uint32_t* cend = s.coarse_end + (size_t)v * d.NST;
        uint32_t carry = 0;
        for (int k0 = 0; k0 < d.NST; k0 += WG) {
            const int k = k0 + threadIdx.x;
            const uint32_t c = k < d.NST ? ccnt[k] : 0u;
            uint32_t total;
            const uint32_t ex = carry + block_excl_scan(c, &total);
            if (k < d.NST) {
                cur[k] = ex + row[k];
                if (blockIdx.x == 0) cend[k] = ex + c;
            }
            carry += total;
        }
    }
    const uint32_t tiles = i < d.P ? s.tiles_touched[pv + i] : 0u;
    // finish the offsets scan inside the block: exclusive prefix of the block (k_coarse_colscan's extra workgroup) + in-block scan
    const uint32_t slot_base = s.block_sums[(size_t)v * splat_blocks(d.Pa) + blockIdx.x] + block_excl_scan(tiles, nullptr);  // syncs: cur is ready
    if (i >= d.P) return;
    s.point_offsets[pv + i] = slot_base + tiles;
    if (tiles == 0 || (s.flags[v * 4 + 0] & 1u)) return;
    const GeomRec* rec = s.geom + pv + i;
    const uint32_t rmin = rec->rect_min, rmax = rec->rect_max;
    const uint32_t depth = __float_as_uint(rec->depth);
    uint4* list = s.coarse_list + (size_t)v * d.Rcap;
    uint32_t* dl = s.coarse_depth + (size_t)v * d.Rcap;
    const int x0 = rmin & 0xffff, y0 = rmin >> 16, x1 = rmax & 0xffff, y1 = rmax >> 16;
    const int sx0 = x0 / STILE, sx1 = (x1 - 1) / STILE + 1, sy0 = y0 / STILE, sy1 = (y1 - 1) / STILE + 1;
    for (int sy = sy0; sy < sy1; sy++)
        for (int sx = sx0; sx < sx1; sx++) {
            const int st = sy * d.sgx + sx;
            if (CUT && depth > s.stile_zcut[(size_t)v * d.NST + st]) continue;   // not counted by the projection either (k_preprocess)
            const uint32_t pos = atomicAdd(&cur[st], 1u);  // LDS: the block owns [its first position, + its count) of the list
            list[pos] = make_uint4((uint32_t)i, rmin, rmax, slot_base);
            dl[pos] = depth;
        }
}

int launch_coarse_scatter(const Dims& d, const Scratch& s, hipStream_t st) {
    if (d.P == 0 || d.VG == 0) return GS_OK;
    const size_t lds = (size_t)d.NST * sizeof(uint32_t);
    if (d.cut) {
        GS_TRY(allow_dynamic_lds((const void*)k_coarse_scatter<true>, lds));
        hipLaunchKernelGGL(k_coarse_scatter<true>, dim3((d.P + WG - 1) / WG, d.VG), dim3(WG), lds, st, d, s);
    } else {
        GS_TRY(allow_dynamic_lds((const void*)k_coarse_scatter<false>, lds));
        hipLaunchKernelGGL(k_coarse_scatter<false>, dim3((d.P + WG - 1) / WG, d.VG), dim3(WG), lds, st, d, s);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// per-tile entry counts: one workgroup per super-tile walks its candidates once, LDS atomics only
// ---------------------------------------------------------------------------------------------
#ifndef GS_WIDE_BIN_MAX_GROUPS
#define GS_WIDE_BIN_MAX_GROUPS 4  // launches of up to this many cameras use the 1024-thread form (tuning hook; 8 cameras: 256 is faster)
#endif
// NT threads per workgroup: 256, or 1024 for launches of up to four cameras (a super-tile's workgroup is then 16 waves wide: the few
// hundred workgroups of such a launch do not fill the chip, and its time is the latency of one workgroup's candidate walk)
template <int NT, bool CUT>
__global__ __launch_bounds__(NT) void k_tile_count(Dims d, Scratch s) {
    __shared__ uint32_t cnt[STILE * STILE];
    const int st = blockIdx.x, v = blockIdx.y;
    const int stx = st % d.sgx, sty = st / d.sgx;
    if (threadIdx.x < STILE * STILE) cnt[threadIdx.x] = 0;
    __syncthreads();
    if (!(s.flags[v * 4 + 0] & 1u)) {
        const size_t c0 = (size_t)v * d.NST + st;
        const uint32_t nc = s.coarse_count[c0];
        const uint32_t cstart = s.coarse_end[c0] - nc;
        const uint4* list = s.coarse_list + (size_t)v * d.Rcap + cstart;
        const int tx0 = stx * STILE, ty0 = sty * STILE;
        // A wave takes 64 consecutive candidates and asks, tile by tile, which of them cover the tile: the ballot's population
        // count is the wave's contribution — 16 scalar counters per wave and 16 LDS atomics at the end, instead of one
        // same-address LDS atomic per (candidate, tile) pair (which serialised: 27 M conflict cycles per launch at 1M splats).
        const int lane = threadIdx.x & 63;
        uint32_t mine = 0;  // lane tl < 16 accumulates the wave's count of tile tl
        if constexpr (CUT) {
            // Depth cut (Dims::cut): an entry behind the bound the previous step's forward left for the tile is not listed;
            // k_tile_scatter applies the same test.
            const uint32_t* dl = s.coarse_depth + (size_t)v * d.Rcap + cstart;
            uint32_t zc = 0xFFFFFFFFu;
            if (lane < STILE * STILE) {
                const int tx = tx0 + (lane % STILE), ty = ty0 + (lane / STILE);
                if (tx < d.gx && ty < d.gy) zc = s.tile_zcut[(size_t)v * d.T + ty * d.gx + tx];
            }
            for (uint32_t c0 = (threadIdx.x >> 6) * 64; c0 < nc; c0 += NT) {
                const uint32_t c = c0 + lane;
                int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
                uint32_t dz = 0;
                if (c < nc) {
                    const uint4 e = list[c];
                    dz = dl[c];
                    x0 = max((int)(e.y & 0xffff), tx0); x1 = min((int)(e.z & 0xffff), tx0 + STILE);
                    y0 = max((int)(e.y >> 16), ty0); y1 = min((int)(e.z >> 16), ty0 + STILE);
                }
#pragma unroll
                for (int tl = 0; tl < STILE * STILE; tl++) {
                    const int x = tx0 + (tl % STILE), y = ty0 + (tl / STILE);
                    const bool in = x >= x0 && x < x1 && y >= y0 && y < y1;
                    const unsigned long long mk = __ballot(in && dz <= (uint32_t)__builtin_amdgcn_readlane((int)zc, tl));
                    mine += (lane == tl) ? (uint32_t)__popcll(mk) : 0u;
                }
            }
            if (lane < STILE * STILE && mine) atomicAdd(&cnt[lane], mine);
        } else {
        for (uint32_t c0 = (threadIdx.x >> 6) * 64; c0 < nc; c0 += NT) {
            const uint32_t c = c0 + lane;
            int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
            if (c < nc) {
                const uint4 e = list[c];
                x0 = max((int)(e.y & 0xffff), tx0); x1 = min((int)(e.z & 0xffff), tx0 + STILE);
                y0 = max((int)(e.y >> 16), ty0); y1 = min((int)(e.z >> 16), ty0 + STILE);
            }
#pragma unroll
            for (int tl = 0; tl < STILE * STILE; tl++) {
                const int x = tx0 + (tl % STILE), y = ty0 + (tl / STILE);
                const unsigned long long m = __ballot(x >= x0 && x < x1 && y >= y0 && y < y1);
                mine += (lane == tl) ? (uint32_t)__popcll(m) : 0u;
            }
        }
        if (lane < STILE * STILE && mine) atomicAdd(&cnt[lane], mine);
        }
    }
    __syncthreads();
    if (threadIdx.x < STILE * STILE) {
        const int tx = stx * STILE + (threadIdx.x % STILE), ty = sty * STILE + (threadIdx.x / STILE);
        if (tx < d.gx && ty < d.gy) s.tile_count[(size_t)v * d.T + ty * d.gx + tx] = cnt[threadIdx.x];
    }
}

// Dims::cut, first launch of the step: the largest depth bound among a super-tile's tiles.  A candidate behind it lies behind the bound of
// every tile it could be listed in, so the projection's count matrix and the coarse scatter leave it out (a super-tile with one unbounded
// tile keeps everything).
__global__ __launch_bounds__(WG) void k_stile_zcut(Dims d, Scratch s) {
    const int st = blockIdx.x * WG + threadIdx.x, v = blockIdx.y;
    if (st >= d.NST) return;
    const int tx0 = (st % d.sgx) * STILE, ty0 = (st / d.sgx) * STILE;
    uint32_t z = 0;
    for (int ly = 0; ly < STILE; ly++)
        for (int lx = 0; lx < STILE; lx++)
            if (tx0 + lx < d.gx && ty0 + ly < d.gy) z = max(z, s.tile_zcut[(size_t)v * d.T + (ty0 + ly) * d.gx + tx0 + lx]);
    s.stile_zcut[(size_t)v * d.NST + st] = z;
}
int launch_stile_zcut(const Dims& d, const Scratch& s, hipStream_t st) {
    if (d.NST == 0 || d.VG == 0) return GS_OK;
    hipLaunchKernelGGL(k_stile_zcut, dim3((d.NST + WG - 1) / WG, d.VG), dim3(WG), 0, st, d, s);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

int launch_tile_count(const Dims& d, const Scratch& s, hipStream_t st) {
    if (d.NST == 0 || d.VG == 0) return GS_OK;
    if (d.cut) {
        if (d.VG <= GS_WIDE_BIN_MAX_GROUPS) hipLaunchKernelGGL((k_tile_count<1024, true>), dim3(d.NST, d.VG), dim3(1024), 0, st, d, s);
        else hipLaunchKernelGGL((k_tile_count<WG, true>), dim3(d.NST, d.VG), dim3(WG), 0, st, d, s);
    } else {
        if (d.VG <= GS_WIDE_BIN_MAX_GROUPS) hipLaunchKernelGGL((k_tile_count<1024, false>), dim3(d.NST, d.VG), dim3(1024), 0, st, d, s);
        else hipLaunchKernelGGL((k_tile_count<WG, false>), dim3(d.NST, d.VG), dim3(WG), 0, st, d, s);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// longest-list-first tile order.  The per-tile kernels take tiles in this order, so the heaviest workgroups start
// first and the tail of a launch is made of light ones (a one-camera launch is only ~3 waves of workgroups deep:
// row-major order leaves the CUs that drew a heavy tile last running alone).  Counting sort on min(count, 1023) / 4.
// ---------------------------------------------------------------------------------------------
constexpr int ORDER_BINS = 256;
// bin of a list length: 8-wide bins below 1024 entries, 64-wide bins from there to 9216 (bin boundaries fall on
// SORT_SMALL_CAP = 2048: "long" tiles, sorted by k_tile_sort_long, are exactly the bins >= ORDER_LONG_BIN)
__device__ inline int order_bin(uint32_t c) { return c < 1024u ? (int)(c >> 3) : min(ORDER_BINS - 1, 128 + (int)((c - 1024u) >> 6)); }
constexpr int ORDER_LONG_BIN = 128 + (SORT_SMALL_CAP - 1024) / 64;
static_assert(SORT_TINY_CAP % 8 == 0 && SORT_TINY_CAP <= 1024, "the short-list limit must be a bin boundary");
constexpr int ORDER_MID_BIN = SORT_TINY_CAP / 8;  // lists of SORT_TINY_CAP entries and more (k_tile_sort_mid's and k_tile_sort_long's)
// One workgroup per camera: tile_end = inclusive scan of tile_count (SCAN), tile_order, the longest list (flags[1]) and
// the lengths of the order's heads that hold all long / mid tiles (sort_marks; flags[3] tells the host about them).
//
// XCD affinity.  Workgroups are dealt round-robin over the chip's eight XCDs (position p of the order runs on XCD p % 8 —
// observed, not promised: it only ever matters for speed), and each XCD has its own L2.  With one global longest-first list the
// tiles that share a splat's record run on all eight XCDs and half of the record gathers of the render kernels miss L2.  So the
// 64x64-px super-tiles are dealt to XC = 8 classes (a hash of the super-tile's coordinates: every class is a fair sample of the
// image), every class gets its OWN longest-first list, and the lists are interleaved: position 8 k + x holds the k-th longest tile
// of class x (lists of unequal length are compacted, which only disturbs the tail).  An XCD then only ever touches the records
// of an eighth of the image, and the launch keeps its longest-first balance inside every XCD.
#ifndef GS_XCD_TILE_CLASSES
#define GS_XCD_TILE_CLASSES 8  // 1: one global longest-first list (tuning hook, tools/build_variant.sh)
#endif
#ifndef GS_XCD_BLOCK_W
#define GS_XCD_BLOCK_W STILE  // the blocks of tiles dealt to the classes, in tiles
#define GS_XCD_BLOCK_H STILE
#endif
constexpr int ORDER_CLASSES = GS_XCD_TILE_CLASSES;
__device__ inline int order_class(const Dims& d, int t) {
    if (ORDER_CLASSES == 1) return 0;
    const int bx = (t % d.gx) / GS_XCD_BLOCK_W, by = (t / d.gx) / GS_XCD_BLOCK_H;
    return (bx + 3 * by) % ORDER_CLASSES;
}
template <bool SCAN>
__device__ inline void tile_scan_order_body(const Dims& d, const Scratch& s, int v) {
    __shared__ uint32_t hist[ORDER_CLASSES][ORDER_BINS], start[ORDER_CLASSES][ORDER_BINS], csize[ORDER_CLASSES], clong[ORDER_CLASSES], cmid[ORDER_CLASSES];
    const uint32_t* cnt = s.tile_count + (size_t)v * d.T;
    uint32_t* order = s.tile_order + (size_t)v * d.T;
#pragma unroll
    for (int x = 0; x < ORDER_CLASSES; x++) hist[x][threadIdx.x] = 0;
    __syncthreads();
    // The counts are read ORDER_CHUNK at a time into registers — independent loads, one memory latency per chunk; one load per
    // trip made this workgroup a chain of T / 256 latencies, and at one or two cameras per launch (the per-GPU load of an 8-GPU
    // run) it is the longest workgroup of k_tile_scatter.
    constexpr int ORDER_CHUNK = 16;
    uint32_t longest = 0;
    for (int t0 = threadIdx.x; t0 < d.T; t0 += ORDER_CHUNK * WG) {
        uint32_t c[ORDER_CHUNK];
#pragma unroll
        for (int k = 0; k < ORDER_CHUNK; k++) c[k] = t0 + k * WG < d.T ? cnt[t0 + k * WG] : 0u;
#pragma unroll
        for (int k = 0; k < ORDER_CHUNK; k++) {
            const int t = t0 + k * WG;
            if (t < d.T) {
                longest = max(longest, c[k]);
                atomicAdd(&hist[order_class(d, t)][ORDER_BINS - 1 - order_bin(c[k])], 1u);  // bin 0 = longest
            }
        }
    }
    // statistic: the longest tile list of the group (a global atomicMax per TILE cost 210 us: same-address atomics serialise)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, o));
    __shared__ uint32_t wmax[WG / 64];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = longest;
    __syncthreads();
    if (threadIdx.x == 0) s.flags[v * 4 + 1] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    if (SCAN) scan_single(cnt, s.tile_end + (size_t)v * d.T, d.T);  // tile_end: the same workgroup has the counts in cache
    __syncthreads();
    for (int x = 0; x < ORDER_CLASSES; x++) {  // every class: exclusive scan of its bins, longest first
        uint32_t total;
        const uint32_t ex = block_excl_scan(hist[x][threadIdx.x], &total);
        start[x][threadIdx.x] = ex;
        if (threadIdx.x == ORDER_BINS - ORDER_LONG_BIN) clong[x] = ex;  // long tiles (k_tile_sort_long's) of the class: its first clong[x]
        if (threadIdx.x == ORDER_BINS - ORDER_MID_BIN) cmid[x] = ex;    // tiles of SORT_TINY_CAP entries and more
        if (threadIdx.x == 0) csize[x] = total;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // Every class's list is longest first, so a rank k of the interleaved order is reached at position
        // before(k) = sum over the classes of min(k, class size): all tiles of SORT_SMALL_CAP entries and more sit in front of
        // before(max clong), all of SORT_TINY_CAP and more in front of before(max cmid) — the heads k_tile_sort_long and
        // k_tile_sort_mid walk (skipping what is not theirs) — and nothing shorter sits in front of before(min cmid).
        uint32_t kl = 0, km = 0, ks = 0xFFFFFFFFu;
        for (int x = 0; x < ORDER_CLASSES; x++) { kl = max(kl, clong[x]); km = max(km, cmid[x]); ks = min(ks, cmid[x]); }
        uint32_t nl = 0, nm = 0, ns = 0;
        for (int x = 0; x < ORDER_CLASSES; x++) { nl += min(kl, csize[x]); nm += min(km, csize[x]); ns += min(ks, csize[x]); }
        s.sort_marks[v * 2] = nl;
        s.sort_marks[v * 2 + 1] = nm;
        const uint32_t unit = order_hint_unit(d.T);  // to the host, rounded to the safe side (capi.hip: launch grids of the next steps)
        s.flags[v * 4 + 3] = (min(0xFFFFu, (nm + unit - 1) / unit) << 16) | min(0xFFFFu, ns / unit);
    }
    uint32_t sizes[ORDER_CLASSES];
#pragma unroll
    for (int y = 0; y < ORDER_CLASSES; y++) sizes[y] = csize[y];
    for (int t0 = threadIdx.x; t0 < d.T; t0 += ORDER_CHUNK * WG) {
        uint32_t c[ORDER_CHUNK];
#pragma unroll
        for (int k = 0; k < ORDER_CHUNK; k++) c[k] = t0 + k * WG < d.T ? cnt[t0 + k * WG] : 0u;
#pragma unroll
        for (int j = 0; j < ORDER_CHUNK; j++) {
            const int t = t0 + j * WG;
            if (t < d.T) {
                const int x = order_class(d, t);
                const uint32_t k = atomicAdd(&start[x][ORDER_BINS - 1 - order_bin(c[j])], 1u);  // rank in the class's longest-first list
                uint32_t pos = 0;  // tiles in front of (k, x) in the interleaved order: ranks < k of every class, rank k of the classes before x
#pragma unroll
                for (int y = 0; y < ORDER_CLASSES; y++) pos += min(k, sizes[y]) + ((y < x && sizes[y] > k) ? 1u : 0u);
                order[pos] = (uint32_t)t;
            }
        }
    }
}
__global__ __launch_bounds__(WG) void k_tile_scan_order_noscan(Dims d, Scratch s) { tile_scan_order_body<false>(d, s, blockIdx.x); }
__global__ __launch_bounds__(WG) void k_tile_scan_order_scan(Dims d, Scratch s) { tile_scan_order_body<true>(d, s, blockIdx.x); }

// ---------------------------------------------------------------------------------------------
// per-tile scatter: the super-tile's workgroup walks its candidates a second time and appends every (candidate, tile)
// entry to the tile's segment [tile_end - tile_count, tile_end) — LDS cursors, no global atomics.  Entries land
// unsorted: key (depth_bits << 32 | slot) in the not yet used gradient-row buffer (u64 index 2 * start + pos: the
// segment doubles as the in-place area of the long-list bitonic sort), splat id in point_list.  The per-tile sort
// then reads exactly its own entries; before, every tile re-scanned all candidates of its super-tile (16 scans).
//
// SELF: the launch carries one extra workgroup per camera that does the tile scan + longest-first order
// (tile_scan_order_body) while the scatter workgroups run, and every scatter workgroup derives the starts of its own
// 16 tile segments from tile_count itself (the exclusive prefix at its four tile rows: one pass over the counts in
// front of it) instead of waiting for the scan as a launch of its own: two dependent launches less per step, and
// the single-workgroup scan (14 us) disappears behind the scatter.  !SELF (images with more tiles than one workgroup
// scans, or the "scan_single_max" test switch): tile_end comes from the separate scan launches in front.
// ---------------------------------------------------------------------------------------------
template <bool SELF, int NT, bool CUT>
__global__ __launch_bounds__(NT) void k_tile_scatter(Dims d, Scratch s) {
    __shared__ uint32_t cur[STILE * STILE], first[STILE * STILE], big[STILE * STILE];
    __shared__ uint32_t rowsum[STILE][NT / 64];
    const int st = blockIdx.x, v = blockIdx.y;
    if (SELF && st == d.NST) {  // the scan + order workgroup is written for 256 threads: the other waves of a wide workgroup leave
        if (threadIdx.x < WG) tile_scan_order_body<true>(d, s, v);
        return;
    }
    if (s.flags[v * 4 + 0] & 1u) return;
    const int stx = st % d.sgx, sty = st / d.sgx;
    const int tx0 = stx * STILE, ty0 = sty * STILE;
    const uint32_t* __restrict__ cnt = s.tile_count + (size_t)v * d.T;
    if (SELF) {
        // exclusive prefix of tile_count at the first tile of each of this super-tile's four tile rows
        int rs[STILE];
#pragma unroll
        for (int r = 0; r < STILE; r++) rs[r] = min(d.T, (ty0 + r) * d.gx + tx0);
        uint32_t acc[STILE] = { 0, 0, 0, 0 };
        for (int t = threadIdx.x; t < rs[STILE - 1]; t += NT) {
            const uint32_t c = cnt[t];
#pragma unroll
            for (int r = 0; r < STILE; r++) acc[r] += t < rs[r] ? c : 0u;
        }
#pragma unroll
        for (int r = 0; r < STILE; r++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[r] += (uint32_t)__shfl_xor((int)acc[r], o);
            if ((threadIdx.x & 63) == 0) rowsum[r][threadIdx.x >> 6] = acc[r];
        }
        __syncthreads();
    }
    if (threadIdx.x < STILE * STILE) {
        const int lx = threadIdx.x % STILE, ly = threadIdx.x / STILE;
        const int tx = tx0 + lx, ty = ty0 + ly;
        uint32_t n = 0, f = 0;
        if (tx < d.gx && ty < d.gy) {
            n = cnt[ty * d.gx + tx];
            if (SELF) {
                for (int w = 0; w < NT / 64; w++) f += rowsum[ly][w];
                for (int x = 0; x < lx; x++) f += cnt[ty * d.gx + tx0 + x];
            } else {
                f = s.tile_end[(size_t)v * d.T + ty * d.gx + tx] - n;
            }
        }
        // only lists that take the global-scratch sort need ids by slot: those beyond the long-list kernel's LDS, or — when the
        // trainer skips that launch — beyond the per-tile kernel's
        cur[threadIdx.x] = 0; first[threadIdx.x] = f; big[threadIdx.x] = n > (uint32_t)((d.long_sort && n >= (uint32_t)SORT_SMALL_CAP) ? SORT_LDS_CAP
                                                                                                                              : (d.mid_sort && n >= (uint32_t)SORT_TINY_CAP) ? SORT_SMALL_CAP : SORT_TINY_CAP);  // the list will not fit its sorter's LDS
    }
    __syncthreads();
    const size_t c0 = (size_t)v * d.NST + st;
    const uint32_t nc = s.coarse_count[c0];
    const uint32_t cstart = s.coarse_end[c0] - nc;
    const uint4* __restrict__ list = s.coarse_list + (size_t)v * d.Rcap + cstart;
    const uint32_t* __restrict__ dlist = s.coarse_depth + (size_t)v * d.Rcap + cstart;
    uint64_t* keys = reinterpret_cast<uint64_t*>(s.G + (size_t)v * d.Rcap * G_STRIDE);
    uint32_t* pl = s.point_list + (size_t)v * d.Rcap;
    uint32_t* ids = s.id_of_slot + (size_t)v * d.Rcap;
    // A wave takes 64 consecutive candidates and appends them TILE BY TILE: the candidates that cover the tile (a ballot) get
    // consecutive positions of a run reserved by the wave, so the key / id stores of a tile are contiguous runs instead of
    // 64 different cache lines per store instruction, and the 16 cursors are no longer hit by 64 lanes at once.
    // (One thread per candidate looping over its own tiles took 0.70 ms per launch at 1M splats @2048^2, almost all of it
    // waiting for scattered 8-byte stores and same-address LDS atomics.)
    const int lane = threadIdx.x & 63;
    const uint32_t seg_first = lane < STILE * STILE ? first[lane] : 0u;  // lane tl: start of tile tl's segment
    uint32_t zc = 0xFFFFFFFFu;   // lane tl: the depth bound of tile tl (Dims::cut; k_tile_count counted with the same test)
    if (CUT && lane < STILE * STILE) {
        const int tx = tx0 + (lane % STILE), ty = ty0 + (lane / STILE);
        if (tx < d.gx && ty < d.gy) zc = s.tile_zcut[(size_t)v * d.T + ty * d.gx + tx];
    }
    const unsigned long long big_mask = __ballot(lane < STILE * STILE && big[lane] != 0u);
    for (uint32_t c0 = (threadIdx.x >> 6) * 64; c0 < nc; c0 += NT) {
        const uint32_t c = c0 + lane;
        uint4 e = make_uint4(0, 0, 0, 0);
        uint64_t dz = 0;
        uint32_t dbits = 0;
        int rx0 = 0, ry0 = 0, rx1 = 0, x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        if (c < nc) {
            e = list[c];
            dbits = dlist[c];
            dz = (uint64_t)dbits << 32;
            rx0 = e.y & 0xffff; ry0 = e.y >> 16; rx1 = e.z & 0xffff;
            const int ry1 = e.z >> 16;
            x0 = max(rx0, tx0); x1 = min(rx1, tx0 + STILE);
            y0 = max(ry0, ty0); y1 = min(ry1, ty0 + STILE);
        }
        // pass 1: how many of the wave's candidates cover each of the 16 tiles (lane tl collects tile tl's count) ...
        uint32_t mine = 0;
#pragma unroll
        for (int tl = 0; tl < STILE * STILE; tl++) {
            const int x = tx0 + (tl % STILE), y = ty0 + (tl / STILE);
            const uint32_t k = (uint32_t)__popcll(__ballot(x >= x0 && x < x1 && y >= y0 && y < y1 && (!CUT || dbits <= (uint32_t)__builtin_amdgcn_readlane((int)zc, tl))));
            mine = lane == tl ? k : mine;
        }
        // ... ONE LDS atomic instruction reserves the runs of all 16 tiles (a returning atomic per tile round was 16 dependent
        // LDS round trips per 64 candidates: -6 % on the dense scene, nothing at cfg3) ...
        uint32_t runs = 0;
        if (lane < STILE * STILE && mine != 0u) runs = atomicAdd(&cur[lane], mine);
        runs += seg_first;  // lane tl: where this wave's entries of tile tl go (index into point_list)
        // ... pass 2: the entries, tile by tile; the candidates that cover a tile take consecutive positions of its run
#pragma unroll
        for (int tl = 0; tl < STILE * STILE; tl++) {
            const int x = tx0 + (tl % STILE), y = ty0 + (tl / STILE);
            const bool in = x >= x0 && x < x1 && y >= y0 && y < y1 && (!CUT || dbits <= (uint32_t)__builtin_amdgcn_readlane((int)zc, tl));
            const unsigned long long m = __ballot(in);
            if (m == 0ull) continue;
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)seg_first, tl);
            const uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)runs, tl);
            if (in) {
                const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t slot = e.w + (uint32_t)((y - ry0) * (rx1 - rx0) + (x - rx0));
                keys[(size_t)f + pos] = dz | slot;  // u64 index 2 * f + position in the segment; order inside the segment is irrelevant: the key is unique
                pl[pos] = e.x;
                if ((big_mask >> tl) & 1ull) ids[slot] = e.x;  // the long-list path looks the id up by slot after sorting keys only
            }
        }
    }
    // (Built and measured, not kept — rocprofv3 medians at cfg3, 55 us for this form: parking a wave's entries in LDS and
    // storing them as full 64-lane runs, 58 us; every lane walking its own tiles instead of 16 rounds, 70 us, of which 50 are
    // there without any global store.  k_tile_count, the same loads and ballots without pass 2, takes 15 us: the kernel pays
    // for pass 2's instructions, and the longest walk of 64 lanes is rarely shorter than 16 rounds.)
}

// Per-tile segments + tile_end (inclusive scan of tile_count) + tile_order (tiles by descending count) — one launch.
// Only when the tile count exceeds what one workgroup scans are the scan (three phases) and the order launches of their own.
int launch_tile_scatter(const Dims& d, const Scratch& s, uint32_t* partials, hipStream_t st) {
    if (d.NST == 0 || d.VG == 0 || d.T == 0) return GS_OK;
    static_assert(ORDER_BINS == WG, "one bin per thread");
    // every scatter workgroup of the one-launch form reads the tile counts in front of its rows: T / 2 words on average,
    // 64 MB per camera at 2048 x 2048 (T = 16384) — beyond that the separate scan is cheaper than the repeated reads
#ifdef GS_DIAG_SEPARATE_ORDER  // timing experiment: the scan + order workgroup as a launch of its own
    if (false) {
#else
    if (d.T <= g_scan_single_max && d.T <= 16384) {
#endif
        if (d.cut) {
            if (d.VG <= GS_WIDE_BIN_MAX_GROUPS) hipLaunchKernelGGL((k_tile_scatter<true, 1024, true>), dim3(d.NST + 1, d.VG), dim3(1024), 0, st, d, s);
            else hipLaunchKernelGGL((k_tile_scatter<true, WG, true>), dim3(d.NST + 1, d.VG), dim3(WG), 0, st, d, s);
        } else {
            if (d.VG <= GS_WIDE_BIN_MAX_GROUPS) hipLaunchKernelGGL((k_tile_scatter<true, 1024, false>), dim3(d.NST + 1, d.VG), dim3(1024), 0, st, d, s);
            else hipLaunchKernelGGL((k_tile_scatter<true, WG, false>), dim3(d.NST + 1, d.VG), dim3(WG), 0, st, d, s);
        }
    } else if (d.T <= g_scan_single_max) {
        hipLaunchKernelGGL(k_tile_scan_order_scan, dim3(d.VG), dim3(WG), 0, st, d, s);
        if (d.cut) hipLaunchKernelGGL((k_tile_scatter<false, WG, true>), dim3(d.NST, d.VG), dim3(WG), 0, st, d, s);
        else hipLaunchKernelGGL((k_tile_scatter<false, WG, false>), dim3(d.NST, d.VG), dim3(WG), 0, st, d, s);
    } else {
        GS_TRY(launch_scan_u32(s.tile_count, s.tile_end, d.T, d.T, d.VG, partials, st));
        hipLaunchKernelGGL(k_tile_scan_order_noscan, dim3(d.VG), dim3(WG), 0, st, d, s);
        if (d.cut) hipLaunchKernelGGL((k_tile_scatter<false, WG, true>), dim3(d.NST, d.VG), dim3(WG), 0, st, d, s);
        else hipLaunchKernelGGL((k_tile_scatter<false, WG, false>), dim3(d.NST, d.VG), dim3(WG), 0, st, d, s);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// per-tile sort on the unique 64-bit key: the tile's workgroup loads its entries (k_tile_scatter) into LDS and sorts:
// counting-rank sort for n <= 128, depth-bucketed rank sort up to the LDS capacity (2048 entries in the one-tile-per-
// workgroup kernel, 8192 in the long-list kernel), bitonic in global scratch (the not yet used gradient-row buffer G)
// beyond — the "tile-list spill path".
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ inline void bitonic_sort(uint64_t* a, uint32_t n2) {
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (n2 >> 1); t += NT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t l = i | j;
                const bool up = ((i & k) == 0);
                const uint64_t x = a[i], y = a[l];
                if ((x > y) == up) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
}

#ifndef GS_RANK_DIRECT_MAX
#define GS_RANK_DIRECT_MAX 128  // tuning hook (tools/build_variant.sh)
#endif

// Sort one tile's segment with NT threads and room for CAP entries in LDS (sk: CAP keys, sid: CAP ids).
// Every thread of the workgroup must call it (it synchronises); n > 0.
template <int NT, int CAP, int NBMAX>
__device__ inline void sort_tile(const Dims& d, const Scratch& s, int v, int tile, uint32_t n, uint64_t* sk, uint32_t* sid) {
    const uint32_t start = s.tile_end[(size_t)v * d.T + tile] - n;
    uint32_t* ids = s.id_of_slot + (size_t)v * d.Rcap;
    uint32_t* pl = s.point_list + (size_t)v * d.Rcap + start;
    uint32_t* sl = s.slot_list + (size_t)v * d.Rcap + start;
    // the tile's unsorted keys (k_tile_scatter); lists beyond CAP are sorted in place there (n2 < 2n slots reserved)
    uint64_t* a = reinterpret_cast<uint64_t*>(s.G + (size_t)v * d.Rcap * G_STRIDE) + 2 * (size_t)start;
    if (n <= (uint32_t)CAP) {
        for (uint32_t t = threadIdx.x; t < n; t += NT) { sk[t] = a[t]; sid[t] = pl[t]; }
        __syncthreads();
        constexpr uint32_t KPT = CAP / NT;  // keys per thread
        if (n <= GS_RANK_DIRECT_MAX) {
            // counting-rank sort: rank = #keys smaller is the final position (keys are unique); every thread ranks its
            // key against the whole list with broadcast 16-byte LDS reads
            const uint32_t ne = (n + 1) & ~1u;
            if (threadIdx.x == 0 && ne != n) sk[n] = ~0ull;
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < n; t += NT) {
                const uint64_t mine = sk[t];
                uint32_t rank = 0;
                const ulonglong2* pairs = reinterpret_cast<const ulonglong2*>(sk);
                for (uint32_t j = 0; j < ne / 2; j++) {
                    const ulonglong2 kk = pairs[j];
                    rank += (kk.x < mine) ? 1u : 0u;
                    rank += (kk.y < mine) ? 1u : 0u;
                }
                sl[rank] = (uint32_t)mine;
                pl[rank] = sid[t];
            }
            return;
        }
        // Longer lists: bucket by depth first (a monotone map of the depth bits onto NB buckets), then rank inside the
        // bucket — n*n/NB comparisons instead of n*n.  Skewed depth distributions only cost speed, never order.
        // NB grows with the list (about four keys per bucket, up to NBMAX): with a fixed 64 buckets the rank loops of a
        // dense scene (1100-2900 entries per tile at 1M splats @2048^2) ran 17-45 dependent LDS reads per key.
        uint32_t NB = 64;
        while (NB < (uint32_t)NBMAX && NB * 4 < n) NB <<= 1;
        __shared__ uint32_t dmin, dmax, bcount[NBMAX], bstart[NBMAX + 1];
        uint64_t key[KPT];
        uint32_t id[KPT], bk[KPT], bp[KPT];
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        if (threadIdx.x == 0) { dmin = 0xFFFFFFFFu; dmax = 0u; }
        for (uint32_t b = threadIdx.x; b < NB; b += NT) bcount[b] = 0;
#pragma unroll
        for (uint32_t k = 0; k < KPT; k++) {
            const uint32_t t = threadIdx.x + k * NT;
            key[k] = t < n ? sk[t] : 0ull;
            id[k] = t < n ? sid[t] : 0u;
            if (t < n) { const uint32_t dz = (uint32_t)(key[k] >> 32); lo = min(lo, dz); hi = max(hi, dz); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { atomicMin(&dmin, lo); atomicMax(&dmax, hi); }
        __syncthreads();
        const uint32_t base = dmin;
        const float scale = (float)NB / ((float)(dmax - base) + 1.0f);
#pragma unroll
        for (uint32_t k = 0; k < KPT; k++) {
            const uint32_t t = threadIdx.x + k * NT;
            if (t < n) {
                bk[k] = min(NB - 1, (uint32_t)((float)((uint32_t)(key[k] >> 32) - base) * scale));
                bp[k] = atomicAdd(&bcount[bk[k]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // exclusive scan of the NB bucket counts by one wave: NB / 64 consecutive buckets per lane
            const uint32_t per = NB >> 6, b0 = threadIdx.x * per;
            uint32_t sum = 0;
            for (uint32_t j = 0; j < per; j++) sum += bcount[b0 + j];
            uint32_t run = wave_incl_scan(sum) - sum;
            for (uint32_t j = 0; j < per; j++) { bstart[b0 + j] = run; run += bcount[b0 + j]; }
            if (threadIdx.x == 63) bstart[NB] = run;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < KPT; k++) {
            const uint32_t t = threadIdx.x + k * NT;
            if (t < n) sk[bstart[bk[k]] + bp[k]] = key[k];
        }
        __syncthreads();
        uint32_t rank[KPT];
#pragma unroll
        for (uint32_t k = 0; k < KPT; k++) {
            const uint32_t t = threadIdx.x + k * NT;
            rank[k] = 0;
            if (t < n) {
                const uint32_t b0 = bstart[bk[k]], b1 = bstart[bk[k] + 1];
                uint32_t r = b0;
                for (uint32_t j = b0; j < b1; j++) r += (sk[j] < key[k]) ? 1u : 0u;
                rank[k] = r;
            }
        }
        // the sorted (slot, id) pairs go out through LDS: the ranks are a permutation, so writing them straight to global
        // memory is n scattered 4-byte stores per array; parked at their rank in the key area they leave as coalesced runs
        __syncthreads();  // every rank loop has finished reading sk
#pragma unroll
        for (uint32_t k = 0; k < KPT; k++) {
            const uint32_t t = threadIdx.x + k * NT;
            if (t < n) sk[rank[k]] = ((uint64_t)id[k] << 32) | (uint32_t)key[k];
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < n; t += NT) {
            const uint64_t e = sk[t];
            sl[t] = (uint32_t)e;
            pl[t] = (uint32_t)(e >> 32);
        }
        return;
    }
    // the spill path: bitonic network in global scratch
    uint32_t n2 = 2;
    while (n2 < n) n2 <<= 1;
    for (uint32_t t = n + threadIdx.x; t < n2; t += NT) a[t] = ~0ull;
    __syncthreads();
    bitonic_sort<NT>(a, n2);
    for (uint32_t t = threadIdx.x; t < n; t += NT) {
        const uint32_t slot = (uint32_t)a[t];
        sl[t] = slot;
        pl[t] = ids[slot];
    }
}

// Three sorters by list length, each sized for its class (the host skips the launches of classes the scene does not have:
// Dims::mid_sort / long_sort; a list that outgrows its sorter's LDS anyway takes the global-scratch path there — slow, correct).
//
// Lists below SORT_TINY_CAP (512) entries: one 128-thread workgroup per tile, 7 KB of LDS.  A tile's sort is a chain of
// dependent latencies (order -> count / end -> keys, then a handful of LDS phases) with little work in between, so what
// counts is how many tiles a CU has in flight: 16-22 of these against the 6 of a 256-thread workgroup with room for 2048
// entries (cfg3, 100-500 entries per tile: 88 -> 50 us per 8 cameras).
__global__ __launch_bounds__(GS_SORT_TILE_NT) void k_tile_build_sort(Dims d, Scratch s) {
    __shared__ __align__(16) uint64_t sk[SORT_TINY_CAP];
    __shared__ uint32_t sid[SORT_TINY_CAP];
    const int v = blockIdx.y;
    if (s.flags[v * 4 + 0] & 1u) return;  // overflowed view: nothing was scattered
    const int tile = (int)s.tile_order[(size_t)v * d.T + d.small_first + blockIdx.x];  // (positions in front of small_first: k_tile_sort_mid)
    if ((unsigned)tile >= (unsigned)d.T) return;  // never an address unless it is a tile (k_render_fwd)
    const uint32_t n = s.tile_count[(size_t)v * d.T + tile];
    if (n == 0) return;
    if (n >= (uint32_t)SORT_TINY_CAP && (d.mid_sort || (d.long_sort && n >= (uint32_t)SORT_SMALL_CAP))) return;  // the next sorters' lists
    sort_tile<GS_SORT_TILE_NT, SORT_TINY_CAP, 128>(d, s, v, tile, n, sk, sid);
}

// Lists of SORT_TINY_CAP entries and more sit in the first sort_marks[1] positions of the tile order, the lists of SORT_SMALL_CAP
// entries and more in its first sort_marks[0] positions (the prefixes interleave the XCD classes' lists: a walker skips what is not its
// class).  Persistent workgroups walk the prefixes: 256 threads with 2048 entries of LDS (24 KB, 6 per CU) ...
// k_tile_sort_mid takes ONE tile per workgroup (grid: the head's length as the host knows it from two steps ago, or the whole
// order); k_tile_sort_mid_walk, a few workgroups per camera, walks what is left of the head behind that grid — normally
// nothing.  (One kernel that walks in strides is the obvious form; the sort inlined into a loop takes 120 registers instead of
// 80 — four workgroups per CU where the LDS allows six — and ran 15-25 % slower on a dense scene.)
__device__ inline bool mid_sorter_owns(const Dims& d, uint32_t idx, uint32_t n) {
    // short lists only where the short-list sorter's grid does not reach; long ones only without their own sorter
    return n != 0 && (n < (uint32_t)SORT_TINY_CAP ? idx < (uint32_t)d.small_first : !(d.long_sort && n >= (uint32_t)SORT_SMALL_CAP));
}
__global__ __launch_bounds__(WG) void k_tile_sort_mid(Dims d, Scratch s) {
    __shared__ __align__(16) uint64_t sk[SORT_SMALL_CAP];
    __shared__ uint32_t sid[SORT_SMALL_CAP];
    const int v = blockIdx.y;
    if (s.flags[v * 4 + 0] & 1u) return;
    const uint32_t idx = blockIdx.x;
    if (idx >= max(s.sort_marks[v * 2 + 1], (uint32_t)d.small_first)) return;
    const int tile = (int)s.tile_order[(size_t)v * d.T + idx];
    if ((unsigned)tile >= (unsigned)d.T) return;
    const uint32_t n = s.tile_count[(size_t)v * d.T + tile];
    if (!mid_sorter_owns(d, idx, n)) return;
    sort_tile<WG, SORT_SMALL_CAP, 256>(d, s, v, tile, n, sk, sid);
}
__global__ __launch_bounds__(WG) void k_tile_sort_mid_walk(Dims d, Scratch s, uint32_t first) {
    __shared__ __align__(16) uint64_t sk[SORT_SMALL_CAP];
    __shared__ uint32_t sid[SORT_SMALL_CAP];
    const int v = blockIdx.y;
    if (s.flags[v * 4 + 0] & 1u) return;
    const uint32_t n_walk = max(s.sort_marks[v * 2 + 1], (uint32_t)d.small_first);
    for (uint32_t idx = first + blockIdx.x; idx < n_walk; idx += gridDim.x) {
        const int tile = (int)s.tile_order[(size_t)v * d.T + idx];
        if ((unsigned)tile >= (unsigned)d.T) continue;
        const uint32_t n = s.tile_count[(size_t)v * d.T + tile];
        if (!mid_sorter_owns(d, idx, n)) continue;
        sort_tile<WG, SORT_SMALL_CAP, 256>(d, s, v, tile, n, sk, sid);
        __syncthreads();  // LDS is reused by the next tile
    }
}

// ... and 1024 threads with SORT_LDS_CAP (8192) entries — 96 KB, one workgroup per CU, which is why shorter lists do not go
// through here; only lists beyond that spill to global scratch.
constexpr int LONG_NT = 1024;
__global__ __launch_bounds__(LONG_NT) void k_tile_sort_long(Dims d, Scratch s) {
    __shared__ __align__(16) uint64_t sk[SORT_LDS_CAP];
    __shared__ uint32_t sid[SORT_LDS_CAP];
    const int v = blockIdx.y;
    if (s.flags[v * 4 + 0] & 1u) return;
    const uint32_t n_long = s.sort_marks[v * 2];
    for (uint32_t idx = blockIdx.x; idx < n_long; idx += gridDim.x) {
        const int tile = (int)s.tile_order[(size_t)v * d.T + idx];
        if ((unsigned)tile >= (unsigned)d.T) continue;
        const uint32_t n = s.tile_count[(size_t)v * d.T + tile];
        if (n < (uint32_t)SORT_SMALL_CAP) continue;
        sort_tile<LONG_NT, SORT_LDS_CAP, 1024>(d, s, v, tile, n, sk, sid);
        __syncthreads();  // LDS is reused by the next tile
    }
}

constexpr int MID_WALK_ONLY = 512;  // heads up to this many tiles per camera (by the host's hint) are left to the walker kernel alone
int launch_tile_build_sort(const Dims& d, const Scratch& s, hipStream_t st) {
    if (d.T == 0 || d.VG == 0) return GS_OK;
    const int small_first = d.mid_sort ? std::min(std::max(d.small_first, 0), d.T) : 0;  // (a head without its walker would stay unsorted)
    Dims dd = d;
    dd.small_first = small_first;
    if (small_first < d.T) hipLaunchKernelGGL(k_tile_build_sort, dim3(d.T - small_first, d.VG), dim3(GS_SORT_TILE_NT), 0, st, dd, s);
    if (d.mid_sort) {
        // a workgroup per tile of the head when the host knows its length (and a few that walk on behind, should the head have
        // outgrown the hint); else one per tile of the order
#ifdef GS_DIAG_SORT_MID_GRID_T
        const int per_cam = d.T;
#else
        const int per_cam = d.mid_grid > 0 ? std::min(d.T, std::max(d.mid_grid, small_first)) : d.T;
#endif
        if (per_cam <= MID_WALK_ONLY && per_cam < d.T) {
            // a short head (a scene at the edge of having such lists: cfg3's longest is 509): the walker alone, one workgroup
            // per tile of the hint — its four workgroups per CU do not matter for a few tiles, a second launch does
            hipLaunchKernelGGL(k_tile_sort_mid_walk, dim3(per_cam, d.VG), dim3(WG), 0, st, dd, s, 0u);
        } else {
            hipLaunchKernelGGL(k_tile_sort_mid, dim3(per_cam, d.VG), dim3(WG), 0, st, dd, s);
            if (per_cam < d.T) hipLaunchKernelGGL(k_tile_sort_mid_walk, dim3(std::min(d.T - per_cam, 32), d.VG), dim3(WG), 0, st, dd, s, (uint32_t)per_cam);
        }
    }
    if (d.long_sort) {
        // one long-list workgroup fills a CU (96 KB LDS): about one per CU over all cameras
        const int per_cam = std::min(d.T, std::max(16, std::min(256, 256 / std::max(d.VG, 1))));
        hipLaunchKernelGGL(k_tile_sort_long, dim3(per_cam, d.VG), dim3(LONG_NT), 0, st, dd, s);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ranges[tile] = (first, one-past-last), (0,0) for empty tiles — upstream's identifyTileRanges output
__global__ void k_ranges(Dims d, Scratch s, uint32_t* ranges) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= d.T * d.VG) return;
    const uint32_t n = s.tile_count[t], e = s.tile_end[t];
    ranges[2 * (size_t)t] = n ? e - n : 0;
    ranges[2 * (size_t)t + 1] = n ? e : 0;
}

int launch_ranges(const Dims& d, const Scratch& s, uint32_t* ranges, hipStream_t st) {
    const int n = d.T * d.VG;
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(k_ranges, dim3((n + 255) / 256), dim3(256), 0, st, d, s, ranges);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
