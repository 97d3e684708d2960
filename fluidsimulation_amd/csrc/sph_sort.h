// sph_sort.h -- wavefront-ballot / LDS radix sort (stable, LSD, 8-bit digits) and the cell
// (start,end) table.  Replaces the reference's bitonic network (FluidSolver.compute:47-83 driven
// by FluidSPHByGPUSystem.cs:279-299: log2P(log2P+1)/2 dispatches of compare-and-swap over global
// memory) and its ClearTable / MapTable kernels (FluidSolver.compute:85-112).
//
// Per 8-bit pass, three launches over tiles of 4096 keys:
//   k_radix_hist     per-tile digit counts (LDS sub-histograms)            reads 4 B/key
//   k_radix_scan     per-digit exclusive scan over tiles + digit totals    (small)
//   k_radix_scatter  stable rank by wave ballots (64-wide match-any), per-wave digit counters in
//                    LDS, tile sorted by digit in LDS, coalesced write-out  reads 8, writes 8 B/key
// Algorithmic bytes: 20 B per key per pass.  HBM-bound; no MFMA.
#pragma once
#include "sph_device.h"

#define RS_THREADS 256
#define RS_WAVES 4
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over a 256-thread block; tmp: 4 words of LDS
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *tmp,
                                                        uint32_t *total)
{
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) tmp[w] = incl;
    __syncthreads();
    uint32_t basew = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; i++) {
        uint32_t t = tmp[i];
        if (i < w) basew += t;
        tot += t;
    }
    __syncthreads();
    if (total) *total = tot;
    return basew + incl - v;
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint32_t *__restrict__ keys,
                                                           int n, int shift,
                                                           uint32_t *__restrict__ hist,
                                                           int nblocks)
{
    __shared__ uint32_t h[256 * 8]; // [digit][sub]: 8 sub-histograms spread over 8 banks
    const int tid = threadIdx.x, sub = tid & 7;
    for (int i = tid; i < 256 * 8; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        long i = base + k * RS_THREADS + tid;
        if (i < n) atomicAdd(&h[((keys[i] >> shift) & 255u) * 8 + sub], 1u);
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int s = 0; s < 8; s++) c += h[tid * 8 + s];
    hist[(size_t)tid * nblocks + blockIdx.x] = c;
}

// one block per digit: in-place exclusive scan of that digit's row of tile counts
__global__ __launch_bounds__(RS_THREADS) void k_radix_scan(uint32_t *__restrict__ hist,
                                                           uint32_t *__restrict__ rowTotal,
                                                           int nblocks)
{
    __shared__ uint32_t tmp[RS_WAVES];
    uint32_t *row = hist + (size_t)blockIdx.x * nblocks;
    const int tid = threadIdx.x;
    const int per = (nblocks + RS_THREADS - 1) / RS_THREADS;
    const int lo = min(tid * per, nblocks), hi = min(lo + per, nblocks);
    uint32_t sum = 0;
    for (int i = lo; i < hi; i++) sum += row[i];
    uint32_t total;
    uint32_t run = block_excl_scan_256(sum, tmp, &total);
    for (int i = lo; i < hi; i++) {
        uint32_t c = row[i];
        row[i] = run;
        run += c;
    }
    if (tid == 0) rowTotal[blockIdx.x] = total;
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(
    const uint32_t *__restrict__ keysIn, const uint32_t *__restrict__ valsIn,
    uint32_t *__restrict__ keysOut, uint32_t *__restrict__ valsOut, int n, int shift,
    const uint32_t *__restrict__ histScanned, const uint32_t *__restrict__ rowTotal, int nblocks)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint32_t digitStart[256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t tmp[RS_WAVES];
    __shared__ uint2 stage[RS_TILE];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    // global base of every digit: exclusive scan of the digit totals
    const uint32_t digitBase = block_excl_scan_256(rowTotal[tid], tmp, nullptr);
    const uint32_t myTileBase = histScanned[(size_t)tid * nblocks + blockIdx.x];

    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
    const long tile0 = (long)blockIdx.x * RS_TILE;
    const long wbase = tile0 + (long)w * (RS_ITEMS * 64);
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        long i = wbase + k * 64 + lane;
        bool valid = i < n;
        key[k] = valid ? keysIn[i] : 0xffffffffu;
        val[k] = valid ? (valsIn ? valsIn[i] : (uint32_t)i) : 0u;
    }
    __syncthreads(); // counters zeroed

    // stable rank inside the wave: lanes holding the same digit find each other with 8 ballots
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const bool valid = (wbase + k * 64 + lane) < n;
        const uint32_t d = valid ? ((key[k] >> shift) & 255u) : 255u;
        unsigned long long peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = cnt[w][d];
        const int r = mbcnt64(peers);
        __builtin_amdgcn_wave_barrier();
        if (r == 0) cnt[w][d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[k] = before + (uint32_t)r;
    }
    __syncthreads();

    // thread = digit: prefix over the 4 waves, tile total, tile-local start, global base
    uint32_t tileCount = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; i++) {
        uint32_t c = cnt[i][tid];
        cnt[i][tid] = tileCount;
        tileCount += c;
    }
    const uint32_t dstart = block_excl_scan_256(tileCount, tmp, nullptr);
    digitStart[tid] = dstart;
    gbase[tid] = digitBase + myTileBase;
    __syncthreads();

#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const bool valid = (wbase + k * 64 + lane) < n;
        const uint32_t d = valid ? ((key[k] >> shift) & 255u) : 255u;
        const uint32_t lp = digitStart[d] + cnt[w][d] + rank[k];
        stage[lp] = make_uint2(key[k], val[k]);
    }
    __syncthreads();

    const long remain = (long)n - tile0;
    const int nvalid = remain < RS_TILE ? (int)remain : RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const int s = k * RS_THREADS + tid;
        if (s < nvalid) {
            const uint2 kv = stage[s];
            const uint32_t d = (kv.x >> shift) & 255u;
            const uint32_t gi = gbase[d] + ((uint32_t)s - digitStart[d]);
            keysOut[gi] = kv.x;
            valsOut[gi] = kv.y;
        }
    }
}

// ClearTable + MapTable (FluidSolver.compute:85-112): the tables are zeroed with memset nodes,
// then every run of equal keys writes its (start, end); ttable does the same per tile
// (tile = key / SPH_TCELLS) for the neighbour kernels' one-workgroup-per-tile launch.
__global__ void k_map_table(const uint32_t *__restrict__ skeys, int n, int2 *__restrict__ table,
                            int2 *__restrict__ ttable)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t k = skeys[s];
    const uint32_t kp = s > 0 ? skeys[s - 1] : 0u, kn = s < n - 1 ? skeys[s + 1] : 0u;
    if (s == 0 || k != kp) table[k].x = s;
    if (s == n - 1 || k != kn) table[k].y = s + 1;
    if (ttable) {
        const uint32_t t = k / SPH_TCELLS;
        if (s == 0 || t != kp / SPH_TCELLS) ttable[t].x = s;
        if (s == n - 1 || t != kn / SPH_TCELLS) ttable[t].y = s + 1;
    }
}

// Compacts the non-empty tiles into tileList, tileCtl[0] = their number.  One atomic per block of
// 1024 tiles (a single word serves 88 returning atomics per microsecond: with one per 256 tiles the
// atomics were the whole kernel on C3); the order of the list is arbitrary, it only decides which
// workgroup takes which tile.
#define TL_ITEMS 4
__global__ __launch_bounds__(256) void k_tile_list(const int2 *__restrict__ ttable, int tiles,
                                                   uint32_t *__restrict__ tileList,
                                                   uint32_t *__restrict__ tileCtl)
{
    __shared__ uint32_t waveCount[TL_ITEMS][4], base;
    const int w = threadIdx.x >> 6;
    bool full[TL_ITEMS];
    unsigned long long m[TL_ITEMS];
#pragma unroll
    for (int j = 0; j < TL_ITEMS; j++) {
        const int t = (blockIdx.x * TL_ITEMS + j) * 256 + (int)threadIdx.x;
        full[j] = false;
        if (t < tiles) {
            const int2 r = ttable[t];
            full[j] = r.y > r.x;
        }
    }
#pragma unroll
    for (int j = 0; j < TL_ITEMS; j++) {
        m[j] = __ballot(full[j]);
        if (lane_id() == 0) waveCount[j][w] = (uint32_t)__popcll(m[j]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int j = 0; j < TL_ITEMS; j++)
            for (int i = 0; i < 4; i++) total += waveCount[j][i];
        base = total ? atomicAdd(&tileCtl[0], total) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TL_ITEMS; j++) {
        if (full[j]) {
            uint32_t off = base + (uint32_t)mbcnt64(m[j]);
            for (int jj = 0; jj < j; jj++)
                for (int i = 0; i < 4; i++) off += waveCount[jj][i];
            for (int i = 0; i < w; i++) off += waveCount[j][i];
            tileList[off] = (uint32_t)((blockIdx.x * TL_ITEMS + j) * 256 + (int)threadIdx.x);
        }
    }
}

// ------------------------------------------------------------------------------------------
// CELL COUNTING SORT -- what the step entry points use instead of the radix passes.
// The keys are cell numbers and the arrays are in last step's cell order, so a step moves few
// particles to another cell.  Three passes over the particles' 4-byte words do what the radix sort
// needs 3 x 3 launches and 20 B per key and pass for (FluidSolver.compute:47-112 in one go):
//   k_cell_count   every run of equal keys inside a wave takes its places in the cell with ONE
//                  atomic (about ten per wave), rank[i] = place inside the cell     reads 4, writes 4 B
//   k_cell_sums / k_cell_scan_sums / k_cell_table   exclusive scan of the cell counts -> the
//                  (start, end) tables of MapTable, per cell and per tile           12 B per CELL
//   k_cell_place   perm[start(key) + rank] = i, sorted keys                         reads 8, writes 8 B
// The order INSIDE a cell is the arrival order of the atomics, i.e. arbitrary: k_reorder re-ranks
// every cell by particle id (it always has), so the resident order stays "stable by (cell, id)"
// and nothing downstream sees the difference.  sph_radix_sort / sph_sort_pairs keep the radix sort
// (their permutation is the stable one the oracle gives).
#define CS_CELLS_PER_BLOCK 1024 // 256 threads x 4 cells

#define CS_COUNT_ITEMS 4 // keys per thread: the returning atomics are latency, not throughput --
                         // four independent ones in flight per wave instead of one
__global__ __launch_bounds__(256) void k_cell_count(const uint32_t *__restrict__ keys, int n,
                                                    uint32_t *__restrict__ counts,
                                                    uint32_t *__restrict__ rank)
{
    const int lane = lane_id();
    const long first = (long)blockIdx.x * (256 * CS_COUNT_ITEMS) + threadIdx.x;
    uint32_t key[CS_COUNT_ITEMS], base[CS_COUNT_ITEMS];
    int start[CS_COUNT_ITEMS];
    bool valid[CS_COUNT_ITEMS];
#pragma unroll
    for (int j = 0; j < CS_COUNT_ITEMS; j++) {
        const long i = first + j * 256;
        valid[j] = i < n;
        key[j] = valid[j] ? keys[i] : 0xffffffffu; // (no cell has this number)
    }
    // every run of equal keys inside a wave takes its places in the cell with one atomic, issued
    // by its first lane (a run: from a head to the next head)
#pragma unroll
    for (int j = 0; j < CS_COUNT_ITEMS; j++) {
        const uint32_t prev = __shfl_up(key[j], 1, 64);
        const bool head = lane == 0 || key[j] != prev;
        const unsigned long long heads = __ballot(head);
        const unsigned long long upto = heads & (~0ull >> (63 - lane));
        start[j] = 63 - __clzll(upto);
        const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
        const int next = above ? lane + 1 + __builtin_ctzll(above) : 64;
        base[j] = 0;
        if (head && valid[j]) base[j] = atomicAdd(&counts[key[j]], (uint32_t)(next - lane));
    }
#pragma unroll
    for (int j = 0; j < CS_COUNT_ITEMS; j++) {
        const uint32_t b = __shfl(base[j], start[j], 64);
        if (valid[j]) rank[first + j * 256] = b + (uint32_t)(lane - start[j]);
    }
}

// sums of CS_CELLS_PER_BLOCK counts (ncells4 = number of uint4 quads of counts, zero-padded)
__global__ __launch_bounds__(256) void k_cell_sums(const uint4 *__restrict__ counts4, int nquads,
                                                   uint32_t *__restrict__ blockSums)
{
    __shared__ uint32_t tmp[RS_WAVES];
    const int q = blockIdx.x * 256 + threadIdx.x;
    uint32_t s = 0;
    if (q < nquads) {
        const uint4 c = counts4[q];
        s = (c.x + c.y) + (c.z + c.w);
    }
    uint32_t total;
    block_excl_scan_256(s, tmp, &total);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}

// one block: in-place exclusive scan of the block sums
__global__ __launch_bounds__(RS_THREADS) void k_cell_scan_sums(uint32_t *__restrict__ blockSums, int nblocks)
{
    __shared__ uint32_t tmp[RS_WAVES];
    const int tid = threadIdx.x;
    const int per = (nblocks + RS_THREADS - 1) / RS_THREADS;
    const int lo = min(tid * per, nblocks), hi = min(lo + per, nblocks);
    uint32_t sum = 0;
    for (int i = lo; i < hi; i++) sum += blockSums[i];
    uint32_t run = block_excl_scan_256(sum, tmp, nullptr);
    for (int i = lo; i < hi; i++) {
        const uint32_t c = blockSums[i];
        blockSums[i] = run;
        run += c;
    }
}

// table[c] = (start, end) of cell c, (0, 0) for an empty one, as k_map_table leaves it; the same
// per tile of SPH_TCELLS = 32 consecutive cells (eight threads).  cells is a multiple of 32.
__global__ __launch_bounds__(256) void k_cell_table(const uint4 *__restrict__ counts4, int nquads,
                                                    const uint32_t *__restrict__ blockSums, int cells,
                                                    int2 *__restrict__ table, int2 *__restrict__ ttable)
{
    static_assert(SPH_TCELLS == 32, "a tile is the cells of eight threads");
    __shared__ uint32_t tmp[RS_WAVES];
    const int q = blockIdx.x * 256 + threadIdx.x;
    uint4 c = make_uint4(0, 0, 0, 0);
    if (q < nquads) c = counts4[q];
    const uint32_t s = (c.x + c.y) + (c.z + c.w);
    const uint32_t at = blockSums[blockIdx.x] + block_excl_scan_256(s, tmp, nullptr);
    // the tile's particles: the sum over its eight threads (aligned groups of eight lanes)
    uint32_t ts = s;
    ts += __shfl_xor(ts, 1, 64);
    ts += __shfl_xor(ts, 2, 64);
    ts += __shfl_xor(ts, 4, 64);
    const int cell0 = q * 4;
    if (cell0 >= cells) return; // (the padding, and the bucket of the dead records behind the cells)
    const int a0 = (int)at, a1 = a0 + (int)c.x, a2 = a1 + (int)c.y, a3 = a2 + (int)c.z, a4 = a3 + (int)c.w;
    int4 lo = make_int4(c.x ? a0 : 0, c.x ? a1 : 0, c.y ? a1 : 0, c.y ? a2 : 0);
    int4 hi = make_int4(c.z ? a2 : 0, c.z ? a3 : 0, c.w ? a3 : 0, c.w ? a4 : 0);
    int4 *out = (int4 *)(table + cell0);
    out[0] = lo;
    out[1] = hi;
    if ((threadIdx.x & 7) == 0) ttable[cell0 / SPH_TCELLS] = ts ? make_int2(a0, a0 + (int)ts) : make_int2(0, 0);
}

// places every particle: deadStart = first slot of the records whose key is `cells` (slab mode:
// dead ghosts, cut off by the caller)
#define CS_PLACE_ITEMS 4 // particles per thread (their look-ups in flight together)
__global__ __launch_bounds__(256) void k_cell_place(const uint32_t *__restrict__ keys,
                                                    const uint32_t *__restrict__ rank, int n,
                                                    const int2 *__restrict__ table, uint32_t cells,
                                                    uint32_t deadStart, uint32_t *__restrict__ perm,
                                                    uint32_t *__restrict__ skeys)
{
    const long first = (long)blockIdx.x * (256 * CS_PLACE_ITEMS) + threadIdx.x;
    uint32_t key[CS_PLACE_ITEMS], r[CS_PLACE_ITEMS], st[CS_PLACE_ITEMS];
#pragma unroll
    for (int j = 0; j < CS_PLACE_ITEMS; j++) {
        const long i = first + j * 256;
        key[j] = r[j] = 0;
        if (i < n) {
            key[j] = keys[i];
            r[j] = rank[i];
        }
    }
#pragma unroll
    for (int j = 0; j < CS_PLACE_ITEMS; j++)
        st[j] = (first + j * 256 < n && key[j] < cells) ? (uint32_t)table[key[j]].x : deadStart;
#pragma unroll
    for (int j = 0; j < CS_PLACE_ITEMS; j++) {
        const long i = first + j * 256;
        if (i < n) {
            const uint32_t dst = st[j] + r[j];
            perm[dst] = (uint32_t)i;
            skeys[dst] = key[j];
        }
    }
}
