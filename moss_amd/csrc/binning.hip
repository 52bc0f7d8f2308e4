// binning.hip -- (Gaussian, tile) instance binning and the per-tile depth sort.
//
// What the reference does (DGR/cuda_rasterizer/rasterizer_impl.cu:279-320): inclusive scan of tiles_touched,
// duplicateWithKeys writes one (tile<<32|depth_bits, gaussian_id) pair per instance grouped by Gaussian, a
// device-wide 64-bit CUB radix sort (6 passes over 12-byte pairs) orders them by (tile, depth) -- stable, so ties
// keep ascending Gaussian id -- and identifyTileRanges finds the tile boundaries.
//
// What this file does instead (same result, bit for bit, see moss_raster_export_binning):
//   1. the per-tile instance COUNTS are already known (histogram filled by the preprocess kernel), so an exclusive
//      scan over the tiles gives every tile's [start,end) range directly -- no boundary search, and the tile id
//      never has to be part of a sort key;
//   2. scatter: each instance is dropped into its tile's bucket (slot order inside a bucket is arbitrary) as a
//      64-bit key (depth_bits << 32 | gaussian_id) -- by scatter_kernel into the tile's final range on the synchronous path; on the
//      asynchronous path by the PREPROCESS kernel itself into fixed-stride buckets (preprocess.hip, scatter mode: no scan in front
//      of the keys, one kernel and one pass over the Gaussians less), the sort workgroups then derive their chunk tables from the
//      tile counts themselves and the scan rides along with the sort kernel as one extra block;
//   3. each bucket is sorted on those 64-bit keys in two instance-parallel steps: 1024-key chunks are sorted in LDS
//      (bitonic network, one workgroup per chunk), then every instance finds its final rank by binary search in its
//      tile's other chunks.  (depth_bits, id) is a total order, so the result is unique and equals the reference's
//      stable sort regardless of scatter order -- and no workgroup ever owns more than 1024 keys (no long-tile tail).
//   Sorting 8-byte keys once in LDS replaces 6 global passes over 12-byte pairs: algorithmic HBM traffic of the
//   sort falls from >= 24 B/instance/pass to 8 B write + 8 B read + 4 B write per instance in total.
#include "common.h"

namespace moss {

namespace {

// Keys sorted per workgroup by chunk_sort_kernel: SORT_THREADS threads with KPT keys each (element i of a chunk in thread i / KPT,
// register i % KPT).  KPT = 1: 1024-key chunks, as in rounds 1-4.  Round 5 measured larger chunks (round 4's review, item 4: "4 096-key
// chunks with 4 keys per thread"): scripts/micro/chunk_sort_kpt.hip (profiles/r05_chunk_sort_kpt_microbench.log) sorts the bench frame's
// 239k keys in 13.4 / 14.0 / 21.9 us as chunks of 1024 / 2048 / 4096 (a 4096-key chunk is 36k cycles on its own workgroup and the frame
// then has 186 workgroups for 256 CUs); and with KPT = 2 in THIS file -- the network below is written for any KPT -- the frame's sort
// stayed at 12.8 us while merge_gather, which the larger chunks were meant to relieve of sibling loads and searches, went 17.8 -> 19.0 us
// (configs[1]: 6.9 -> 10.2 us, configs[4]: sort 27.6 -> 33.8 us): four 512-thread parts per chunk, most of them idle for tiles below
// 2048 entries, and each part of a two-chunk tile still loads the whole sibling (profiles/r05_notes.md).  Not adopted.
constexpr int SORT_THREADS = 1024, KPT = 1;
constexpr int CHUNK = SORT_THREADS * KPT;

// The scan's outputs for one block of `NT` threads (NT a multiple of 64, <= 1024): see scan_kernel.  Shared by scan_kernel (its own
// launch: synchronous mode, where the host sizes the binning buffer from R before anything else can run) and by the LAST block of
// scatter_kernel (asynchronous mode: the scan rides along with the scatter instead of costing a 5 us launch of its own).
template <int NT>
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t* s_wave /* >= NT/64 */, uint32_t& total)
{
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();                 // s_wave may still be read from a previous call
    if (lane == 63) s_wave[w] = x;
    __syncthreads();
    uint32_t base = 0u, tot = 0u;
#pragma unroll
    for (int i = 0; i < NW; i++) { const uint32_t t = s_wave[i]; if (i < w) base += t; tot += t; }
    total = tot;
    return base + x - v;
}

template <int NT>
__device__ __forceinline__ void scan_outputs(int T, const uint32_t* __restrict__ tile_count, uint2* __restrict__ ranges,
                                             uint32_t* __restrict__ chunk_base, uint32_t* __restrict__ tile_order,
                                             uint32_t* __restrict__ header, uint32_t capacity, int light_log2,
                                             uint32_t* s_wave, uint32_t* s_max, uint32_t* s_bucket /* 34 */,
                                             int n_groups,
                                             uint32_t* __restrict__ flags_acc, uint32_t* __restrict__ queues, uint4* __restrict__ work_table,
                                             const uint32_t* __restrict__ group_rtot, uint32_t* __restrict__ group_rbase,
                                             uint32_t pool_cap /* cells the record pool of this binning buffer holds (0xffffffff: sized afterwards) */,
                                             uint32_t key_stride = 0u /* != 0: the keys sit in per-tile buckets of this many slots (preprocess.hip, scatter mode) */,
                                             bool forward_only = false /* MOSS_FORWARD_ONLY: there is no record pool -- nothing to overflow */)
{
    const int tid = threadIdx.x;
    // every header word is WRITTEN here and the queue words are zeroed (the blend kernels pop from them): nothing in the image
    // buffer needs a clear in front of the forward when the counters kernels add to live in the caller's frame state
    if (tid < Q_LINES) queues[(size_t)tid * QLINE_WORDS] = 0u;
    if (tid >= 8 && tid < HEADER_WORDS && tid != 9) header[tid] = 0u;     // ([9]: the pool's cells in use, written below)
    if (tid == 0) *s_max = 0;
    if (tid < 34) s_bucket[tid] = 0;
    // FIRST (while nothing else is live: this block shares the sort kernel's register budget -- 64 VGPRs for two workgroups per CU):
    // where the cell runs of every group of 256 Gaussians start in the record pool (the preprocess kernel left group-relative run starts
    // and the groups' totals), and how many cells the frame needs: header[9].
    // (forward only: the count is still taken -- header[3] stays the capacity a TRAINING forward of this frame would need, so a
    // forward-only probe can size a training step's capacity -- but there is no pool to overflow)
    uint32_t pool_total;
    {
        const int gchunk = (n_groups + NT - 1) / NT;
        const int gb = tid * gchunk, ge = min(n_groups, gb + gchunk);
        uint32_t gs = 0;
        for (int i = gb; i < ge; i++) gs += group_rtot[i];
        uint32_t goff = block_scan<NT>(gs, s_wave, pool_total);
        for (int i = gb; i < ge; i++) { group_rbase[i] = goff; goff += group_rtot[i]; }
    }
    const int chunk = (T + NT - 1) / NT;
    const int b = tid * chunk, e = min(T, b + chunk);
    uint32_t sum = 0, mx = 0, nch = 0;
    for (int i = b; i < e; i++) { const uint32_t v = tile_count[i]; sum += v; mx = max(mx, v); nch += (v + CHUNK - 1) / CHUNK; }
    uint32_t total, total_chunks;
    uint32_t off = block_scan<NT>(sum, s_wave, total);
    uint32_t coff = block_scan<NT>(nch, s_wave, total_chunks);
    // Asynchronous mode (no host read-back): the binning buffer was sized for `capacity` instances before R was known.
    // If this frame needs more, render NOTHING (all ranges empty, no queued work) and raise the overflow flag: the
    // following kernels stay inside the buffer and the host reports the error at its next check.
    // (bucketed keys: a tile that outgrew its bucket lost keys -- the same answer, and `needed` is scaled so that the capacity the
    // caller derives from it holds the longest list: the bucket size is proportional to the capacity)
    bool overflow = total > capacity;
    uint32_t needed = total;                             // the CAPACITY that holds this frame: header[3] (its instances: header[6])
    // A frame that needs more cells than the record pool holds is dropped like one that needs more instances than the capacity (`needed`
    // so that the pool of the caller's next capacity holds it: POOL_CELLS_PER_INSTANCE R + 4096 cells, BinView::default_pool_cells).
    needed = max(needed, pool_total / (uint32_t)POOL_CELLS_PER_INSTANCE + 1u);
    if (!forward_only && (pool_total > pool_cap || pool_total > POOL_MAX_CELLS)) overflow = true;
    if (key_stride != 0u) {
        if (mx) atomicMax(s_max, mx);
        __syncthreads();
        const uint32_t longest = *s_max;
        if (longest > key_stride) {
            overflow = true;
            // (in float, rounded up by a margin: a hint for the caller's capacity policy; a 64-bit division here cost this kernel --
            // the scan block shares the sort kernel's register budget -- its second workgroup per CU: 70 VGPRs, 12.9 -> 16.7 us)
            const float want = (float)longest * (float)capacity / (float)key_stride * 1.001f + 2.0f;
            needed = max(needed, want >= 4.29e9f ? 0xfffffff0u : (uint32_t)want);
        }
        // (s_max keeps the longest list: the atomicMax below repeats it for the lists that are kept, header[1] reports it either way)
    }
    const uint32_t off_first = off;                      // where this thread's first tile starts (the loop below walks on from here)
    for (int i = b; i < e; i++) {
        const uint32_t v = overflow ? 0u : tile_count[i];
        ranges[i] = make_uint2(off, off + v); off += v;
        chunk_base[i] = coff; coff += (v + CHUNK - 1) / CHUNK;
    }
    if (overflow) { sum = 0; mx = 0; off = 0; coff = 0; }
    if (mx) atomicMax(s_max, mx);
    // tile_order: tiles grouped by floor(log2(list length)), longest class first, empty tiles last.  The blend kernels
    // pull (tile, quadrant) work items in this order from an atomic queue (longest-processing-time-first balancing).
    for (int i = b; i < e; i++) { const uint32_t v = overflow ? 0u : tile_count[i]; atomicAdd(&s_bucket[v ? (uint32_t)__clz((int)v) : 32u], 1u); }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int k = 0; k < 33; k++) { const uint32_t c = s_bucket[k]; s_bucket[k] = acc; acc += c; }
        header[0] = overflow ? 0u : total; header[1] = *s_max; header[4] = overflow ? 0u : total_chunks;
        header[5] = s_bucket[32];                      // number of tiles that own at least one instance (they come first)
        header[6] = total;                             // instances this frame needs
        header[3] = needed;                            // ... and the capacity that holds it (>= [6]: the record pool and the key buckets
                                                       // are sized from the capacity too) -- for the host's capacity policy
        header[9] = overflow ? 0u : pool_total;        // cells of the record pool in use: what merge_gather clears of the validity bits
        header[7] = s_bucket[32 - light_log2];         // heavy tiles: list length >= 2^light_log2 (classes clz <= 31 - log2)
        header[2] = *flags_acc | (overflow ? ERRFLAG_OVERFLOW : 0u) | (forward_only ? ERRFLAG_FORWARD_ONLY : 0u);   // (the preprocess kernel's flags: it finished before this one)
        if (flags_acc != header + 2) {                                      // frame state: zero again for the next forward ...
            *flags_acc = 0u;
            if (overflow) flags_acc[FS_DROPPED_WORD] += 1u;                 // ... except its STICKY count of frames that rendered nothing
        }
    }
    __syncthreads();
    {
        uint32_t o = overflow ? 0u : off_first;
        for (int i = b; i < e; i++) {
            const uint32_t v = overflow ? 0u : tile_count[i];
            const uint32_t rank = atomicAdd(&s_bucket[v ? (uint32_t)__clz((int)v) : 32u], 1u);
            tile_order[rank] = (uint32_t)i;
            work_table[rank] = make_uint4((uint32_t)i, o, o + v, 0u);       // {tile, list start, list end}: an item's start-up in one load
            o += v;
        }
    }
}

// One 1024-thread block: ranges[t] = [start,end) from the exclusive scan of the per-tile histogram, chunk_base[t] = number
// of sort chunks in front of tile t; header[0] = R (num_rendered), header[1] = longest tile list, header[4] = total chunks.
// (The per-Gaussian offsets are produced by the preprocess kernel.)
__global__ void __launch_bounds__(1024)
scan_kernel(int T, const uint32_t* __restrict__ tile_count, uint2* __restrict__ ranges, uint32_t* __restrict__ chunk_base,
            uint32_t* __restrict__ tile_order, uint32_t* __restrict__ header, uint32_t capacity, int light_log2,
            int n_groups,
            uint32_t* __restrict__ flags_acc, uint32_t* __restrict__ queues, uint4* __restrict__ work_table,
            const uint32_t* __restrict__ group_rtot, uint32_t* __restrict__ group_rbase, uint32_t pool_cap, int forward_only)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_max;
    __shared__ uint32_t s_bucket[34];
    scan_outputs<1024>(T, tile_count, ranges, chunk_base, tile_order, header, capacity, light_log2, s_wave, &s_max, s_bucket,
                       n_groups, flags_acc, queues, work_table, group_rtot, group_rbase, pool_cap, 0u, forward_only != 0);
}

// duplicateWithKeys equivalent (rasterizer_impl.cu:70-111) of the SYNCHRONOUS path (and of frames with more tiles than the LDS
// histogram holds): the scan kernel ran in front, the tiles' ranges are known.  A block reserves, per tile, a contiguous run of slots
// with ONE returning global atomic (after counting its own instances in LDS) and hands the slots out with LDS atomics.
// (Rounds 2-4 also ran it on the asynchronous path, with the scan folded in as one extra block; since round 5 the preprocess kernel
// writes the keys there itself: preprocess.hip, scatter mode.)
// FOLD_SCAN (the asynchronous forward whose keys are NOT bucketed by the preprocess kernel: MOSS_FORWARD_ONLY renders, whose lean binning
// buffer has no room for buckets; rounds 2-4: every asynchronous forward): there is no scan kernel in front of this one.  Every block
// turns the tile histogram into the tiles' start offsets ITSELF (T <= 8192 counts: one coalesced read and a block scan, overlapped
// with its instance counting), and one extra block at the end of the grid writes what the later kernels need (scan_outputs) -- the
// scan costs no launch of its own (a 1024-thread, one-block kernel: 5.5-8.8 us) and no memory round trip between the two kernels.
__global__ void __launch_bounds__(256)
scatter_kernel(int P, int gx, int T, GeomView g, uint2* __restrict__ ranges, uint32_t* __restrict__ tile_cursor,
               uint64_t* __restrict__ keys, int lds_hist, uint32_t* __restrict__ header,
               unsigned long long* __restrict__ stamps /* diagnostics: 8 words per block, else NULL */,
               int fold_scan, const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ chunk_base, uint32_t* __restrict__ tile_order,
               uint32_t capacity, int light_log2, uint32_t* __restrict__ flags_acc, uint32_t* __restrict__ queues,
               uint4* __restrict__ work_table, uint32_t pool_cap, int forward_only)
{
#define SSTAMP(i) if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime()   /* 100 MHz, device-wide */
    SSTAMP(0);
    extern __shared__ uint32_t s_mem[];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_max;
    __shared__ uint32_t s_bucket[34];
    int n_blocks = (int)gridDim.x;                     // blocks that scatter
    if (fold_scan) {
        n_blocks--;
        if ((int)blockIdx.x == n_blocks) {
            scan_outputs<256>(T, tile_count, ranges, chunk_base, tile_order, header, capacity, light_log2, s_wave, &s_max, s_bucket,
                              (P + 255) / 256, flags_acc, queues, work_table, g.group_rtot, g.group_rbase, pool_cap, 0u, forward_only != 0);
            SSTAMP(5);
            return;
        }
    } else if (header[0] == 0u) return;                // nothing rendered (or capacity overflow: see scan_kernel)
    uint32_t* s_cnt = s_mem;
    uint32_t* s_base = s_mem + T;
    // rectangle (geo[3]) and depth (geo[2].w) of the two Gaussians of a trip, requested together, unconditionally
    auto fetch2 = [&](int base, float4& a, float4& b, float& da, float& db) {
        const size_t i0 = (size_t)min(base + (int)threadIdx.x, P - 1), i1 = (size_t)min(base + (int)threadIdx.x + n_blocks * (int)blockDim.x, P - 1);
        a = g.geo[4 * i0 + 3]; b = g.geo[4 * i1 + 3]; da = g.geo[4 * i0 + 2].w; db = g.geo[4 * i1 + 2].w;
    };
    float4 pre_gd[2]; float pre_dep[2];
    fetch2((int)(blockIdx.x * blockDim.x), pre_gd[0], pre_gd[1], pre_dep[0], pre_dep[1]);
    if (lds_hist) {
        // (fold_scan) this thread's share of the histogram: requested now, summed after the counting pass
        const int per = (T + 255) / 256, tb = (int)threadIdx.x * per, te = min(T, tb + per);
        uint32_t cnt4[4] = { 0u, 0u, 0u, 0u };
        if (fold_scan && per <= 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) cnt4[u] = tile_count[min(tb + u, T - 1)];
        }
        for (int i = threadIdx.x; i < T; i += blockDim.x) s_cnt[i] = 0;
        __syncthreads();
        // (two Gaussians per thread per trip, both rectangles requested before the first is walked: one memory round trip per trip;
        // the first trip's -- with the default grid the only one -- were requested at the top and serve the scatter pass as well)
        for (int base = blockIdx.x * blockDim.x; base < P; base += 2 * n_blocks * blockDim.x) {  // whole waves stay converged
            float4 gd0, gd1; float d0_, d1_;
            if (base == (int)(blockIdx.x * blockDim.x)) { gd0 = pre_gd[0]; gd1 = pre_gd[1]; }
            else fetch2(base, gd0, gd1, d0_, d1_);
            const int idx0 = base + (int)threadIdx.x, idx1 = idx0 + n_blocks * (int)blockDim.x;
            const uint2 r0 = idx0 < P ? make_uint2(__float_as_uint(gd0.x), __float_as_uint(gd0.y)) : make_uint2(0u, 0u);
            const uint2 r1 = idx1 < P ? make_uint2(__float_as_uint(gd1.x), __float_as_uint(gd1.y)) : make_uint2(0u, 0u);
            wave_for_each_tile(r0, gx, 0ull, [&](int t, uint64_t) { atomicAdd(&s_cnt[t], 1u); });
            wave_for_each_tile(r1, gx, 0ull, [&](int t, uint64_t) { atomicAdd(&s_cnt[t], 1u); });
        }
        SSTAMP(1);                                           // counted
        if (fold_scan) {
            uint32_t sum = 0u;
            if (per <= 4) { for (int u = 0; u < 4; u++) sum += tb + u < te ? cnt4[u] : 0u; }
            else for (int i = tb; i < te; i++) sum += tile_count[i];
            uint32_t total;
            uint32_t off = block_scan<256>(sum, s_wave, total);      // (its barriers also end the counting pass)
            if (total == 0u || total > capacity) return;             // nothing rendered / capacity overflow (flag: the scan block)
            if (per <= 4) { for (int u = 0; u < 4; u++) if (tb + u < te) { s_base[tb + u] = off; off += cnt4[u]; } }
            else for (int i = tb; i < te; i++) { s_base[i] = off; off += tile_count[i]; }
        }
        __syncthreads();
        SSTAMP(2);                                           // tile starts known
        // One returning atomic per (block, non-empty tile) reserves the block's run in the tile's bucket.  Four tiles per thread at a
        // time, as BUFFER atomics whose offset is out of range for an empty tile (dropped without a memory request, returns 0):
        // no branch around the atomic, so the four are in flight together -- inside `if (c)` each one was followed by a full
        // s_waitcnt (a device-scope returning atomic is ~2 us) and a thread's tiles were reserved one round trip after the other.
        {
            const __amdgpu_buffer_rsrc_t rs_cur = __builtin_amdgcn_make_buffer_rsrc((void*)tile_cursor, 0, 0xffffff00u, 0x00020000u);
            for (int b0 = 0; b0 < T; b0 += 4 * (int)blockDim.x) {
                uint32_t c[4], start[4], old[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = b0 + u * (int)blockDim.x + (int)threadIdx.x;
                    c[u] = i < T ? s_cnt[i] : 0u;
                    start[u] = fold_scan ? s_base[min(i, T - 1)] : ranges[min(i, T - 1)].x;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = b0 + u * (int)blockDim.x + (int)threadIdx.x;
                    old[u] = lds_hist == 2 ? 0u : (uint32_t)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32((int)c[u], rs_cur, c[u] ? (uint32_t)i * 4u : 0xfffffffcu /* out of range AND dword-aligned: a misaligned atomic faults before the range check */, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = b0 + u * (int)blockDim.x + (int)threadIdx.x;
                    if (c[u]) { s_base[i] = start[u] + old[u]; s_cnt[i] = 0; }
                }
            }
        }
        __syncthreads();
        SSTAMP(3);                                           // runs reserved
    }
    for (int base = blockIdx.x * blockDim.x; base < P; base += 2 * n_blocks * blockDim.x) {
        int idx[2]; uint2 r[2]; uint64_t key[2];
        float4 gd[2]; float dep[2];
#pragma unroll
        for (int u = 0; u < 2; u++) idx[u] = base + (int)threadIdx.x + u * n_blocks * (int)blockDim.x;
        if (base == (int)(blockIdx.x * blockDim.x)) { gd[0] = pre_gd[0]; gd[1] = pre_gd[1]; dep[0] = pre_dep[0]; dep[1] = pre_dep[1]; }
        else fetch2(base, gd[0], gd[1], dep[0], dep[1]);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            r[u] = idx[u] < P ? make_uint2(__float_as_uint(gd[u].x), __float_as_uint(gd[u].y)) : make_uint2(0u, 0u);
            const bool any = idx[u] < P && (r[u].y & 0xffffu) > (r[u].x & 0xffffu) && (r[u].y >> 16) > (r[u].x >> 16);
            key[u] = any ? (((uint64_t)__float_as_uint(dep[u]) << 32) | (uint32_t)idx[u]) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            wave_for_each_tile(r[u], gx, key[u], [&](int t, uint64_t k) {
                uint32_t pos;
                if (lds_hist) pos = s_base[t] + atomicAdd(&s_cnt[t], 1u);
                else pos = ranges[t].x + atomicAdd(&tile_cursor[t], 1u);
                keys[pos] = k;
            });
        }
    }
    SSTAMP(4);                                               // keys written
#undef SSTAMP
}

// Which tile owns sort chunk c: the tile t with chunk_base[t] <= c < chunk_base[t] + ceil(n_t / CHUNK).  Every thread of the workgroup
// tests the tiles tid, tid + blockDim, ... -- two independent loads each, ONE memory round trip for the workgroup (a binary search by
// thread 0 was ten DEPENDENT loads, ~6 us, with 1023 threads waiting at the barrier behind it).  The finder also leaves the tile's
// range and chunk base in LDS, and the device-side chunk count (header[4]) is loaded in the same round trip: the prologue of a sort
// workgroup is ONE round trip, not four (header -> lookup -> range of the tile -> keys were 4 x ~2 us of its ~12).  Ends with a barrier.
struct ChunkOwner { int tile; uint32_t start, end, cbase, n_chunks, c; };
// `chunk_of(n_chunks)` maps this workgroup's turn to a chunk index (or ~0u: nothing to do); it is evaluated AFTER the tile loads
// have been issued, so the header word and the tile table arrive in the same round trip.
template <typename F>
__device__ __forceinline__ void find_chunk_tile(int T, F&& chunk_of, const uint2* __restrict__ ranges, const uint32_t* __restrict__ chunk_base,
                                                const uint32_t* __restrict__ header, ChunkOwner* s_own)
{
    int t = (int)threadIdx.x;
    uint32_t cb = chunk_base[min(t, T - 1)];
    uint2 rg = ranges[min(t, T - 1)];
    const uint32_t n_chunks = header[4];
    const uint32_t c = chunk_of(n_chunks);
    if (threadIdx.x == 0) { s_own->n_chunks = n_chunks; s_own->c = c; }
    for (;;) {
        const uint32_t nch = (rg.y - rg.x + CHUNK - 1) / CHUNK;
        if (t < T && cb <= c && c < cb + nch) { s_own->tile = t; s_own->start = rg.x; s_own->end = rg.y; s_own->cbase = cb; }
        t += (int)blockDim.x;
        if (t >= T) break;
        cb = chunk_base[t]; rg = ranges[t];
    }
    __syncthreads();
}

// ---- the chunk sort's compare-exchange network.  Thread tid holds one 64-bit key in registers; the partner of step j is tid ^ j.
// For j < 64 the partner is in the same wave and is reached WITHOUT LDS: quad permutes (j = 1, 2), half-row mirror + quad reverse
// (j = 4: i ^ 7 ^ 3), row rotation by 8 (j = 8), v_permlane16_swap / v_permlane32_swap (j = 16, 32).  (Round 2's stamps: the
// ds_bpermute version of this network was 21k of a full chunk's 25k cycles -- 380 cycles per step with the LDS round trip inside.)
// Steps with j >= 64 go through LDS, double-buffered: ONE barrier per step.
#define MOSS_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), (ctrl), 0xf, 0xf, true))
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, uint32_t lane)
{
    if constexpr (J == 1) return MOSS_DPP(v, 0xB1);                               // quad_perm:[1,0,3,2]
    else if constexpr (J == 2) return MOSS_DPP(v, 0x4E);                          // quad_perm:[2,3,0,1]
    else if constexpr (J == 4) { const uint32_t m = MOSS_DPP(v, 0x141); return MOSS_DPP(m, 0x1B); }   // row_half_mirror, quad_perm:[3,2,1,0]
    else if constexpr (J == 8) return MOSS_DPP(v, 0x128);                         // row_ror:8
    else if constexpr (J == 16) {
        // swaps the odd rows of the first operand with the even rows of the second: r[0] = {v0, v0, v2, v2}, r[1] = {v1, v1, v3, v3}
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16u) ? r[0] : r[1];
    } else {
        static_assert(J == 32, "intra-wave partner distance");
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);            // r[0] = {lower, lower}, r[1] = {upper, upper}
        return (lane & 32u) ? r[0] : r[1];
    }
}
#undef MOSS_DPP

// Element i of a chunk lives in thread i / KPT, register i % KPT: steps with partner distance J < KPT are compare-exchanges between a
// thread's own registers; J / KPT < 64: the partner thread is in the same wave (lane_xor); larger: through LDS.
template <int K, int J>
__device__ __forceinline__ void network_steps(uint64_t (&key)[KPT], uint32_t tid, uint64_t (*s_buf)[CHUNK], int& p)
{
    if constexpr (J < KPT) {
#pragma unroll
        for (int e = 0; e < KPT; e++) {
            if ((e & J) == 0) {
                const bool asc = ((tid * (uint32_t)KPT + (uint32_t)e) & (uint32_t)K) == 0u;
                const uint64_t a = key[e], b = key[e | J];
                const bool sw = (a > b) == asc;
                key[e] = sw ? b : a; key[e | J] = sw ? a : b;
            }
        }
    } else if constexpr (J / KPT >= 64) {
#pragma unroll
        for (int e = 0; e < KPT; e++) s_buf[p][tid * KPT + e] = key[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < KPT; e++) {
            const uint32_t i = tid * (uint32_t)KPT + (uint32_t)e;
            const uint64_t other = s_buf[p][i ^ (uint32_t)J];
            const bool take_min = ((i & (uint32_t)J) == 0u) == ((i & (uint32_t)K) == 0u);     // lower partner of an ascending run
            const bool lt = key[e] < other;
            key[e] = (lt == take_min) ? key[e] : other;
        }
        p ^= 1;                                            // the next cross-wave step writes the other buffer: no second barrier
    } else {
#pragma unroll
        for (int e = 0; e < KPT; e++) {
            const uint32_t lo = lane_xor<J / KPT>((uint32_t)key[e], tid), hi = lane_xor<J / KPT>((uint32_t)(key[e] >> 32), tid);
            const uint64_t other = ((uint64_t)hi << 32) | lo;
            const uint32_t i = tid * (uint32_t)KPT + (uint32_t)e;
            const bool take_min = ((i & (uint32_t)J) == 0u) == ((i & (uint32_t)K) == 0u);
            const bool lt = key[e] < other;
            key[e] = (lt == take_min) ? key[e] : other;
        }
    }
    if constexpr (J > 1) network_steps<K, J / 2>(key, tid, s_buf, p);
}
template <int K>
__device__ __forceinline__ void network_phases(uint64_t (&key)[KPT], uint32_t tid, uint32_t npad, uint64_t (*s_buf)[CHUNK], int& p)
{
    if constexpr (K > 2) network_phases<K / 2>(key, tid, npad, s_buf, p);
    if ((uint32_t)K <= npad) network_steps<K, K / 2>(key, tid, s_buf, p);          // (wave-uniform)
}

// Stage A of the sort: one workgroup per CHUNK of a tile's bucket (a tile of n entries has ceil(n/CHUNK) chunks), sorted and
// written back in place.  Every workgroup has at most 1024 keys, so there is no long-tile tail here.
// One key per thread, in REGISTERS: a bitonic network (above).  Threads past n hold the maximum key, and the network stops at the
// padded size, so short chunks run few steps.
// The grid is bounded (launch_tile_sort) and a workgroup loops over the chunks c = blockIdx.x, blockIdx.x + gridDim.x, ...: the
// host-side chunk count is only an upper bound when R stays on the device, and a thousand empty 1024-thread workgroups are not free.
// (80 SGPRs: the CU admits waves per SIMD by the scalar file too -- 800 per SIMD in blocks of 16 + 16 -- and TWO 1024-thread workgroups,
// eight waves per SIMD, only fit up to 80; with the 106 the unrolled network asked for, 84 of cfg3's 340 chunks waited for a first-round
// workgroup to leave: scripts/sort_stamps.py, starts at 6.4 us)
// SELF-SCAN (key_stride != 0: the asynchronous path, keys in per-tile buckets written by the preprocess kernel): there is no scan in
// front of this kernel.  Every workgroup derives its chunk table from the tile counts itself -- thread i owns the tiles i PER ..
// i PER + PER - 1 (PER = ceil(T / 1024) <= 8: T <= MAX_LDS_TILES), one block scan of their chunk counts, kept in registers for every
// turn of the workgroup -- and ONE EXTRA BLOCK at the end of the grid writes what the later kernels need (ranges, chunk bases, tile
// order, work table, header, group bases: scan_outputs), so the scan costs no launch of its own.  A frame that overflowed (more
// instances than the capacity, or a tile that outgrew its bucket) is seen by every workgroup alike: nothing is sorted.
// SORT_PER: tiles per thread of the self-scan, at most -- 1 (T <= 1024: a 512 x 512 frame; the instantiation that must keep its two
// workgroups per CU: 64 VGPRs) or MAX_LDS_TILES / SORT_THREADS = 8
template <int SORT_PER>
__global__ void __launch_bounds__(SORT_THREADS) __attribute__((amdgpu_num_sgpr(80)))
chunk_sort_kernel(int T, uint2* __restrict__ ranges, uint32_t* __restrict__ chunk_base, uint64_t* __restrict__ keys,
                  uint32_t* __restrict__ header, unsigned long long* __restrict__ stamps /* diagnostics: 8 words per workgroup, else NULL */,
                  uint4* __restrict__ frame_state, uint32_t frame_state_n16,
                  uint32_t key_stride, const uint32_t* __restrict__ tile_count, uint32_t capacity, int light_log2,
                  uint32_t* __restrict__ tile_order, int n_groups,
                  uint32_t* __restrict__ flags_acc, uint32_t* __restrict__ queues, uint4* __restrict__ work_table,
                  const uint32_t* __restrict__ group_rtot, uint32_t* __restrict__ group_rbase, uint32_t pool_cap)
{
    __shared__ __attribute__((aligned(16))) uint64_t s_keys[2][CHUNK];
    __shared__ ChunkOwner s_own;
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_max;
    __shared__ uint32_t s_bucket[34];
    const uint32_t tid = threadIdx.x;
    const bool self_scan = key_stride != 0u;
    uint32_t grid = gridDim.x, wg = blockIdx.x;              // workgroups that sort, and this one's index among them
    if (self_scan) {
        // (block 0: dispatched FIRST.  As the last block of a grid that fills every workgroup slot of the device it waited for a sort
        // workgroup to leave and then ran alone: 12.1 -> 15.4 us)
        grid--; wg--;
        if (blockIdx.x == 0) {                               // the scan block
            if (stamps && tid == 0) stamps[5] = __builtin_amdgcn_s_memrealtime();
            scan_outputs<1024>(T, tile_count, ranges, chunk_base, tile_order, header, capacity, light_log2, s_wave, &s_max, s_bucket,
                               n_groups, flags_acc, queues, work_table, group_rtot, group_rbase, pool_cap, key_stride);
            if (stamps && tid == 0) stamps[7] = __builtin_amdgcn_s_memrealtime();
            return;
        }
    } else {
        // the caller's frame state (tile histogram, cursors, flag word) is dead once the scatter kernel has ended: all-zero again for the
        // next forward (every workgroup of the grid takes a slice, before anything can make it leave).  (Self-scan: this kernel still
        // READS the counts -- merge_gather_kernel does the zeroing there.)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < frame_state_n16; i += gridDim.x * blockDim.x) frame_state[i] = make_uint4(0u, 0u, 0u, 0u);
    }
#define KSTAMP(i) if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime()
#define RSTAMP(i) if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime()   /* 100 MHz, device-wide */
    KSTAMP(0); RSTAMP(5);
    // self-scan: this thread's tiles, their counts and chunk bases
    uint32_t my_cnt[SORT_PER], my_cb[SORT_PER], n_chunks_dev = 0u;
    const int per = (T + SORT_THREADS - 1) / SORT_THREADS;
    if (self_scan) {
        // (a count above the bucket size means the preprocess kernel dropped keys: the scan block flags the frame and empties every
        // range; here the count is clamped so that whatever is sorted stays inside the bucket.  Likewise a frame with more instances
        // than the capacity: the buckets themselves are always in bounds, only the compact arrays behind the scan are not -- and
        // nobody writes those in a flagged frame.  ONE block scan, not three reductions: every barrier of 16 waves is ~0.3 us)
        uint32_t nch = 0u;
#pragma unroll
        for (int u = 0; u < SORT_PER; u++) {
            const int t = (int)tid * per + u;
            my_cnt[u] = (u < per && t < T) ? min(tile_count[t], key_stride) : 0u;
            nch += (my_cnt[u] + CHUNK - 1) / CHUNK;
        }
        uint32_t cb = block_scan<1024>(nch, s_wave, n_chunks_dev);
        if (n_chunks_dev == 0u) return;                      // nothing rendered
#pragma unroll
        for (int u = 0; u < SORT_PER; u++) { my_cb[u] = cb; cb += (my_cnt[u] + CHUNK - 1) / CHUNK; }
    }
    for (uint32_t c = wg;; c += grid) {
        if (self_scan) {
            if (c >= n_chunks_dev) return;
#pragma unroll
            for (int u = 0; u < SORT_PER; u++) {
                const uint32_t nch = (my_cnt[u] + CHUNK - 1) / CHUNK;
                if (my_cb[u] <= c && c < my_cb[u] + nch) {
                    const int t = (int)tid * per + u;
                    s_own.tile = t; s_own.start = (uint32_t)t * key_stride; s_own.end = s_own.start + my_cnt[u]; s_own.cbase = my_cb[u];
                }
            }
            if (tid == 0) { s_own.n_chunks = n_chunks_dev; s_own.c = c; }
            __syncthreads();
        } else {
            find_chunk_tile(T, [&](uint32_t) { return c; }, ranges, chunk_base, header, &s_own);
        }
        const uint32_t n_chunks = s_own.n_chunks;
        if (c >= n_chunks) return;
        KSTAMP(1);
        const uint32_t first = s_own.start + (c - s_own.cbase) * CHUNK;
        const uint32_t n = min((uint32_t)CHUNK, s_own.end - first);
        uint64_t* gk = keys + first;
        uint32_t npad = 64 * KPT;                          // at least one wave's worth: the intra-wave steps need no branches
        while (npad < n) npad <<= 1;
        const bool last_turn = c + grid >= n_chunks;       // (no further lookup just to find that out: it is a memory round trip)
        if (tid * KPT >= npad) {
            // whole waves (npad is a multiple of 64 KPT) with nothing to sort.  On the workgroup's last turn they leave -- the barriers
            // below count the waves that are still alive (gfx9 s_barrier semantics; 16-wave barriers are what a short chunk's network
            // would otherwise pay) -- on earlier turns they only keep the barrier count: one per cross-wave step.
            if (last_turn) return;
            for (uint32_t k = 128 * KPT; k <= npad; k <<= 1)
                for (uint32_t j = k >> 1; j >= 64u * KPT; j >>= 1) __syncthreads();
            __syncthreads();
            continue;
        }
        uint64_t key[KPT];                                 // (a thread's keys are neighbours in memory: one 16-byte load)
#pragma unroll
        for (int e = 0; e < KPT; e++) key[e] = tid * KPT + e < n ? gk[tid * KPT + e] : ~0ull;
        if (stamps && tid == 0) { stamps[(size_t)blockIdx.x * 8 + 2] = key[0] ? __builtin_amdgcn_s_memtime() : 1ull; stamps[(size_t)blockIdx.x * 8 + 6] = n; }
        int p = 0;
        network_phases<CHUNK>(key, tid, npad, s_keys, p);
        KSTAMP(3);
#pragma unroll
        for (int e = 0; e < KPT; e++) if (tid * KPT + e < n) gk[tid * KPT + e] = key[e];
        KSTAMP(4); RSTAMP(7);
        if (last_turn) return;
        __syncthreads();                                   // s_own is rewritten by the next round's lookup
    }
}

// Stage B: one workgroup per chunk again, one thread per instance.  An instance's final rank inside its tile = its rank inside its own
// (sorted) chunk + the number of smaller keys in each of the tile's OTHER chunks (keys (depth_bits, id) are unique).  The other
// chunks are brought into LDS TOGETHER (up to MERGE_OC at a time: one coalesced 8-byte load per thread and chunk, all in flight at
// once) and searched there side by side -- independent binary searches whose LDS reads overlap.  (Round 1 searched in global memory:
// 20-40 dependent L2 round trips per instance; round 2 brought the chunks into LDS one after the other: a load round trip and two
// barriers per sibling chunk, 4 x ~2.5 us for a five-chunk tile.)  The same thread then emits everything that is per-instance: the
// sorted id, the 48-byte record the blend kernels stream, the block mask, and the Gaussian -> instance back-pointer used by the
// backward gather.
constexpr int MERGE_OC = 6;                                // sibling chunks searched per round (48 KB of LDS)
constexpr int SEARCH_STEPS = 11;                           // halvings of a binary search over one chunk: log2(CHUNK) + 1
static_assert((1 << (SEARCH_STEPS - 1)) == CHUNK, "SEARCH_STEPS = log2(CHUNK) + 1");
// MERGE_THREADS threads = a PART of a chunk's instances per workgroup (512: two workgroups per chunk).  One 1024-thread workgroup per
// chunk put two workgroups on 84 of the 256 CUs for cfg3's 340 chunks and one on the rest: the kernel ended with the double-loaded CUs
// (workgroup ends 8 us median, 17 us last).  Halves spread 680 workgroups three to a CU at most; each loads the sibling chunks itself.
constexpr int MERGE_THREADS = 512, MERGE_PARTS = CHUNK / MERGE_THREADS;
__global__ void __launch_bounds__(MERGE_THREADS) __attribute__((amdgpu_num_sgpr(80)))     // (see chunk_sort_kernel)
merge_gather_kernel(const uint32_t* __restrict__ header, int gx, int T, GeomView g, const uint2* __restrict__ ranges,
                    const uint32_t* __restrict__ chunk_base, const uint64_t* __restrict__ keys,
                    uint32_t* __restrict__ point_list,
                    float4* __restrict__ inst_rec, uint32_t* __restrict__ cell_valid, uint16_t* __restrict__ inst_bmask,
                    unsigned long long* __restrict__ stamps /* diagnostics: 8 words per workgroup (after the sort's), else NULL */,
                    uint32_t key_stride /* != 0: tile t's keys start at t * key_stride (buckets), else at its range */,
                    uint4* __restrict__ frame_state, uint32_t frame_state_n16,
                    int forward_only /* MOSS_FORWARD_ONLY: no record pool -- no validity bits to clear, no cell word in the record */)
{
    __shared__ __attribute__((aligned(16))) uint64_t s_keys[MERGE_OC][CHUNK];
    __shared__ ChunkOwner s_own;
    const uint32_t tid = threadIdx.x;
    // bucketed keys (asynchronous path): the sort kernel read the tile counts of the caller's frame state itself, so the state is
    // re-zeroed for the next forward HERE (every workgroup of the grid takes a slice, before anything can make it leave)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < frame_state_n16; i += gridDim.x * blockDim.x) frame_state[i] = make_uint4(0u, 0u, 0u, 0u);
    // no gradient record yet: one bit per cell of the record pool the frame uses (header[9], written by the scan block; set by the
    // backward blend).  (Rounds 2-4: a 4-byte mask word per instance, zeroed by the instance's own thread below.)
    if (!forward_only) {
        const uint32_t n_words = (header[9] + 31u) / 32u + 1u;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) cell_valid[i] = 0u;
    }
    KSTAMP(0); RSTAMP(5);
    for (uint32_t it = blockIdx.x;; it += gridDim.x) {
    // XCD-aware chunk order: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), chunks are in tile order, and neighbouring
    // tiles gather the same Gaussians' 64-byte records: XCD k takes the k-th contiguous eighth of the chunks (counted from the
    // device-side total), so a Gaussian's record is fetched into one or two L2s instead of all eight.
    const uint32_t NONE = 0xffffffffu;
    // turn `it` = (chunk turn it / MERGE_PARTS on this XCD ... , part it % MERGE_PARTS)?  No: the XCD of a workgroup is blockIdx % 8, so
    // the part index sits ABOVE the XCD digit: it = 8 * (MERGE_PARTS * nth + part) + xcd.
    const uint32_t part = (it / 8u) % (uint32_t)MERGE_PARTS, nth_u = (it / 8u) / (uint32_t)MERGE_PARTS;
    find_chunk_tile(T, [&](uint32_t n_chunks) -> uint32_t {
        const int per_xcd = (int)n_chunks / 8, extra = (int)n_chunks % 8, xcd = (int)it % 8, nth = (int)nth_u;
        return nth < per_xcd + (xcd < extra ? 1 : 0) ? (uint32_t)(xcd * per_xcd + min(xcd, extra) + nth) : NONE;
    }, ranges, chunk_base, header, &s_own);
    const uint32_t c = s_own.c;
    if (c == NONE) return;                                    // (turns only grow: later ones are past this XCD's share as well)
    KSTAMP(1);
    const bool last_turn = ((it + gridDim.x) / 8u) / (uint32_t)MERGE_PARTS >= s_own.n_chunks / 8 + ((it % 8) < s_own.n_chunks % 8 ? 1u : 0u);
    const uint32_t tile = (uint32_t)s_own.tile;
    const uint2 rg = make_uint2(s_own.start, s_own.end);
    const uint32_t n = rg.y - rg.x, nch = (n + CHUNK - 1) / CHUNK, own = c - s_own.cbase;
    const uint32_t n_own = min((uint32_t)CHUNK, n - own * CHUNK);
    // where the tile's keys are: its bucket (asynchronous path) or its range; kq = the tile's key array, entries 0 .. n - 1
    const uint64_t* const kq = keys + (key_stride != 0u ? (size_t)tile * key_stride : (size_t)rg.x);
    const uint32_t my = part * (uint32_t)MERGE_THREADS + tid; // this thread's instance of the chunk
    const bool mine = my < n_own;
    const uint64_t key_ld = kq[min(own * CHUNK + my, n - 1u)];
    uint32_t rank = my;                                       // rank inside the own (sorted) chunk
    // Sibling group s0: chunks s0 .. s0 + MERGE_OC - 1 of the tile's nch - 1 OTHER chunks (unconditional, clamped loads: all in
    // flight together).
    uint32_t on[MERGE_OC];
    uint64_t v[MERGE_OC][MERGE_PARTS];
    auto load_group = [&](uint32_t s0) {
#pragma unroll
        for (int q = 0; q < MERGE_OC; q++) {
            const uint32_t si = s0 + (uint32_t)q;             // sibling index: the tile's chunks without the own one
            const uint32_t oc = si + (si >= own ? 1u : 0u);
            const uint32_t ofirst = oc * CHUNK;
            on[q] = si + 1 < nch ? min((uint32_t)CHUNK, n - ofirst) : 0u;
#pragma unroll
            for (int u = 0; u < MERGE_PARTS; u++)
                v[q][u] = kq[on[q] ? min(ofirst + (uint32_t)u * MERGE_THREADS + tid, n - 1u) : own * CHUNK];
        }
    };
    if (part * (uint32_t)MERGE_THREADS >= n_own) {            // (a short chunk has no second part: wave-uniform)
        if (last_turn) return;
        __syncthreads();
        continue;
    }
    // Request order = arrival order: the own key, the first group of siblings, and -- as soon as the key is there (its low word names
    // the Gaussian) -- the Gaussian's 64-byte record and its group's slot base.  Those last loads stay in flight under the LDS fill
    // and the searches; after the searches they were one more memory round trip on the critical path of every workgroup of a
    // multi-chunk tile.  (A thread without an instance re-reads the record of the chunk's last key: a valid address.)
    float4 ga, gb, gc, gd; uint32_t slot_base, id;
    auto load_gaussian = [&]() {
        id = (uint32_t)key_ld;
        asm volatile("" : "+v"(id));                          // (opaque: otherwise the address arithmetic -- and with it the wait for
                                                              // the key -- is hoisted in front of the siblings' loads)
        const float4* const gsrc = g.geo + 4 * (size_t)id;    // the Gaussian's one 64-byte record
        ga = gsrc[0]; gb = gsrc[1]; gc = gsrc[2]; gd = gsrc[3]; slot_base = g.group_rbase[id >> 8];      // (where the group's CELL runs start)
    };
    // the group's chunks into LDS, searched side by side
    auto search_group = [&](uint32_t s0) {
        uint32_t k_lo = (uint32_t)key_ld, k_hi = (uint32_t)(key_ld >> 32);
        asm volatile("" : "+v"(k_lo), "+v"(k_hi));            // (opaque, like the id: the key is first NEEDED here, behind the requests)
        const uint64_t key = mine ? (((uint64_t)k_hi << 32) | k_lo) : ~0ull;
        __syncthreads();                                      // the previous group's readers are done
#pragma unroll
        for (int q = 0; q < MERGE_OC; q++) {
#pragma unroll
            for (int u = 0; u < MERGE_PARTS; u++) s_keys[q][u * MERGE_THREADS + tid] = v[q][u];   // (unconditional: slots past on[q] are never read for a decision, and a store
                                                                                                  // inside a branch pulls its load in with it -- behind a wait for ALL loads)
        }
        __syncthreads();
        const int ns = (int)min((uint32_t)MERGE_OC, nch - 1u - s0);      // siblings in this group (wave-uniform)
        uint32_t lo[MERGE_OC], hi[MERGE_OC];
#pragma unroll
        for (int q = 0; q < MERGE_OC; q++) { lo[q] = 0u; hi[q] = on[q]; }
        for (int step = 0; step < SEARCH_STEPS; step++) {      // 2^10 keys: eleven halvings; the siblings' searches run side by side
#pragma unroll
            for (int q = 0; q < MERGE_OC; q++) {
                if (q < ns) {                                 // (scalar branch: absent siblings cost nothing)
                    // branch-free halving: a finished search (lo == hi) re-reads a valid slot and keeps its bounds
                    const uint32_t mid = min((lo[q] + hi[q]) >> 1, (uint32_t)CHUNK - 1u);
                    const bool less = s_keys[q][mid] < key && lo[q] < hi[q];
                    const bool open = lo[q] < hi[q];
                    lo[q] = less ? mid + 1u : lo[q];
                    hi[q] = (open && !less) ? mid : hi[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < MERGE_OC; q++) rank += q < ns ? lo[q] : 0u;     // keys < key in that sorted chunk
    };
    if (nch > 1u) {                                           // (wave-uniform)
        load_group(0u);
        __builtin_amdgcn_sched_barrier(0);                    // (this order, in ONE basic block: left alone, the compiler requests the
        load_gaussian();                                      // record first and sinks the siblings' loads behind the wait for the key)
        __builtin_amdgcn_sched_barrier(0);
        search_group(0u);
    } else {
        load_gaussian();
    }
    for (uint32_t s0 = MERGE_OC; s0 + 1 < nch; s0 += MERGE_OC) { load_group(s0); search_group(s0); }   // tiles of more than 7 chunks
    if (stamps && tid == 0) { stamps[(size_t)blockIdx.x * 8 + 2] = rank + 1u ? __builtin_amdgcn_s_memtime() : 1ull; stamps[(size_t)blockIdx.x * 8 + 6] = nch; }
    if (mine) {
    const uint32_t pos = rg.x + rank;
    point_list[pos] = id;
    float4* rec = inst_rec + 3 * (size_t)pos;
    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + 3] = ga.x == 12345.678f ? 1ull : __builtin_amdgcn_s_memtime();
    const uint2 r = make_uint2(__float_as_uint(gd.x), __float_as_uint(gd.y));
    const int tx = (int)tile % gx, ty = (int)tile / gx;
    // Where the instance's gradient-record cells are (common.h: box_cells): the Gaussian's run in the pool (group base + its start in
    // the group, the geometry record's last word) + the cells in front of this tile, in closed form from the box.  It travels in the
    // record's third word, packed with the box's width in the tile, so that the backward blend finds the cell of a block with three
    // integer operations (pack_cell_word) and the per-Gaussian gather needs no table at all: it sums its run.
    uint32_t cell_word = 0u;                                  // (forward only: nobody will look for the instance's cells)
    if (!forward_only) {
        const BoxCells bc = box_cells(ga.x, ga.y, ga.z, ga.w, r);
        const TileCells tc = tile_cells(bc, tx, ty);
        cell_word = pack_cell_word(slot_base + __float_as_uint(gd.z) + (uint32_t)tc.first, tc, tx, ty);
    }
    // (fourth word: the entry's 1-based position in its tile's list, as a float -- what the blend kernels record as the last contributor
    // and compare against it; exact below 2^24.  With it in the record a blender reads nothing but the record ring per entry.)
    // (second word: {B, C, A, opacity} -- the order in which the blend trips multiply the conic by (dx, dy) as register pairs: blend.hip, pair_power)
    rec[0] = make_float4(ga.x, ga.y, __uint_as_float(cell_word), (float)(rank + 1u)); rec[1] = make_float4(gb.y, gb.z, gb.x, gb.w); rec[2] = gc;
    // (non-temporal or write-through stores here: +3 / +7 us -- the write-back at the kernel's end is cheaper)
    {
        // which 4x4 pixel blocks of this tile the entry's alpha >= 1/255 bounding box {x, y, hx, hy} touches (pixel centres are
        // integers; block b spans [bx0, bx0+3] x [by0, by0+3]); NaN extents give no bit, infinite ones every bit.  The blend
        // kernels scan these 2-byte masks instead of the 48-byte records.
        uint32_t xm = 0u, ym = 0u;
#pragma unroll
        for (int bq = 0; bq < 4; bq++) {
            const float bx0 = (float)(tx * TILE + 4 * bq), by0 = (float)(ty * TILE + 4 * bq);
            if (ga.x + ga.z >= bx0 && ga.x - ga.z <= bx0 + 3.0f) xm |= 1u << bq;
            if (ga.y + ga.w >= by0 && ga.y - ga.w <= by0 + 3.0f) ym |= 1u << bq;
        }
        uint32_t bmask = 0u;
#pragma unroll
        for (int by = 0; by < 4; by++) if (ym & (1u << by)) bmask |= xm << (4 * by);
        // Refinement: the box is loose for elongated, rotated Gaussians.  A pixel can only pass the alpha >= 1/255 test if
        // q(d) = 1/2 (A dx^2 + C dy^2) + B dx dy <= tau = ln(255 opacity) (forward.cu:336-350), i.e. inside the ellipse
        // A dx^2 + 2 B dx dy + C dy^2 <= K.  K = 2 (tau + 1e-3) / 0.9999 carries the margin for fp32 rounding (1e-4 relative + 1e-3
        // absolute, against |q| <= 5.6 at the threshold).  A block [x0, x0+3] x [y0, y0+3] (pixel centres) meets the ellipse exactly
        // if [x0, x0+3] overlaps the x-extent of (ellipse intersected with the strip y in [y0, y0+3]) -- that intersection is
        // convex, so its x-projection is an interval [xl, xr], attained where the strip comes closest to the ellipse's leftmost /
        // rightmost point (-+X_R, -+Y_R): one pair of square roots per block ROW instead of a constrained minimum per block (this
        // kernel spent most of its 580 wave-instructions per 64 instances on the latter).  Only done for finite extents and a
        // positive-definite conic (otherwise the box decision stands).  Every wasted (entry, block) pair costs a quarter of a blend
        // trip, forward and backward.
        const float A = gb.x, B = gb.y, C = gb.z, opa = gb.w;
        const float det = A * C - B * B;
        if (bmask != 0u && ga.z < 3.0e38f && ga.w < 3.0e38f && A > 0.0f && C > 0.0f && det > 0.0f && opa > 0.0f) {
            // (v_log / v_sqrt / v_rcp, ~1 ulp each: this file is compiled with correctly rounded division and square root, ~15
            // instructions apiece, and the eleven roots and four quotients below were 40 % of this kernel's instructions.  The
            // margins -- 1e-3 on tau, `slack` on every interval end -- are four orders of magnitude above those errors.)
            const float K = 2.0f * (__logf(255.0f * opa) + 1.0e-3f) * (1.0f / 0.9999f);
            uint32_t keep = 0u;
            if (K >= 0.0f) {
                const float inv_det = __builtin_amdgcn_rcpf(det), invA = __builtin_amdgcn_rcpf(A);
                const float aK = A * K;
                const float Ymax = __builtin_amdgcn_sqrtf(aK * inv_det);                   // |dy| on the ellipse
                const float X_R = __builtin_amdgcn_sqrtf(C * K * inv_det), Y_R = -B * X_R * __builtin_amdgcn_rcpf(C);     // its rightmost point
                const float slack = 1.0e-3f + 1.0e-5f * (fabsf(ga.x) + X_R);      // rounding of the roots / divisions, in pixels
                const float bx_first = (float)(tx * TILE) - ga.x, by_first = (float)(ty * TILE) - ga.y;
#pragma unroll
                for (int by = 0; by < 4; by++) {
                    const float y0 = by_first + (float)(4 * by), y1 = y0 + 3.0f;
                    if (y0 > Ymax + slack || y1 < -Ymax - slack) continue;            // the strip misses the ellipse
                    const float yr = fminf(fmaxf(Y_R, y0), y1), yl = fminf(fmaxf(-Y_R, y0), y1);
                    const float xr = (-B * yr + __builtin_amdgcn_sqrtf(fmaxf(aK - det * yr * yr, 0.0f))) * invA + slack;
                    const float xl = (-B * yl - __builtin_amdgcn_sqrtf(fmaxf(aK - det * yl * yl, 0.0f))) * invA - slack;
#pragma unroll
                    for (int bx = 0; bx < 4; bx++) {
                        const float x0 = bx_first + (float)(4 * bx);
                        if (x0 + 3.0f >= xl && x0 <= xr) keep |= 1u << (4 * by + bx);
                    }
                }
            }
            bmask &= keep;
        }
        inst_bmask[pos] = (uint16_t)bmask;
    }
    }   // mine
    KSTAMP(4); RSTAMP(7);
#undef KSTAMP
#undef RSTAMP
    if (last_turn) return;
    __syncthreads();                                          // s_own / s_keys are rewritten by the next round
    }
}

__global__ void __launch_bounds__(256)
export_binning_kernel(int T, GeomView g, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      uint64_t* __restrict__ keys_out, uint32_t* __restrict__ list_out, uint32_t* __restrict__ ranges_out)
{
    const int tile = blockIdx.x;
    if (tile >= T) return;
    const uint2 rg = ranges[tile];
    if (ranges_out && threadIdx.x == 0) {
        // the reference memsets ranges to 0 and only touches tiles that own instances (rasterizer_impl.cu:312-320)
        ranges_out[2 * tile] = (rg.y > rg.x) ? rg.x : 0u;
        ranges_out[2 * tile + 1] = (rg.y > rg.x) ? rg.y : 0u;
    }
    for (uint32_t i = rg.x + threadIdx.x; i < rg.y; i += blockDim.x) {
        const uint32_t id = point_list[i];
        if (list_out) list_out[i] = id;
        if (keys_out) keys_out[i] = ((uint64_t)(uint32_t)tile << 32) | (uint64_t)__float_as_uint(g.geo[4 * (size_t)id + 2].w);
    }
}

}  // anonymous namespace

__global__ void __launch_bounds__(256) clear_words_kernel(uint4* __restrict__ p, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// Zero `bytes` (a multiple of 16, 16-byte aligned) with a kernel.  Used instead of hipMemsetAsync for the per-frame state: a memset
// node inside a captured hipGraph did not re-execute on replay here (ROCm 7.2), which left stale histograms behind.
void launch_clear(void* ptr, size_t bytes, hipStream_t s)
{
    const size_t n16 = bytes / 16;
    if (n16 == 0) return;
    const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 256);
    hipLaunchKernelGGL(clear_words_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(ptr), n16);
}

__global__ void __launch_bounds__(256) zero_floats_kernel(float* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0f;
}

// The per-frame words of a caller's frame state: the flag word (its 16 bytes) and the tile counters; the sticky dropped-frame count stays.
void clear_frame_state(char* frame_state, size_t bytes, hipStream_t s)
{
    launch_clear(frame_state, 16, s);
    launch_clear(frame_state + FS_COUNTERS_OFFSET, bytes - FS_COUNTERS_OFFSET, s);
}

// Zero n floats of any alignment with a kernel (capture-safe, see launch_clear).
void launch_zero_floats(float* ptr, size_t n, hipStream_t s)
{
    if (n == 0) return;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(zero_floats_kernel, dim3(blocks), dim3(256), 0, s, ptr, n);
}

static int light_log2_knob()
{
    static const int v = std::max(0, std::min(31, knob("MOSS_LIGHT_LOG2", LIGHT_TILE_LOG2)));
    return v;
}

void launch_scan(int P, GeomView g, ImageView im, int num_tiles, long long capacity, hipStream_t s, bool forward_only)
{
    // (the scan kernel of its own: the synchronous path, whose pool is sized AFTER the host has read the frame's cell count -- no bound
    // here -- and asynchronous frames of more tiles than the LDS histogram holds, whose pool is the default one of the capacity)
    const uint32_t cap = capacity < 0 ? 0xffffffffu : (uint32_t)capacity;
    const uint32_t pool_cap = capacity < 0 ? 0xffffffffu : (uint32_t)std::min<size_t>(BinView::default_pool_cells((int)capacity), 0xfffffff0u);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, num_tiles, im.tile_count, im.ranges, im.chunk_base, im.tile_order,
                       im.header, cap, light_log2_knob(), (P + 255) / 256, im.flags_acc, im.queues, im.work_table,
                       g.group_rtot, g.group_rbase, pool_cap, forward_only ? 1 : 0);
}

// Whether the asynchronous forward can do without a scan (and a scatter) kernel of its own: the tile histogram must fit the LDS of the
// preprocess kernel, which then writes the keys into buckets, and of the sort workgroups, which scan it themselves.
bool forward_buckets_keys(const FrameParams& fp)
{
    static const int on = knob("MOSS_BUCKET_KEYS", 1);
    return on && fp.gx * fp.gy <= MAX_LDS_TILES;
}

// Bucketed keys (asynchronous path): the key area of the binning buffer -- the gradient-record pool, dead until the backward -- cut
// into one bucket per tile.  Proportional to the capacity the buffer was sized for: 36 keys per instance of capacity / tiles, i.e. ~70x
// the average list at the usual 2x margin (the bench frame's longest list: 20x its average).
uint32_t bucket_key_stride(const BinView& b, int num_tiles)
{
    const unsigned long long area = (unsigned long long)b.pool_cells * GRAD_REC_FLOATS * 4ull / sizeof(uint64_t);    // keys the area holds
    unsigned long long stride = num_tiles > 0 ? area / (unsigned long long)num_tiles : 0ull;
    stride = std::min<unsigned long long>(stride, 0x7fffffffull / (unsigned long long)std::max(num_tiles, 1));    // 32-bit key indices
    return (uint32_t)(stride & ~1ull);                       // (buckets start 16-byte aligned)
}

// Whether the scan can ride along with the scatter (no launch of its own): the tile histogram must fit the scatter's LDS and nobody may
// have to read R back before the binning buffer exists (an asynchronous forward).
bool scatter_folds_scan(const FrameParams& fp)
{
    static const int on = knob("MOSS_FOLD_SCAN", 1);
    return on && fp.gx * fp.gy <= MAX_LDS_TILES;
}

void launch_scatter(const FrameParams& fp, GeomView g, ImageView im, BinView b, hipStream_t s, bool fold_scan, long long capacity)
{
    const int T = fp.gx * fp.gy;
    // (2: timing experiment of MOSS_DIAG builds, no reservation atomics -- the keys are garbage and the forward stops behind this kernel)
    const int lds_hist = (T <= MAX_LDS_TILES) ? (((knob("MOSS_EXPERIMENT", 0) & 2) != 0) ? 2 : 1) : 0;
    static const int per_thread = knob("MOSS_SCATTER_ITEMS", 2);
    int blocks = (fp.P + 256 * per_thread - 1) / (256 * per_thread);
    if (blocks < 1) blocks = 1;
    const size_t lds = lds_hist ? 2 * (size_t)T * sizeof(uint32_t) : 0;
    const int fold = (fold_scan && lds_hist && capacity >= 0) ? 1 : 0;
    const uint32_t cap = capacity < 0 ? 0xffffffffu : (uint32_t)capacity;
    const uint32_t pool_cap = capacity < 0 ? 0xffffffffu : (uint32_t)std::min<size_t>(BinView::default_pool_cells((int)capacity), 0xfffffff0u);
    hipLaunchKernelGGL(scatter_kernel, dim3(blocks + fold), dim3(256), lds, s, fp.P, fp.gx, T, g, im.ranges, im.tile_cursor, b.keys,
                       lds_hist, im.header, (g_stamps && knob("MOSS_SORT_STAMPS", 0)) ? g_stamps + 131072 + 16384 : nullptr,
                       fold, im.tile_count, im.chunk_base, im.tile_order, cap, light_log2_knob(), im.flags_acc, im.queues, im.work_table,
                       pool_cap, fp.forward_only);
}

void launch_tile_sort(const FrameParams& fp, GeomView g, ImageView im, BinView b, int R, int total_chunks, hipStream_t s,
                      char* frame_state, size_t frame_state_bytes, int part, uint32_t key_stride, long long capacity)
{
    const int T = fp.gx * fp.gy;
    // R / total_chunks are exact in synchronous mode and upper bounds (capacity) in asynchronous mode; the kernels bound
    // themselves with the device-side values in the header
    if (R <= 0 || total_chunks <= 0) {
        if (frame_state && part == 0) clear_frame_state(frame_state, frame_state_bytes, s);
        return;
    }
    // at most two 1024-thread workgroups per CU in flight; the workgroups loop over the chunks
    static const int max_grid = std::max(8, knob("MOSS_SORT_GRID", 512));
    const int grid = std::min(total_chunks, max_grid);
    // diagnostics (scripts/sort_stamps.py): the stamp buffer's words [131072, 131072 + 16384) -- behind the forward blend's item stamps
    static const int stamps_on = knob("MOSS_SORT_STAMPS", 0);
    unsigned long long* const sort_stamps = (stamps_on && g_stamps) ? g_stamps + 131072 : nullptr;
    // (the first line of the frame state belongs to the scan block; the kernel that re-zeroes the rest: the sort on the synchronous
    // path, the merge when the keys are bucketed -- the sort workgroups then still READ the tile counts)
    uint4* const fs = reinterpret_cast<uint4*>(frame_state ? frame_state + FS_COUNTERS_OFFSET : nullptr);
    const uint32_t fs_n16 = (uint32_t)(frame_state ? (frame_state_bytes - FS_COUNTERS_OFFSET) / 16 : 0);
    const uint32_t cap = capacity < 0 ? 0xffffffffu : (uint32_t)capacity;
#define SORT_ARGS T, im.ranges, im.chunk_base, b.keys, im.header, sort_stamps, key_stride ? nullptr : fs, key_stride ? 0u : fs_n16, key_stride,     \
                  im.tile_count, cap, light_log2_knob(), im.tile_order, (fp.P + 255) / 256, im.flags_acc, im.queues, im.work_table, \
                  g.group_rtot, g.group_rbase, (uint32_t)std::min<size_t>(b.pool_cells, 0xfffffff0u)
    if (part == 0) {
        // (self-scan: one block more, the scan block -- inside the bound on resident workgroups, so that it never waits for a slot)
        const int sort_grid = key_stride ? std::max(1, std::min(grid, max_grid - 1)) + 1 : grid;
        if (T <= SORT_THREADS) MOSS_LAUNCH_TIMED(chunk_sort_kernel<1>, dim3(sort_grid), dim3(SORT_THREADS) /* KPT keys per thread */, 0, s, SORT_ARGS);
        else MOSS_LAUNCH_TIMED((chunk_sort_kernel<MAX_LDS_TILES / SORT_THREADS>), dim3(sort_grid), dim3(SORT_THREADS), 0, s, SORT_ARGS);
    } else
    MOSS_LAUNCH_TIMED(merge_gather_kernel, dim3((grid + 7) / 8 * 8 * MERGE_PARTS), dim3(MERGE_THREADS), 0, s, im.header, fp.gx, T, g, im.ranges, im.chunk_base, b.keys,
                       b.point_list, b.inst_rec, b.cell_valid, b.inst_bmask, sort_stamps ? sort_stamps + 8 * 1024 : nullptr,
                       key_stride, key_stride ? fs : nullptr, key_stride ? fs_n16 : 0u, fp.forward_only);
#undef SORT_ARGS
}

void launch_export_binning(const FrameParams& fp, GeomView g, ImageView im, BinView b, int R,
                           uint64_t* keys, uint32_t* point_list, uint32_t* ranges, float* final_T, uint32_t* n_contrib, hipStream_t s)
{
    (void)R;
    const int T = fp.gx * fp.gy;
    hipLaunchKernelGGL(export_binning_kernel, dim3(T), dim3(256), 0, s, T, g, im.ranges, b.point_list, keys, point_list, ranges);
    const size_t N = (size_t)fp.W * fp.H;
    if (final_T) (void)hipMemcpyAsync(final_T, im.final_T, N * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (n_contrib) (void)hipMemcpyAsync(n_contrib, im.n_contrib, N * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
}

}  // namespace moss
