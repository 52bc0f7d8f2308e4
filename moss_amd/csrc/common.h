// common.h -- internal declarations shared by the HIP translation units of libmoss_raster.so (gfx950 only).
//
// Data layout in HBM (all sub-arrays 256-byte aligned inside the three caller-owned scratch buffers; the
// reference's layout is rasterizer_impl.cu:155-194, ours differs on purpose -- see DESIGN.md "Data layout"):
//
//   geometry buffer (per Gaussian, P entries)
//     geo    4 x float4 = ONE 64-byte record per Gaussian (a per-instance gather in the sort touches one cache line, not five):
//       [0] {pix.x, pix.y, cull_hx, cull_hy}      position + half-extents of the alpha >= 1/255 box
//       [1] {conic.A, conic.B, conic.C, opacity}  } [0..2] are copied, in sorted order, into the per-instance stream the
//       [2] {r, g, b, depth}                      } blend kernels read (inst_rec; there [0] = {pix.x, pix.y, slot, list position + 1}
//                                                   and [1] = {B, C, A, opacity}: blend.hip, pair_power)
//       [3] bits {rect.min (x | y<<16), rect.max, rec_offsets, rec_count}   tile rectangle (getRect result) + the Gaussian's run of
//                                                   gradient-record cells (box_cells below): start relative to its group of 256, length
//     tiles_touched u32, rec_offsets u32, rec_count u32 (plain arrays: the backward reads them coalesced), per group of 256 Gaussians
//     group_rtot / group_rbase u32 (the group's cells; where its runs start in the record pool), radius i32,
//     clamped u8 (bit c = channel c),
//     cov3D float[6] (only written when computed from scale/rotation)
//   image buffer
//     header u32[16]: [0]=R (num_rendered) [1]=longest tile list [2]=error flags [3]=capacity that holds the frame (>= [6]: instances, record-pool cells / 6, key buckets) [4]=sort chunks [5]=non-empty tiles [6]=instances of the frame [9]=record-pool cells in use [16..31]=work-queue heads of the forward/backward blend
//     tile_count u32[T], tile_cursor u32[T], ranges uint2[T], final_T f32[N], n_contrib u32[N]
//   binning buffer (per (Gaussian,tile) instance, R entries)
//     point_list u32[R]   Gaussian ids, tile-major, each tile's run sorted by (depth bits, id)
//     inst_rec   48 B * R  the three records of every instance in sorted order (written by the tile sort)
//     cell_valid 1 bit per record-pool cell: the backward blend left a record there
//     pool       48 B * cells (~6 R; exact in the synchronous forward)   forward: 64-bit sort keys (depth<<32|id);
//                          backward: partial gradients, 3 float4 per (Gaussian, 4x4 pixel block under its bounding box) -- see box_cells
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include "moss_raster.h"

namespace moss {

constexpr int TILE = 16;                 // BLOCK_X = BLOCK_Y = 16, DGR/cuda_rasterizer/config.h:15-16
constexpr int TILE_PIX = TILE * TILE;
constexpr size_t BUF_ALIGN = 256;
constexpr int MAX_LDS_TILES = 8192;      // tile histograms are privatised in LDS up to this many tiles
constexpr uint32_t ERRFLAG_PREFILTERED = 1u;
constexpr uint32_t ERRFLAG_OVERFLOW = 2u;      // asynchronous forward: the frame needs more instances than the caller's capacity
constexpr uint32_t ERRFLAG_FORWARD_ONLY = 4u;  // MOSS_FORWARD_ONLY: the scratch holds no backward state (the backward kernels leave at once, zero gradients)
constexpr int FS_DROPPED_WORD = 4;             // frame state: sticky count of overflowed frames (include/moss_raster.h MOSS_FRAME_STATE_DROPPED_WORD)
constexpr size_t FS_COUNTERS_OFFSET = 256;     // frame state: where the per-frame tile counters start

inline size_t align_up(size_t v, size_t a = BUF_ALIGN) { return (v + a - 1) / a * a; }

template <typename T>
inline T* carve(char*& p, size_t count)
{
    T* r = reinterpret_cast<T*>(p);
    p += align_up(count * sizeof(T));
    return r;
}

struct GeomView {
    float4* geo;             // 4 float4 per Gaussian, see the layout comment at the top of this file
    uint32_t* tiles_touched;
    uint32_t* rec_offsets; uint32_t* rec_count;          // the Gaussian's run of gradient-record CELLS (box_cells below): start relative to its group, length
    uint32_t* group_rtot; uint32_t* group_rbase;         // per group: cells, and where its cell runs start in the record pool
    int* radius;
    uint8_t* clamped;
    float* cov3D;
    static GeomView at(char* base, int P)
    {
        GeomView g; char* p = base; size_t n = (size_t)P;
        g.geo = carve<float4>(p, 4 * n);
        g.tiles_touched = carve<uint32_t>(p, n);
        g.rec_offsets = carve<uint32_t>(p, n); g.rec_count = carve<uint32_t>(p, n);
        g.group_rtot = carve<uint32_t>(p, (n + 255) / 256); g.group_rbase = carve<uint32_t>(p, (n + 255) / 256);
        g.radius = carve<int>(p, n);
        g.clamped = carve<uint8_t>(p, n);
        g.cov3D = carve<float>(p, 6 * n);
        return g;
    }
    static size_t bytes(int P) { char* z = nullptr; GeomView g = at(z, P); return (size_t)((char*)g.cov3D - z) + align_up(6 * (size_t)P * 4); }
};

// header words: [0] R, [1] longest tile list, [2] error flags, [3] the capacity that holds the frame (>= [6]), [4] sort chunks, [5] tiles that own
// instances, [6] instances needed, [7] heavy tiles (list length >= 2^LIGHT_TILE_LOG2; they come first in tile_order), [8]/[9] unused, [10] backward leaver count,
// [16..23] / [24..31] per-XCD queue heads of the wave blend kernels (forward / backward)
constexpr int HEADER_WORDS = 32;
constexpr int LIGHT_TILE_LOG2 = 5;     // tiles with fewer than 2^5 entries are "light": blended one pixel per lane ([7] = heavy tiles)
constexpr int NUM_XCD_QUEUES = 8;
// One gradient record per (instance, 4x4 block): 9 values, padded to GRAD_REC_FLOATS.  16 (one 64-byte cache line per record: the
// per-Gaussian gather touches ONE line per record) or 12 (48 bytes: three of four records straddle two lines; measured: same 35 us).
constexpr int GRAD_REC_FLOATS = 12;
// Depth segments of the backward blend (blend.hip): while it blends a heavy 4x4 block front to back, the forward kernel cuts the
// block's hit list every SEG_HITS hits and leaves, per cut, a descriptor {tile, block, first position, end position} plus the pixel
// state AT THE END of the piece (T and the five running sums, 16 pixels x 6 floats).  The backward kernel then processes every
// piece as a work item of its own -- a long list is walked by several waves at once instead of back to front by one.
// Every queue word lives in a cache line of its own (QLINE_WORDS apart): atomics on words of ONE line are executed one after the
// other for the whole device, ~11.4 ns each whatever their addresses inside the line (scripts/micro/atomic_queue.hip: eight queue
// heads in eight consecutive words behaved like a single counter -- the last of 1024 waves got its first work item 12 us after
// the kernel started).  Lines: Q_FWD + xcd / Q_BWD + xcd = work-queue heads of the forward / backward blend, Q_SEG_ALLOC + xcd =
// segment slots handed out by the forward, Q_SEG_HEAD + xcd = segment pop heads of the backward.
constexpr int QLINE_WORDS = 64;
constexpr int Q_FWD = 0, Q_BWD = 8, Q_SEG_ALLOC = 16, Q_SEG_HEAD = 24, Q_LINES = 32;
constexpr int FWD_PAIRS_PER_WG = 2;         // the forward blend works in wave PAIRS (scanner + blender, blend.hip): two per 256-thread workgroup
constexpr int MAX_FWD_QUEUE_WAVES = 256;    // forward pairs (= blender waves, the ones that cut) per XCD region the segment bookkeeping supports; more = no cuts
constexpr int SEG_STATE_FLOATS = 16 * 6;      // per segment and pixel of the block: T at the segment's far end + the five sums
                                              // (r, g, b, depth, weight) of everything the pixel blends BEHIND it
inline size_t seg_region_cap(int R) { return (size_t)(R > 0 ? R : 0) / 128 + 4096; }   // slots per XCD region (shared equally by its forward waves)
struct ImageView {
    uint32_t* header;
    uint32_t* queues;        // Q_LINES cache lines of QLINE_WORDS words, one counter each (cleared with the header)
    uint32_t* tile_count; uint32_t* tile_cursor; uint2* ranges; uint32_t* chunk_base; uint32_t* tile_order;
    float* final_T; uint32_t* n_contrib;
    uint32_t* flags_acc;     // where the preprocess kernel ORs its error flags: &header[2], or the caller's frame state (below)
    uint32_t* tail_start;    // per (tile, 4x4 block): list position where the part of the block's list NOT covered by depth segments begins
    uint32_t* seg_counts;    // [NUM_XCD_QUEUES][MAX_FWD_QUEUE_WAVES]: segments each forward wave left in its slot range
    uint4* work_table;       // per RANK of tile_order (tiles with work first, longest class first): {tile, list start, list end, -} -- what a
                             // blend wave needs to start an item, in ONE load (rank -> tile -> range were two dependent round trips)
    static ImageView at(char* base, int W, int H)
    {
        ImageView v; char* p = base;
        size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE), N = (size_t)W * H;
        v.header = carve<uint32_t>(p, HEADER_WORDS);
        v.flags_acc = v.header + 2;
        v.queues = carve<uint32_t>(p, (size_t)Q_LINES * QLINE_WORDS);
        v.tile_count = carve<uint32_t>(p, T); v.tile_cursor = carve<uint32_t>(p, T);
        v.ranges = carve<uint2>(p, T); v.chunk_base = carve<uint32_t>(p, T); v.tile_order = carve<uint32_t>(p, T);
        v.final_T = carve<float>(p, N); v.n_contrib = carve<uint32_t>(p, N);
        v.tail_start = carve<uint32_t>(p, 16 * T);
        v.seg_counts = carve<uint32_t>(p, (size_t)NUM_XCD_QUEUES * MAX_FWD_QUEUE_WAVES);
        v.work_table = carve<uint4>(p, T);
        return v;
    }
    static size_t bytes(int W, int H)
    {
        char* z = nullptr; ImageView v = at(z, W, H);
        const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
        return (size_t)((char*)v.work_table - z) + align_up(T * sizeof(uint4));
    }
    // header + queues + tile_count + tile_cursor are contiguous (each carved at 16-byte granularity or coarser): one clear covers them
    size_t clear_bytes() const { return (size_t)((char*)ranges - (char*)header); }
    // FRAME STATE (moss_raster_frame_state): a caller-owned block that is all-zero between forward calls.  With it the per-frame
    // counters that kernels ADD to -- the tile histogram, the tile cursors, the error-flag word -- live there instead of in this
    // buffer, and nothing has to be zeroed before the preprocess kernel: the scan block writes every header word and zeroes the
    // queue words, the sort kernel re-zeroes the frame state for the next forward.  (The clear was a 4 us launch of 20 waves.)
    // Layout: word 0 = error flags of the frame in flight (zeroed by the scan block), word FS_DROPPED_WORD = STICKY count of frames
    // that overflowed their capacity and rendered nothing (only ever incremented by the library: the caller reads and resets it --
    // one look every few hundred replays of a captured step sees every dropped frame, not just the last one), byte 256 on = the
    // tile histogram and cursors (re-zeroed by the sort kernel).
    static size_t frame_state_bytes(int W, int H)
    {
        const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
        return 256 + align_up(2 * T * sizeof(uint32_t));
    }
    void use_frame_state(char* fs, int W, int H)
    {
        const size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
        flags_acc = reinterpret_cast<uint32_t*>(fs);
        tile_count = reinterpret_cast<uint32_t*>(fs + 256); tile_cursor = tile_count + T;
    }
};

#if defined(__HIPCC__)
// Visit every tile of every lane's tile rectangle (packed as in GeomView::rect).  Rectangles of up to COOP_TILES tiles are walked by
// their own lane.  Larger ones -- a Gaussian that grew to cover much of the image owns up to gx*gy tiles, and a single lane walking
// them is a millisecond-long tail -- are walked by the 64 lanes of the wave together, one rectangle at a time, the owner's
// `payload` broadcast to all of them.  Must be called with the whole wave converged.
constexpr int COOP_TILES = 32;
template <typename F>
__device__ __forceinline__ void wave_for_each_tile(uint2 rect, int gx, uint64_t payload, F&& f)
{
    const int x0 = (int)(rect.x & 0xffffu), y0 = (int)(rect.x >> 16), x1 = (int)(rect.y & 0xffffu), y1 = (int)(rect.y >> 16);
    const int w = x1 - x0, h = y1 - y0;
    const int n = (w > 0 && h > 0) ? w * h : 0;
    if (n <= COOP_TILES) {
        for (int ty = y0; ty < y1; ty++)
            for (int tx = x0; tx < x1; tx++) f(ty * gx + tx, payload);
    }
    unsigned long long big = __ballot(n > COOP_TILES);
    const int lane = (int)(threadIdx.x & 63u);
    while (big) {
        const int src = __ffsll(big) - 1;
        big &= big - 1;
        const int bx0 = __shfl(x0, src), by0 = __shfl(y0, src), bw = __shfl(w, src), bn = __shfl(n, src);
        const uint32_t plo = __shfl((uint32_t)payload, src), phi = __shfl((uint32_t)(payload >> 32), src);
        const uint64_t bp = ((uint64_t)phi << 32) | plo;
        for (int j = lane; j < bn; j += 64) {
            const int r = j / bw;
            f((by0 + r) * gx + bx0 + (j - r * bw), bp);
        }
    }
}
#endif

#if defined(__HIPCC__)
// THE GRADIENT-RECORD POOL (round 5; rounds 1-4: sixteen slabs of one 48-byte record per instance, 768 B per instance of address space).
// A record is the nine partial gradient sums one 4x4 pixel block of the image contributes to one Gaussian (blend.hip, backward).  A
// Gaussian can only leave records for the blocks its alpha >= 1/255 bounding box {x, y, hx, hy} (geo[0]) reaches inside its tile
// rectangle: its CELLS, a run of consecutive pool records per Gaussian (group-relative start: GeomView::rec_offsets; bases: the scan).
// Inside the run the cells are ordered by tile (the rectangle's tiles row-major: the Gaussian's instances), inside a tile row-major over
// the box's blocks in that tile -- every instance's cells are consecutive, and where they start follows in closed form from the box:
//   cells in front of tile (tx, ty) = rows_before * NBX + nby(ty) * cols_before,   nby(ty) = box rows inside tile row ty.
// box_cells() is THE definition, evaluated with the same bits by the preprocess kernel (counts), merge_gather (where an instance's
// cells start: it rides in the instance's record) and the per-Gaussian backward (how many cells to sum).
// The box must contain every block merge_gather's block mask can flag: there block column cb (pixels 4cb .. 4cb + 3) passes iff
// fl(x + hx) >= 4cb and fl(x - hx) <= 4cb + 3; here cb <= floor(fl(x + hx) / 4) (the same condition: the division by 4 is exact) and
// cb >= floor(fl(x - hx) / 4) (<= the exact bound ceil((fl(x - hx) - 3) / 4): at most one column more).  Infinite extents (NaN opacity)
// clip to the rectangle; negative ones (opacity <= 0) give an empty box.
struct BoxCells { int gx0, gy0, nbx, nby; };               // global block coordinates of the first cell; columns, rows (<= 0: empty)
__device__ __forceinline__ BoxCells box_cells(float x, float y, float hx, float hy, uint2 rect)
{
#pragma clang fp contract(off)
    const int x0 = (int)(rect.x & 0xffffu), y0 = (int)(rect.x >> 16), x1 = (int)(rect.y & 0xffffu), y1 = (int)(rect.y >> 16);
    BoxCells c;
    const float lx = fmaxf(floorf((x - hx) * 0.25f), (float)(4 * x0)), ux = fminf(floorf((x + hx) * 0.25f), (float)(4 * x1 - 1));
    const float ly = fmaxf(floorf((y - hy) * 0.25f), (float)(4 * y0)), uy = fminf(floorf((y + hy) * 0.25f), (float)(4 * y1 - 1));
    // (NaN-free: x, y finite for a Gaussian with a rectangle; hx, hy finite, +inf or -1)
    c.gx0 = (int)lx; c.gy0 = (int)ly;
    c.nbx = (x1 > x0 && y1 > y0 && ux >= lx && uy >= ly) ? (int)ux - c.gx0 + 1 : 0;
    c.nby = c.nbx > 0 ? (int)uy - c.gy0 + 1 : 0;
    return c;
}
// where the cells of the instance in tile (tx, ty) start inside the Gaussian's run, and that tile's share of the box
struct TileCells { int first; int gx0, gy0, nbx, nby; };   // (global block coordinates again; nbx / nby <= 0: the box misses the tile)
__device__ __forceinline__ TileCells tile_cells(const BoxCells& c, int tx, int ty)
{
    TileCells t;
    t.gx0 = max(c.gx0, 4 * tx); t.gy0 = max(c.gy0, 4 * ty);
    t.nbx = min(c.gx0 + c.nbx, 4 * tx + 4) - t.gx0; t.nby = min(c.gy0 + c.nby, 4 * ty + 4) - t.gy0;
    const int rows_before = max(0, min(c.gy0 + c.nby, 4 * ty) - c.gy0), cols_before = max(0, min(c.gx0 + c.nbx, 4 * tx) - c.gx0);
    t.first = rows_before * c.nbx + max(t.nby, 0) * cols_before;
    return t;
}
// The word an instance's record carries in place of the slot of rounds 2-4 (inst_rec[0].z): from it the backward blend finds the cell of
// block (bx, by) of the instance's tile -- cell = (w >> 2) - 16 + by * ((w & 3) + 1) + bx -- i.e. w = ((base - by0 * nbx - bx0 + 16) << 2) |
// (nbx - 1) with base = the pool index of the instance's first cell and (bx0, by0) that cell's block inside the tile.
__device__ __forceinline__ uint32_t pack_cell_word(uint32_t base, const TileCells& t, int tx, int ty)
{
    const int nbx = max(t.nbx, 1);
    return (uint32_t)(((int)base - (t.gy0 - 4 * ty) * nbx - (t.gx0 - 4 * tx) + 16) << 2) | (uint32_t)(nbx - 1);
}
constexpr uint32_t POOL_MAX_CELLS = 80u * 1000u * 1000u;    // the kernels address the pool with 32-bit byte offsets: 48 B x 80M < 4 GB (and < the word's 30 bits)
#endif

// Per-kernel timing (moss_raster_profile_*): while a single-kernel stage is being timed, its launcher dispatches through
// hipExtLaunchKernelGGL with the stage's two events attached to the KERNEL (begin / end of execution, what rocprofv3 reports),
// instead of hipEventRecord calls around the launch (which also see the launch latency in front of the kernel: +5-10 us eager).
struct StageEvents { hipEvent_t start = nullptr, stop = nullptr; bool used = false; };
extern thread_local StageEvents g_stage_events;          // raster_api.hip
#define MOSS_LAUNCH_TIMED(kernel, grid, block, lds, stream, ...)                                                          \
    do {                                                                                                                  \
        if (moss::g_stage_events.start != nullptr && !moss::g_stage_events.used) {                                        \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, moss::g_stage_events.start, moss::g_stage_events.stop, 0, __VA_ARGS__); \
            moss::g_stage_events.used = true;                                                                             \
        } else {                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                            \
        }                                                                                                                 \
    } while (0)

// ---- tuning knobs and diagnostics.  The PRODUCT build has neither: knob(name, dflt) is the constant dflt, the stamp buffers are
// null constants, and no translation unit under csrc/ reads the environment.  A build with -DMOSS_DIAG (python -m moss_amd.build --diag ->
// moss_amd/lib_diag/, used by scripts/ only) reads MOSS_* environment variables through knob() and exports the
// moss_raster_debug_set_*stamps entry points; some of its knobs produce WRONG results on purpose (timing experiments).
#ifdef MOSS_DIAG
int knob(const char* name, int dflt);                  // scripts/diag/knobs.cpp: the integer value of an environment variable, else dflt
extern unsigned long long* g_stamps;                   // optional forward-blend phase stamps, blend.hip
extern unsigned long long* g_bwd_stamps;               // optional per-wave stamps of the backward blend kernel, blend.hip
#else
constexpr int knob(const char*, int dflt) { return dflt; }
constexpr unsigned long long* g_stamps = nullptr;
constexpr unsigned long long* g_bwd_stamps = nullptr;
#endif

// Record pool capacity of a binning buffer sized for R instances when nothing better is known (the asynchronous forward, whose R is the
// caller's capacity -- about twice the frame's -- and moss_raster_binning_bytes): POOL_CELLS_PER_INSTANCE cells per instance.  Measured
// need, cells per instance of the FRAME: 3.7 (bench frame, configs[2]), 4.5 (configs[4]), 5.0 (configs[1]: Gaussians that span 4-12
// tiles), 5.5 (configs[0]); the most cells of one Gaussian there: 88 / 240 / 81 / 70.
// The synchronous forward reads the frame's exact cell count back with R and sizes the pool for it.
constexpr int POOL_CELLS_PER_INSTANCE = 6;
struct BinView {
    uint32_t* point_list;
    uint16_t* inst_bmask;    // per instance (sorted order): bit b set <=> its alpha >= 1/255 bounding box touches 4x4 block b of its tile
    uint64_t* keys;          // aliases the record pool (dead after the sort)
    float4* inst_rec;        // 3 float4 per instance, sorted order: what the blend kernels stage (contiguous per tile)
    uint4* seg_desc;         // NUM_XCD_QUEUES regions of seg_cap descriptors {tile, block, first position, end position}
    float* seg_state;        // SEG_STATE_FLOATS per descriptor
    uint32_t seg_cap;        // slots per region
    uint32_t* cell_valid;    // one bit per pool cell: the backward blend left a record there (zeroed by merge_gather); sized for 16 cells
                             // per instance -- the most a frame can need -- so that its place does not depend on the pool's size
    float4* inst_grad;       // THE POOL: 3 float4 per cell (see box_cells); the LAST array of the buffer: only its size varies
    size_t pool_cells;       // its capacity
    // forward_only (MOSS_FORWARD_ONLY: an evaluation render, render_ZJU.py:56-72 -- no backward will follow): what the forward itself
    // reads -- ids, block masks, the 48-byte records -- and the 8-byte sort keys; no depth-segment slots, no validity bits, no record
    // pool: 62 B per instance where the training layout takes ~370.
    static BinView at(char* base, int R, long long pool_cells = -1, bool forward_only = false)
    {
        BinView b; char* p = base; size_t n = (size_t)(R > 0 ? R : 1);
        b.point_list = carve<uint32_t>(p, n);
        b.inst_bmask = carve<uint16_t>(p, n);
        b.inst_rec = carve<float4>(p, 3 * n);
        if (forward_only) {
            b.seg_cap = (uint32_t)seg_region_cap(R);         // (the NUMBER the training layout would have: the forward blend cuts its sums where
                                                             // the training forward does -- same image bits -- without keeping the pieces)
            b.seg_desc = nullptr; b.seg_state = nullptr; b.cell_valid = nullptr;
            b.inst_grad = reinterpret_cast<float4*>(p);
            b.keys = reinterpret_cast<uint64_t*>(p);
            b.pool_cells = 0;
            return b;
        }
        b.seg_cap = (uint32_t)seg_region_cap(R);
        b.seg_desc = carve<uint4>(p, (size_t)NUM_XCD_QUEUES * b.seg_cap);
        b.seg_state = carve<float>(p, (size_t)NUM_XCD_QUEUES * b.seg_cap * SEG_STATE_FLOATS);
        b.cell_valid = carve<uint32_t>(p, (16 * n + 31) / 32 + 2);
        b.inst_grad = reinterpret_cast<float4*>(p);
        b.keys = reinterpret_cast<uint64_t*>(b.inst_grad);
        b.pool_cells = pool_cells >= 0 ? (size_t)pool_cells : default_pool_cells(R);
        return b;
    }
    static size_t default_pool_cells(int R) { return (size_t)POOL_CELLS_PER_INSTANCE * (size_t)(R > 0 ? R : 1) + 4096; }
    // bytes of a buffer for R instances whose pool holds `pool_cells` cells (never less than the sort keys need: they alias it)
    static size_t bytes(int R, long long pool_cells = -1, bool forward_only = false)
    {
        char* z = nullptr; BinView b = at(z, R, pool_cells, forward_only);
        const size_t n = (size_t)(R > 0 ? R : 1);
        return (size_t)((char*)b.inst_grad - z) + align_up(std::max(b.pool_cells * GRAD_REC_FLOATS * 4, n * sizeof(uint64_t)));
    }
};

// Per-call constants.  The camera matrices stay on the device (the boundary hands over device pointers, exactly
// like the reference); kernels read them through wave-uniform (scalar) loads, so no host copy / sync is needed.
// `raw` bits of FrameParams (moss_raster_forward_raw / _backward_raw): the GaussianModel getters applied inside preprocess
constexpr int RAW_OPACITY = 1;      // opacities are logits:            get_opacity  = sigmoid(_opacity)          (scene/gaussian_model.py:160-161)
constexpr int RAW_SCALE = 2;        // scales are logarithms:           get_scaling  = exp(_scaling)              (:142-143)
constexpr int HINT_SPATIAL_ORDER = 8; // (not a raw-parameter bit) index neighbours are spatial neighbours: include/moss_raster.h
constexpr int RAW_POSE = 16;        // means3D are CANONICAL positions: posed inside the op, p = T x (+ translation)   (gaussian_renderer/__init__.py:74-77)
constexpr int RAW_ROTATION = 4;     // rotations are not normalised:    get_rotation = normalize(_rotation)       (:146-147)
constexpr int SH_GRAD_ACTIVE_ONLY = 32; // (not a raw-parameter bit) dL_dsh: only the coefficients of the active degree are written: include/moss_raster.h
struct FrameParams {
    int P, D, M, W, H, gx, gy;
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int prefiltered;
    int raw;                 // RAW_* bits: which of opacity / scales / rotations arrive as MOSS's raw parameters (activated inside the op)
    int forward_only;        // MOSS_FORWARD_ONLY: no backward state is produced (no depth-segment cuts, no gradient-record cells)
    int no_block_cull;       // MOSS_DEBUG_NO_BLOCK_CULL of the call's `debug` argument: the blend kernels ignore the per-instance block masks
    int exact_math;          // MOSS_DEBUG_EXACT_MATH: the blend kernels decide every pixel's list with the reference's source arithmetic (blend.hip)
    const float* view_dev; const float* proj_dev; const float* campos_dev; const float* bg_dev;
};

// ---- launchers (each defined in exactly one .hip file; all enqueue on `s`, none synchronises) -------------
void launch_preprocess_forward(const FrameParams& fp, const float* means3D, const float* shs, const float* colors_precomp,
                               const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                               const float* transforms, const float* translation, GeomView g, ImageView im, int* radii_out, hipStream_t s,
                               uint64_t* scatter_keys = nullptr /* scatter mode: the kernel also writes the sort keys into per-tile buckets */,
                               uint32_t key_stride = 0);
struct FusedAdam;
void launch_preprocess_backward(const FrameParams& fp, const float* means3D, const float* shs, const float* colors_precomp,
                                const float* opacities /* only read in raw mode */,
                                const float* scales, const float* rotations, const float* cov3D_precomp,
                                GeomView g, BinView b, const uint32_t* header, uint32_t* queues /* backward heads are rewound here */,
                                float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                                float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                                const float* transforms, float* dL_dtransforms, const float* translation, float* dL_dtranslation, hipStream_t s,
                                const struct FusedAdam* fused = nullptr /* adamw.h: the kernel also applies the AdamW update */);
void launch_mark_visible(int P, const float* means3D, const float* view16_dev, uint8_t* present, hipStream_t s);

void launch_clear(void* ptr, size_t bytes, hipStream_t s);
void clear_frame_state(char* frame_state, size_t bytes, hipStream_t s);   // its per-frame words (not the sticky dropped-frame count)
void launch_zero_floats(float* ptr, size_t n, hipStream_t s);
void launch_scan(int P, GeomView g, ImageView im, int num_tiles, long long capacity, hipStream_t s, bool forward_only = false);   // offsets, ranges, header, group bases
bool forward_buckets_keys(const FrameParams& fp);                                                // asynchronous forward without scan / scatter kernels, see binning.hip
bool scatter_folds_scan(const FrameParams& fp);                                                   // asynchronous, un-bucketed forward: the scan rides with the scatter
void launch_scatter(const FrameParams& fp, GeomView g, ImageView im, BinView b, hipStream_t s,    // duplicateWithKeys into exact ranges
                    bool fold_scan = false /* no scan kernel ran: every block scans the tile counts itself, one extra block writes the scan's outputs */,
                    long long capacity = -1);
void launch_tile_sort(const FrameParams& fp, GeomView g, ImageView im, BinView b, int R, int total_chunks, hipStream_t s,
                      char* frame_state, size_t frame_state_bytes, int part,   // part 0: chunk sort, part 1: merge + emit
                      uint32_t key_stride = 0 /* != 0: bucketed keys written by the preprocess kernel; the sort scans the tile counts itself */,
                      long long capacity = -1);   // (fp.forward_only: merge_gather emits no cell words and clears no validity bits)
uint32_t bucket_key_stride(const BinView& b, int num_tiles);  // slots per tile bucket the key area of this binning buffer holds (binning.hip)
void launch_export_binning(const FrameParams& fp, GeomView g, ImageView im, BinView b, int R,
                           uint64_t* keys, uint32_t* point_list, uint32_t* ranges, float* final_T, uint32_t* n_contrib, hipStream_t s);
void launch_export_geometry(int P, GeomView g, float* depths, float* means2D, float* conic_opacity, float* rgb,
                            uint32_t* tiles_touched, uint8_t* clamped, float* cov3D, hipStream_t s);

void launch_blend_forward(const FrameParams& fp, GeomView g, ImageView im, BinView b,
                          float* out_color, float* out_depth, float* out_alpha, hipStream_t s);
void launch_blend_backward(const FrameParams& fp, GeomView g, ImageView im, BinView b,
                           const float* dL_dpix, const float* dL_ddepth, const float* dL_dalpha, hipStream_t s);

// shared device helpers -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rect_area(uint2 r)
{
    uint32_t w = (r.y & 0xffffu) - (r.x & 0xffffu), h = (r.y >> 16) - (r.x >> 16);
    return w * h;
}

}  // namespace moss
