// optim.hip -- flat fused AdamW over the frame-parallel gradient bucket.
// MOSS steps torch.optim.AdamW over 8 parameter groups (scene/gaussian_model.py:215-226); per step that is one
// multi-tensor launch per group.  All Gaussian parameters, their gradients (moss_amd/dist.py GradBucket) and both
// moments live in four flat fp32 buffers here, so one streaming kernel (28 B per parameter) updates everything;
// groups differ only in learning rate, looked up from a small segment table.
// Semantics = torch.optim.AdamW (amsgrad=False, maximize=False): decoupled weight decay, bias-corrected moments.
#include "common.h"
#include "adamw.h"

namespace moss {
namespace {

// active[s] (with period[s] > 0): of every `period` elements of segment s only the first `active` have ever received a gradient -- the SH
// record (P,16,3) while MOSS trains below its maximum degree (train_ZJU.py:85-86: degree 0 / 1 / 2 for iterations 1-2999): the rest has
// zero gradients and zero moments, so its AdamW step is the weight decay alone and its moments stay zero: they are not touched.
// inactive_zero: those parameters are also known to be exactly zero (MOSS initialises features_rest with zeros,
// scene/gaussian_model.py:179-181; 0 x decay = 0): nothing of them is read or written at all.  Same bits as the full update either way.
struct Segs { int n; long long end[8]; float lr[8]; int period[8], split[8]; float lr2[8]; int active[8]; int inactive_zero; };
// SEVERAL gradient buffers of the same layout (moss_adamw_flat_args.grads_extra): B views rendered for one optimizer step on one device
// (moss_amd/multiview.py) -- the step's gradient is ((g0 + g1) + g2 ...) x scale, added in THAT order and rounded after every operation
// (what accumulating the views one after the other into one buffer gives), formed here instead of by B passes over the buffers
struct MoreGrads { int n; const float* g[3]; float scale; };

__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))   // (eight waves per SIMD: the scalar file admits six at 106 SGPRs)
adamw_kernel(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
             Segs segs, AdamBetas betas, float eps, float weight_decay, float bc1, float bc2_sqrt,
             const float* __restrict__ step_state, long long first, const uint32_t* __restrict__ skip_word, uint32_t skip_mask, MoreGrads more)
{
    // guard (moss_adamw_flat_guarded): a dropped frame's step is a no-op -- nothing is read or written, the step counter stays
    if (skip_word != nullptr && (*skip_word & skip_mask) != 0u) return;
    // `first`: the arrays are the elements [first, first + n) of the flat buffers the segment table indexes (a rank's shard of the
    // bucket, moss_adamw_flat_range); a multiple of 4, so that a thread's four elements never straddle it.
    // (the step count and the learning-rate flag are requested together: one scalar round trip at the kernel's start, not two)
    int t_prev = 0, lr_flag = 0;
    if (step_state) { t_prev = reinterpret_cast<const int*>(step_state)[0]; lr_flag = reinterpret_cast<const int*>(step_state)[ADAMW_LR_VALID_WORD]; }
    const bool lr_table = lr_flag != 0;
    int t_dev = 0;
    if (step_state)                                          // device-resident step counter (graph replay): adamw.h
        t_dev = adamw_step_begin(step_state, betas, blockIdx.x == 0 && threadIdx.x == 0, bc1, bc2_sqrt, t_prev);
    // (once per thread, correctly rounded; per ELEMENT: a hardware square root and reciprocal, ~1 ulp each -- with this file's correctly
    // rounded sqrt / divisions the update was ~40 vector instructions per element, ten million per step: as much issue time as the
    // kernel's 165 MB are HBM time)
    const float inv_bc1 = 1.0f / bc1, inv_bc2_sqrt = 1.0f / bc2_sqrt;
    // (buffer loads: an out-of-range offset returns 0 WITHOUT a memory request -- what an element that is not to be read costs; the
    // arrays of a launch are < 4 GB: launch_adamw falls back to treating everything as active otherwise)
    constexpr uint32_t OOB = 0xffffffffu, RSRC3 = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xffffff00u, RSRC3);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 0xffffff00u, RSRC3);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void*)m, 0, 0xffffff00u, RSRC3);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)v, 0, 0xffffff00u, RSRC3);
    typedef float v4f __attribute__((ext_vector_type(4)));
    for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 * 4 < n; i4 += (long long)gridDim.x * blockDim.x) {
        const long long i = i4 * 4, gi = first + i;          // index in the arrays / in the flat buffer (segment table)
        // Learning rate of each of the 4 elements: the segment of the first element (static indices only: a run-time index into the
        // kernel-argument tables would spill them to scratch), then its optional periodic pattern with ONE 32-bit modulo per
        // thread, stepped for the other elements.  A thread whose 4 elements straddle a segment end takes the general lookup.
        float lr4[4];
        bool dead4 = false;                                  // all four elements are INACTIVE ones (see Segs): decay only, or nothing
        {
            long long seg_start = 0, seg_end = first + n; float lr_a = 0.f, lr_b = 0.f; int period = 0, split = 0, seg_i = 0, active = 0;
#pragma unroll
            for (int s = 7; s >= 0; s--) if (s < segs.n && gi < segs.end[s]) {
                seg_end = segs.end[s]; seg_start = s > 0 ? segs.end[s - 1] : 0;
                lr_a = segs.lr[s]; lr_b = segs.lr2[s]; period = segs.period[s]; split = segs.split[s]; seg_i = s; active = segs.active[s];
            }
            if (lr_table) { lr_a = step_state[ADAMW_LR_WORD0 + seg_i]; lr_b = step_state[ADAMW_LR2_WORD0 + seg_i]; }   // (device-resident rates)
            if (gi + 3 < seg_end) {
                unsigned ph = period > 0 ? (unsigned)(gi - seg_start) % (unsigned)period : 0u;
                // (a tensor starts 16-byte aligned and the period is a multiple of 4 where `active` is used: the four elements lie in ONE
                // record; they are all inactive iff the first one is at or behind the active part rounded up to a float4)
                dead4 = period > 0 && active > 0 && (period & 3) == 0 && ph >= (unsigned)((active + 3) & ~3);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    lr4[k] = (period > 0 && (int)ph >= split) ? lr_b : lr_a;
                    if (++ph == (unsigned)period) ph = 0u;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const long long idx = gi + k;
                    float lr = 0.f;
#pragma unroll
                    for (int s = 7; s >= 0; s--) if (s < segs.n && idx < segs.end[s]) {
                        lr = lr_table ? step_state[ADAMW_LR_WORD0 + s] : segs.lr[s];
                        if (segs.period[s] > 0) {
                            const unsigned local = (unsigned)(idx - (s > 0 ? segs.end[s - 1] : 0));
                            if ((int)(local % (unsigned)segs.period[s]) >= segs.split[s]) lr = lr_table ? step_state[ADAMW_LR2_WORD0 + s] : segs.lr2[s];
                        }
                    }
                    lr4[k] = lr;
                }
            }
        }
        float pv[4], gv[4], mv[4], vv[4];
        const bool full = i + 4 <= n;
        if (full) {
            const uint32_t o = (uint32_t)i * 4u;
            const bool skip_all = dead4 && segs.inactive_zero != 0;
            const v4f a = __builtin_amdgcn_raw_buffer_load_b128(rs_p, skip_all ? OOB : o, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rs_g, dead4 ? OOB : o, 0, 0);
            const v4f c = __builtin_amdgcn_raw_buffer_load_b128(rs_m, dead4 ? OOB : o, 0, 0), d = __builtin_amdgcn_raw_buffer_load_b128(rs_v, dead4 ? OOB : o, 0, 0);
            pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
            if (more.n > 0) {                                // (kernel-uniform) the other views' gradients, in order, then the scale
                v4f e[3];
#pragma unroll
                for (int q = 0; q < 3; q++)
                    e[q] = q < more.n ? __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)more.g[q], 0, 0xffffff00u, RSRC3), dead4 ? OOB : o, 0, 0)
                                      : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 3; q++) if (q < more.n) {
                    gv[0] = __fadd_rn(gv[0], e[q].x); gv[1] = __fadd_rn(gv[1], e[q].y); gv[2] = __fadd_rn(gv[2], e[q].z); gv[3] = __fadd_rn(gv[3], e[q].w);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) gv[k] = __fmul_rn(gv[k], more.scale);
            }
            mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
        } else {
            for (int k = 0; k < 4; k++) {
                const bool ok = i + k < n; pv[k] = ok ? p[i + k] : 0.f; gv[k] = ok ? g[i + k] : 0.f; mv[k] = ok ? m[i + k] : 0.f; vv[k] = ok ? v[i + k] : 0.f;
                if (more.n > 0) {
                    for (int q = 0; q < more.n; q++) gv[k] = __fadd_rn(gv[k], ok ? more.g[q][i + k] : 0.f);
                    gv[k] = __fmul_rn(gv[k], more.scale);
                }
            }
            dead4 = false;
        }
        if (dead4) {
            // zero gradient, zero moments: adamw_element's p <- fma(-(lr / bc1), m' * rcp(eps), p') with m' = 0 is p' = p (1 - lr wd)
            // exactly (and the moments stay zero): the same bits, one multiply.  Known-zero parameters were not even read.
            if (segs.inactive_zero == 0) {
                float pn[4];
#pragma unroll
                for (int k = 0; k < 4; k++) pn[k] = __fmul_rn(pv[k], __fsub_rn(1.0f, __fmul_rn(lr4[k], weight_decay)));
                if (pn[0] != pv[0] || pn[1] != pv[1] || pn[2] != pv[2] || pn[3] != pv[3])      // (a zero stays a zero: nothing to write)
                    reinterpret_cast<float4*>(p)[i4] = make_float4(pn[0], pn[1], pn[2], pn[3]);
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            adamw_element(pv[k], gv[k], mv[k], vv[k], lr4[k], betas, eps, weight_decay, inv_bc1, inv_bc2_sqrt);
        if (full) {
            reinterpret_cast<float4*>(p)[i4] = make_float4(pv[0], pv[1], pv[2], pv[3]);
            reinterpret_cast<float4*>(m)[i4] = make_float4(mv[0], mv[1], mv[2], mv[3]);
            reinterpret_cast<float4*>(v)[i4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            for (int k = 0; k < 4; k++) if (i + k < n) { p[i + k] = pv[k]; m[i + k] = mv[k]; v[i + k] = vv[k]; }
        }
    }
    if (step_state && threadIdx.x == 0) adamw_step_end(step_state, t_dev);
}

int launch_adamw(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int num_segments,
                 const long long* segment_end, const float* segment_lr, const int* segment_period, const int* segment_split,
                 const float* segment_lr2, double beta1, double beta2, float eps, float weight_decay,
                 float bc1, float bc2_sqrt, const float* step_state, hipStream_t stream, long long first = 0,
                 const uint32_t* skip_word = nullptr, uint32_t skip_mask = 0u, const int* segment_active = nullptr, int inactive_zero = 0,
                 int num_extra = 0, const float* const* grads_extra = nullptr, float grad_scale = 1.0f)
{
    MoreGrads more; more.n = 0; more.scale = 1.0f; more.g[0] = more.g[1] = more.g[2] = nullptr;
    if (num_extra > 0 && grads_extra != nullptr) {
        more.n = num_extra > 3 ? 3 : num_extra; more.scale = grad_scale;
        for (int i = 0; i < more.n; i++) more.g[i] = grads_extra[i];
    }
    Segs segs; segs.n = num_segments;
    // (the degree-aware form addresses the arrays with 32-bit byte offsets: beyond 4 GB per array everything is treated as active)
    // (and eps = 0 would make the full update of an all-zero element 0 x rcp(0) = NaN: no shortcut then)
    const bool aware = segment_active != nullptr && n < (long long)(0xffffff00u / 4u) && eps > 0.0f;
    segs.inactive_zero = aware ? inactive_zero : 0;
    for (int i = 0; i < 8; i++) {
        segs.active[i] = (aware && i < num_segments && segment_period && segment_period[i] > 0 && segment_active[i] > 0 &&
                          segment_active[i] < segment_period[i]) ? segment_active[i] : 0;
        segs.end[i] = i < num_segments ? segment_end[i] : first + n; segs.lr[i] = i < num_segments ? segment_lr[i] : 0.f;
        const bool pat = i < num_segments && segment_period && segment_split && segment_lr2 && segment_period[i] > 0;
        segs.period[i] = pat ? segment_period[i] : 0; segs.split[i] = pat ? segment_split[i] : 0; segs.lr2[i] = pat ? segment_lr2[i] : 0.f;
    }
    long long blocks = (n / 4 + 255) / 256;
    static const long long max_blocks = knob("MOSS_ADAMW_BLOCKS", 2048);
    if (blocks > max_blocks) blocks = max_blocks;            // (2048: eight 256-thread blocks per CU, all resident at once)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, params, grads, exp_avg, exp_avg_sq,
                       segs, AdamBetas(beta1, beta2), eps, weight_decay, bc1, bc2_sqrt, step_state, first, skip_word, skip_mask, more);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}

// ---- SEVERAL parameter tensors with buffers of their own in ONE launch (moss_adamw_multi): the torch-state drop-in optimizer
// (moss_amd.optim.AdamW, MOSS's six single-tensor Gaussian groups -- scene/gaussian_model.py:215-226) took one launch per tensor: six
// launches and six host calls per step of a call pattern that is bound by the host.  Same arithmetic per element as adamw_kernel
// (adamw_element, the same bias corrections from the tensor's own step count): bit-identical to six moss_adamw_flat calls.
struct MultiT {
    int n;
    float* p[8]; const float* g[8]; float* m[8]; float* v[8];
    long long numel[8]; float lr[8], bc1[8], bc2_sqrt[8];
    unsigned block_end[8];                                   // tensor t owns the blocks [block_end[t-1], block_end[t])
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))
adamw_multi_kernel(MultiT T, AdamBetas betas, float eps, float weight_decay)
{
    // this block's tensor (static indices only: a run-time index into the kernel-argument tables would spill them to scratch)
    float* p = nullptr; const float* g = nullptr; float* m = nullptr; float* v = nullptr;
    long long n = 0; float lr = 0.f, bc1 = 1.f, bc2_sqrt = 1.f; unsigned b0 = 0u, b1 = 0u;
#pragma unroll
    for (int t = 7; t >= 0; t--) if (t < T.n && blockIdx.x < T.block_end[t]) {
        p = T.p[t]; g = T.g[t]; m = T.m[t]; v = T.v[t]; n = T.numel[t]; lr = T.lr[t]; bc1 = T.bc1[t]; bc2_sqrt = T.bc2_sqrt[t];
        b1 = T.block_end[t]; b0 = t > 0 ? T.block_end[t - 1] : 0u;
    }
    const float inv_bc1 = 1.0f / bc1, inv_bc2_sqrt = 1.0f / bc2_sqrt;
    const long long stride = (long long)(b1 - b0) * blockDim.x;
    for (long long i4 = (long long)(blockIdx.x - b0) * blockDim.x + threadIdx.x; i4 * 4 < n; i4 += stride) {
        const long long i = i4 * 4;
        float pv[4], gv[4], mv[4], vv[4];
        const bool full = i + 4 <= n;
        if (full) {
            const float4 a = reinterpret_cast<const float4*>(p)[i4], b = reinterpret_cast<const float4*>(g)[i4];
            const float4 c = reinterpret_cast<const float4*>(m)[i4], d = reinterpret_cast<const float4*>(v)[i4];
            pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
            mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
        } else {
            for (int k = 0; k < 4; k++) { const bool ok = i + k < n; pv[k] = ok ? p[i + k] : 0.f; gv[k] = ok ? g[i + k] : 0.f; mv[k] = ok ? m[i + k] : 0.f; vv[k] = ok ? v[i + k] : 0.f; }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) adamw_element(pv[k], gv[k], mv[k], vv[k], lr, betas, eps, weight_decay, inv_bc1, inv_bc2_sqrt);
        if (full) {
            reinterpret_cast<float4*>(p)[i4] = make_float4(pv[0], pv[1], pv[2], pv[3]);
            reinterpret_cast<float4*>(m)[i4] = make_float4(mv[0], mv[1], mv[2], mv[3]);
            reinterpret_cast<float4*>(v)[i4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            for (int k = 0; k < 4; k++) if (i + k < n) { p[i + k] = pv[k]; m[i + k] = mv[k]; v[i + k] = vv[k]; }
        }
    }
}

}  // namespace
}  // namespace moss

extern "C" int moss_adamw_multi(const moss_adamw_multi_args* a, void* stream)
{
    if (!a || a->num_tensors < 1 || a->num_tensors > 8) return MOSS_ERR_INVALID_ARG;
    moss::MultiT T; T.n = 0;
    unsigned blocks = 0u;
    for (int t = 0; t < 8; t++) { T.p[t] = nullptr; T.g[t] = nullptr; T.m[t] = nullptr; T.v[t] = nullptr; T.numel[t] = 0; T.lr[t] = 0.f; T.bc1[t] = 1.f; T.bc2_sqrt[t] = 1.f; T.block_end[t] = 0u; }
    for (int t = 0; t < a->num_tensors; t++) {
        const long long n = a->numel[t];
        if (n < 0 || a->step[t] < 1) return MOSS_ERR_INVALID_ARG;
        if (n == 0) continue;                                // (an empty tensor takes no block)
        if (!a->params[t] || !a->grads[t] || !a->exp_avg[t] || !a->exp_avg_sq[t]) return MOSS_ERR_INVALID_ARG;
        if ((((uintptr_t)a->params[t]) | ((uintptr_t)a->grads[t]) | ((uintptr_t)a->exp_avg[t]) | ((uintptr_t)a->exp_avg_sq[t])) & 15u) return MOSS_ERR_INVALID_ARG;
        const int k = T.n++;
        T.p[k] = a->params[t]; T.g[k] = a->grads[t]; T.m[k] = a->exp_avg[t]; T.v[k] = a->exp_avg_sq[t]; T.numel[k] = n; T.lr[k] = a->lr[t];
        // (the corrections exactly as moss_adamw_flat forms them: doubles, rounded once)
        T.bc1[k] = (float)(1.0 - pow(a->beta1, a->step[t])); T.bc2_sqrt[k] = (float)sqrt(1.0 - pow(a->beta2, a->step[t]));
        long long b = (n / 4 + 255) / 256;
        if (b > 1024) b = 1024;
        if (b < 1) b = 1;
        blocks += (unsigned)b; T.block_end[k] = blocks;
    }
    if (T.n == 0) return 0;
    hipLaunchKernelGGL(moss::adamw_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, T, moss::AdamBetas(a->beta1, a->beta2), a->eps, a->weight_decay);
    return hipGetLastError() == hipSuccess ? 0 : MOSS_ERR_HIP;
}

extern "C" int moss_adamw_flat(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               int num_segments, const long long* segment_end, const float* segment_lr,
                               const int* segment_period, const int* segment_split, const float* segment_lr2,
                               double beta1, double beta2, float eps, float weight_decay, int step, void* stream)
{
    if (n < 0 || num_segments < 1 || num_segments > 8 || !params || !grads || !exp_avg || !exp_avg_sq || !segment_end || !segment_lr || step < 1)
        return MOSS_ERR_INVALID_ARG;
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    return moss::launch_adamw(n, params, grads, exp_avg, exp_avg_sq, num_segments, segment_end, segment_lr, segment_period,
                              segment_split, segment_lr2, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), nullptr,
                              (hipStream_t)stream);
}

extern "C" int moss_adamw_flat_devstep(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                       int num_segments, const long long* segment_end, const float* segment_lr,
                                       const int* segment_period, const int* segment_split, const float* segment_lr2,
                                       double beta1, double beta2, float eps, float weight_decay, void* step_state, void* stream)
{
    if (n < 0 || num_segments < 1 || num_segments > 8 || !params || !grads || !exp_avg || !exp_avg_sq || !segment_end || !segment_lr || !step_state)
        return MOSS_ERR_INVALID_ARG;
    if (n == 0) return MOSS_ERR_INVALID_ARG;                 // the counter advances inside the update kernel: nothing to launch
    return moss::launch_adamw(n, params, grads, exp_avg, exp_avg_sq, num_segments, segment_end, segment_lr, segment_period,
                              segment_split, segment_lr2, beta1, beta2, eps, weight_decay, 1.f, 1.f, (const float*)step_state,
                              (hipStream_t)stream);
}

static_assert(moss::ADAMW_LR_VALID_WORD == MOSS_ADAMW_LR_VALID_WORD && moss::ADAMW_LR_WORD0 == MOSS_ADAMW_LR_WORD0 && moss::ADAMW_LR2_WORD0 == MOSS_ADAMW_LR2_WORD0, "adamw.h and the header disagree");
static_assert(moss::ADAMW_STATE_WORDS * 4 <= MOSS_ADAMW_STATE_BYTES, "the step-state block of adamw.h must fit the size the header promises");
extern "C" size_t moss_adamw_state_bytes(void) { return MOSS_ADAMW_STATE_BYTES; }

// A rank's SHARD of the flat bucket (reduce-scatter -> AdamW on 1/N of the elements -> all-gather of the parameters, moss_amd/dist.py):
// the arrays hold the elements [first, first + count) of the flat buffers the segment table (global indices) describes.
extern "C" int moss_adamw_flat_range(long long first, long long count, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                     int num_segments, const long long* segment_end, const float* segment_lr,
                                     const int* segment_period, const int* segment_split, const float* segment_lr2,
                                     double beta1, double beta2, float eps, float weight_decay, int step, void* step_state, void* stream)
{
    if (first < 0 || (first & 3) || count < 0 || num_segments < 1 || num_segments > 8 || !params || !grads || !exp_avg || !exp_avg_sq ||
        !segment_end || !segment_lr || (!step_state && step < 1))
        return MOSS_ERR_INVALID_ARG;
    if (count == 0) return step_state ? MOSS_ERR_INVALID_ARG : 0;
    const double bc1 = step_state ? 1.0 : 1.0 - pow(beta1, step), bc2 = step_state ? 1.0 : 1.0 - pow(beta2, step);
    return moss::launch_adamw(count, params, grads, exp_avg, exp_avg_sq, num_segments, segment_end, segment_lr, segment_period,
                              segment_split, segment_lr2, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2),
                              (const float*)step_state, (hipStream_t)stream, first);
}

extern "C" int moss_adamw_flat_guarded(long long first, long long count, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                       int num_segments, const long long* segment_end, const float* segment_lr,
                                       const int* segment_period, const int* segment_split, const float* segment_lr2,
                                       double beta1, double beta2, float eps, float weight_decay, void* step_state,
                                       const uint32_t* skip_word, uint32_t skip_mask, void* stream)
{
    if (first < 0 || (first & 3) || count <= 0 || num_segments < 1 || num_segments > 8 || !params || !grads || !exp_avg || !exp_avg_sq ||
        !segment_end || !segment_lr || !step_state || !skip_word)
        return MOSS_ERR_INVALID_ARG;
    return moss::launch_adamw(count, params, grads, exp_avg, exp_avg_sq, num_segments, segment_end, segment_lr, segment_period,
                              segment_split, segment_lr2, beta1, beta2, eps, weight_decay, 1.f, 1.f, (const float*)step_state,
                              (hipStream_t)stream, first, skip_word, skip_mask);
}


// Every form of the flat update behind ONE struct (ABI 6): moss_adamw_flat (step, no step_state), _devstep (step_state), _range
// (first / count), _guarded (skip_word) -- plus the degree-aware SH update (segment_active, inactive_zero: see Segs above).
extern "C" int moss_adamw_flat_ex(const moss_adamw_flat_args* a, void* stream)
{
    if (!a || a->first < 0 || (a->first & 3) || a->count < 0 || a->num_segments < 1 || a->num_segments > 8 || !a->params || !a->grads ||
        !a->exp_avg || !a->exp_avg_sq || !a->segment_end || !a->segment_lr || (!a->step_state && a->step < 1) || (a->skip_word && !a->step_state) ||
        a->num_grads_extra < 0 || a->num_grads_extra > 3)
        return MOSS_ERR_INVALID_ARG;
    for (int i = 0; i < a->num_grads_extra; i++) if (!a->grads_extra[i]) return MOSS_ERR_INVALID_ARG;
    if (a->count == 0) return a->step_state ? MOSS_ERR_INVALID_ARG : 0;
    const double bc1 = a->step_state ? 1.0 : 1.0 - pow(a->beta1, a->step), bc2 = a->step_state ? 1.0 : 1.0 - pow(a->beta2, a->step);
    return moss::launch_adamw(a->count, a->params, a->grads, a->exp_avg, a->exp_avg_sq, a->num_segments, a->segment_end, a->segment_lr,
                              a->segment_period, a->segment_split, a->segment_lr2, a->beta1, a->beta2, a->eps, a->weight_decay,
                              (float)bc1, (float)sqrt(bc2), (const float*)a->step_state, (hipStream_t)stream, a->first, a->skip_word,
                              a->skip_mask, a->segment_active, a->inactive_zero, a->num_grads_extra, a->grads_extra, a->grad_scale);
}
