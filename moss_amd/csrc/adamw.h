// adamw.h -- the AdamW element update and the device-resident step counter, shared by the flat update kernel (optim.hip) and by the
// per-Gaussian backward kernel when it applies the update itself (preprocess.hip, moss_raster_backward_raw_adamw).  Both must give
// the same bits for the same inputs whatever their translation unit's contraction flags, so every rounding is spelled out.
// Semantics = torch.optim.AdamW (amsgrad=False, maximize=False): decoupled weight decay, bias-corrected moments
// (MOSS: scene/gaussian_model.py:215-226).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace moss {

// Words of the step-state block (int32 / float32 views of the same memory; moss_adamw_state_bytes()):
//   [0] step count t, [8..11] cached bias corrections {1 - beta1^t, sqrt(1 - beta2^t)} for the two parities of t,
//   [64] global completion counter, [128 + 64 g] completion counter of block group g -- every counter on a 256-byte line of its own.
//   [12] != 0: the learning rates are read from the block too -- [16 + s] = lr of optimizer segment s, [24 + s] = its second rate
//   (periodic pattern) -- instead of from the launch arguments: a schedule (MOSS decays the position rate every iteration,
//   scene/gaussian_model.py:263-268, train_ZJU.py:82) then needs no re-capture of a hipGraph that has the arguments baked in.
constexpr int ADAMW_LR_VALID_WORD = 12, ADAMW_LR_WORD0 = 16, ADAMW_LR2_WORD0 = 24;
constexpr int ADAMW_NGROUPS = 32;
constexpr int ADAMW_STATE_WORDS = 128 + 64 * ADAMW_NGROUPS;

// p, m, v <- one AdamW step with gradient g.  (Per element a hardware square root and reciprocal, ~1 ulp each: with correctly rounded
// sqrt / divisions the update was ~40 vector instructions per element, as much issue time as the flat kernel's bytes are HBM time.)
// The betas arrive as DOUBLES at the C ABI and reach the kernels as two independently rounded floats each, beta and 1 - beta
// (AdamBetas): torch's kernels round the Python doubles `beta2` and `1 - beta2` separately too, and float(1 - 0.999) = 0.001 is not
// 1 - float(0.999) = 0.00099998713 -- a 1.3e-5 relative bias of the second moment with the rounds 1-3 form of this function.
struct AdamBetas {
    float b1 = 0.9f, b2 = 0.999f, omb1 = 0.1f, omb2 = 0.001f;
    double b1_d = 0.9, b2_d = 0.999;
    AdamBetas() = default;
    AdamBetas(double beta1, double beta2) : b1((float)beta1), b2((float)beta2), omb1((float)(1.0 - beta1)), omb2((float)(1.0 - beta2)), b1_d(beta1), b2_d(beta2) {}
};

__device__ __forceinline__ void adamw_element(float& p, float g, float& m, float& v, float lr, const AdamBetas& B, float eps,
                                              float weight_decay, float inv_bc1, float inv_bc2_sqrt)
{
    p = __fmul_rn(p, __fsub_rn(1.0f, __fmul_rn(lr, weight_decay)));
    m = __fmaf_rn(B.b1, m, __fmul_rn(B.omb1, g));
    v = __fmaf_rn(B.b2, v, __fmul_rn(__fmul_rn(B.omb2, g), g));
    const float denom = __fmaf_rn(__builtin_amdgcn_sqrtf(v), inv_bc2_sqrt, eps);
    p = __fmaf_rn(-__fmul_rn(lr, inv_bc1), __fmul_rn(m, __builtin_amdgcn_rcpf(denom)), p);
}

// Start of a step with the counter on the device (graph replay): returns t (the step being taken) and its bias corrections.  The
// reads are wave-uniform (scalar loads).  `writer` (ONE thread of the launch, at its start) caches the NEXT step's corrections in
// the other slot, which nobody reads during this launch (two double pow() at the end of the last block were a 3-5 us serial tail);
// a launch that may still turn out to be a no-op passes writer = false and calls adamw_cache_next once it knows (a skipped step
// must leave the whole block bit for bit).
__device__ __forceinline__ void adamw_cache_next(const float* step_state, const AdamBetas& B, int t)
{
    float* sf = const_cast<float*>(step_state);
    sf[8 + 2 * ((t + 1) & 1)] = (float)(1.0 - pow(B.b1_d, (double)(t + 1)));
    sf[9 + 2 * ((t + 1) & 1)] = (float)sqrt(1.0 - pow(B.b2_d, (double)(t + 1)));
}

// (t_prev: word 0 of the block if the caller has loaded it already -- together with other words, so that the loads share one round trip)
__device__ __forceinline__ int adamw_step_begin(const float* step_state, const AdamBetas& B, bool writer, float& bc1, float& bc2_sqrt,
                                                int t_prev = -1)
{
    const int t = (t_prev >= 0 ? t_prev : reinterpret_cast<const int*>(step_state)[0]) + 1;
    if (t == 1) {                                            // first step ever: nothing cached yet (pow(x, 1) = x exactly)
        bc1 = (float)(1.0 - B.b1_d);
        bc2_sqrt = (float)sqrt(1.0 - B.b2_d);
    } else {                                                 // cached by the previous step (slot = parity of the step)
        bc1 = step_state[8 + 2 * (t & 1)]; bc2_sqrt = step_state[9 + 2 * (t & 1)];
    }
    if (writer) adamw_cache_next(step_state, B, t);
    return t;
}

// End of a step: called by ONE thread of every block after the block's last use of the step word.  The LAST block of the launch
// stores t back (after every block has read the old value): no separate "tick" launch (a minimal launch costs 4-5 us here).
// Two-level completion count, every counter on a 256-byte line of its own: atomics on words of ONE cache line execute one after the
// other for the whole device (~11 ns each, profiles/r02_notes.md finding 1).
__device__ __forceinline__ void adamw_step_end(const float* step_state, int t)
{
    int* st = reinterpret_cast<int*>(const_cast<float*>(step_state));
    const int grp = (int)(blockIdx.x % (unsigned)ADAMW_NGROUPS);
    const int grp_size = ((int)gridDim.x - grp + ADAMW_NGROUPS - 1) / ADAMW_NGROUPS;      // blocks with this group id
    const int n_groups = min((int)gridDim.x, ADAMW_NGROUPS);
    if (atomicAdd(&st[128 + 64 * grp], 1) == grp_size - 1) {
        st[128 + 64 * grp] = 0;
        if (atomicAdd(&st[64], 1) == n_groups - 1) {
            st[0] = t; st[64] = 0;
        }
    }
}

// The update of the Gaussian parameters as the per-Gaussian backward kernel applies it (include/moss_raster.h: moss_fused_adamw).
// Order of the five tensors: means, sh, opacity, scales, rotations.
constexpr uint32_t OPT_MEANS = 1u, OPT_SH = 2u, OPT_OPACITY = 4u, OPT_SCALES = 8u, OPT_ROTATIONS = 16u;
struct FusedAdam {
    uint32_t tensors = 0u;                                   // which tensors the kernel updates (0: none, a plain backward)
    float* p[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };       // the parameters themselves (= the op's inputs), written in place
    float* m[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    float* v[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    float lr[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };
    float lr_sh_rest = 0.f;                                  // sh: lr[1] for a Gaussian's first 3 floats (features_dc), this for the other 45
    // DEGREE-AWARE SH update (include/moss_raster.h: moss_fused_adamw.sh_active_degree): of a record's 12 float4 only the first
    // sh_active_parts hold a coefficient that has ever received a gradient (degree 0 / 1 / 2 / 3: 1 / 3 / 7 / 12); the float4 behind
    // them have zero moments -- not read, not written -- and take the weight decay alone, or nothing at all when the caller knows their
    // parameters to be exactly zero (sh_inactive_zero).  The same bits as the full update.
    int sh_active_parts = 12, sh_inactive_zero = 0;
    int lr_segment[5] = { -1, -1, -1, -1, -1 };              // each tensor's entry in the step-state block's learning-rate table (-1: none)
    AdamBetas betas;
    float eps = 1e-15f, weight_decay = 0.f;
    const float* step_state = nullptr;
};

}  // namespace moss
