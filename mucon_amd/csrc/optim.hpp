// Gradient clipping + SGD step over a list of tensors in two launches (the tail of the reference's training step:
// src/mucon/trainers.py:137-140 -- clip_grad_norm_(encode_params), clip_grad_norm_(decode_params), optimizer.step()
// with torch.optim.SGD(lr, momentum, weight_decay), trainers.py:18-30).  torch runs this as ~25 small multi-tensor
// launches (0.86 ms of launch latency per step for a 1.6 M-parameter model).
//   sgd_norm_kernel    sum of squares per 4096-float chunk -> partial[block]
//   sgd_apply_kernel   every block re-reduces the partials of its clipping group in a fixed order (so all blocks
//                      agree bitwise), coef = min(max_norm / (norm + 1e-6), 1)  [torch.nn.utils.clip_grad_norm_],
//                      g *= coef (written back, as torch clips in place), g' = g + wd * p,
//                      buf = momentum * buf + g' (optional), p -= lr * g'
#pragma once
#include "common.hpp"

constexpr int SGD_CHUNK = 4096;
constexpr int SGD_MAXGROUPS = 8;

struct SgdTensor {   // mirrors mucon_sgd_tensor + the launch bookkeeping
    float *p, *g, *mom;
    long n;
    int group, block0;
};

template <class TAB>
__device__ __forceinline__ int sgd_find(const TAB *tab, int nt, int block) {
    int lo = 0, hi = nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block0 <= block) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ float sgd_block_sum(float v, float *red) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Which tensor a workgroup belongs to, and which clipping group a block's partial sum counts for: the binary search over
// the table in global memory is a chain of ~7 dependent L2 round trips (4 - 5 us in front of a 10 us kernel).  The two columns
// it needs are copied into LDS once (one round trip, all threads in parallel) and searched there.
constexpr int SGD_LDS_TAB = 1024;
struct SgdIndex {
    int *b0, *grp;
    int nt;
    bool lds;
    template <class TAB>
    __device__ __forceinline__ void init(const TAB *tab, int n, int *s_b0, int *s_grp) {
        b0 = s_b0;
        grp = s_grp;
        nt = n;
        lds = n <= SGD_LDS_TAB;
        if (lds) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                s_b0[i] = tab[i].block0;
                s_grp[i] = tab[i].group;
            }
            __syncthreads();
        }
    }
    template <class TAB>
    __device__ __forceinline__ int find(const TAB *tab, int block) const {
        if (!lds) return sgd_find(tab, nt, block);
        int lo = 0, hi = nt - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (b0[mid] <= block) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    }
    template <class TAB>
    __device__ __forceinline__ int group_of_block(const TAB *tab, int block) const {
        const int i = find(tab, block);
        return lds ? grp[i] : tab[i].group;
    }
};

template <class TAB>
__global__ __launch_bounds__(256) void sgd_norm_kernel(const TAB *tab, int nt, float *partial) {
    __shared__ float red[4];
    __shared__ int s_b0[SGD_LDS_TAB], s_grp[SGD_LDS_TAB];
    SgdIndex ix;
    ix.init(tab, nt, s_b0, s_grp);
    const TAB t = tab[ix.find(tab, blockIdx.x)];
    const long b0 = (long)(blockIdx.x - t.block0) * SGD_CHUNK;
    const long b1 = min(t.n, b0 + SGD_CHUNK);
    float acc = 0.f;
    const bool vec = (reinterpret_cast<uintptr_t>(t.g) & 15) == 0;
    const long nvec = vec ? (b1 - b0) >> 2 : 0;   // whole float4 of the chunk: four per thread, all loads first
    if (nvec > 0) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4 *>(t.g + b0 + 4 * min((long)threadIdx.x + 256 * j, nvec - 1));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((long)threadIdx.x + 256 * j < nvec) acc += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
    }
    for (long e = b0 + 4 * nvec + threadIdx.x; e < b1; e += 256) {
        const float g = t.g[e];
        acc += g * g;
    }
    acc = sgd_block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

struct SgdHyper {
    float max_norm[SGD_MAXGROUPS];   // <= 0: no clipping for the group
    float lr, weight_decay, momentum;
    int ngroups, nblocks;
    int any_clip;                    // 0: no group clips -- the norm pass is skipped altogether
};

__global__ __launch_bounds__(256) void sgd_apply_kernel(const SgdTensor *tab, int nt, const float *partial, SgdHyper h,
                                                        float *norms_out) {
    __shared__ float red[4];
    __shared__ int s_b0[SGD_LDS_TAB], s_grp[SGD_LDS_TAB];
    SgdIndex ix;
    ix.init(tab, nt, s_b0, s_grp);
    const SgdTensor t = tab[ix.find(tab, blockIdx.x)];
    // squared norm of this block's group: partials of every tensor of the group, strided over the threads
    // (thread-strided over ALL blocks, each looked up in the table: walking the tensors one after the other made every
    // block of a 75-tensor model run 75 short dependent loops -- 14 us of the 24 us this kernel took there)
    float acc = 0.f;
    if (h.any_clip) {
        for (int b = threadIdx.x; b < h.nblocks; b += 256)
            if (ix.group_of_block(tab, b) == t.group) acc += partial[b];
    }
    const float norm = h.any_clip ? sqrtf(sgd_block_sum(acc, red)) : 0.f;
    const float mx = h.max_norm[t.group];
    const float coef = mx > 0.f ? fminf(mx / (norm + 1e-6f), 1.f) : 1.f;
    if (blockIdx.x == t.block0 && norms_out) {   // (uniform over the workgroup)
        // the first block of the group's first tensor reports the norm: no earlier tensor of the same group -- asked of all
        // table entries at once (thread i <-> tensor i; one thread walking the table was a chain of up to nt dependent loads
        // in the blocks that finish last)
        bool earlier = false;
        if (ix.lds) {
            for (int i = threadIdx.x; i < nt; i += 256) earlier |= ix.b0[i] < t.block0 && ix.grp[i] == t.group;
        } else if (threadIdx.x == 0) {
            for (int i = 0; i < nt && tab[i].block0 < t.block0; ++i) earlier |= tab[i].group == t.group;
        }
        if (!__syncthreads_or(earlier) && threadIdx.x == 0) {
            norms_out[t.group] = norm;
            // STICKY: the count of steps of this group that were skipped for a non-finite norm, never cleared by the library -- the next healthy
            // step overwrites the norm above, not this (one writer per group and launch, launches ordered by the stream: no atomic needed)
            if (mx > 0.f && !(norm < __builtin_inff())) norms_out[h.ngroups + t.group] += 1.f;
        }
    }
    // A clipped group whose gradient norm is not finite (a NaN / inf gradient: a diverged step, or a kernel that reported failure by
    // poisoning its outputs -- the eight-workgroup decoder's hand-over time-out) is NOT applied: parameters, gradients and momentum stay as
    // they are, norms_out[group] carries the non-finite norm and norms_out[ngroups + group] counts the skipped step (sticky: later healthy steps do
    // not clear it), which ops.check_health() raises on at the caller's next synchronisation point.
    // (torch would scale every gradient of the group by NaN and write NaN into every parameter; the reference has no such guard.)
    if (mx > 0.f && !(norm < __builtin_inff())) return;   // (uniform over the workgroup)
    const long b0 = (long)(blockIdx.x - t.block0) * SGD_CHUNK;
    const long b1 = min(t.n, b0 + SGD_CHUNK);
    auto update = [&](float g, float p, float m, float &g_out, float &p_out, float &m_out) {
        g *= coef;
        g_out = g;
        float u = g + h.weight_decay * p;
        if (t.mom) {
            u = h.momentum * m + u;
            m_out = u;
        }
        p_out = p - h.lr * u;
    };
    // a chunk is 4096 floats = 4 float4 per thread: all loads of the chunk first, then the arithmetic, then the stores (the
    // element-at-a-time loop was a chain of 16 dependent memory round trips per thread: 20 us for 1 M parameters)
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                       reinterpret_cast<uintptr_t>(t.mom ? t.mom : t.p)) & 15) == 0;
    const long nvec = vec ? (b1 - b0) >> 2 : 0;   // whole float4 of the chunk (b0 is a multiple of 4096)
    if (nvec > 0) {
        f32x4 vg[4], vp[4], vm[4];
        long q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            q[j] = b0 + 4 * min((long)threadIdx.x + 256 * j, nvec - 1);   // clamped: loads are never under a branch
            vg[j] = *reinterpret_cast<const f32x4 *>(t.g + q[j]);
            vp[j] = *reinterpret_cast<const f32x4 *>(t.p + q[j]);
            vm[j] = t.mom ? *reinterpret_cast<const f32x4 *>(t.mom + q[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 og, op, om = vm[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a, b, c = vm[j][e];
                update(vg[j][e], vp[j][e], vm[j][e], a, b, c);
                og[e] = a;
                op[e] = b;
                om[e] = c;
            }
            if ((long)threadIdx.x + 256 * j < nvec) {
                *reinterpret_cast<f32x4 *>(t.g + q[j]) = og;
                *reinterpret_cast<f32x4 *>(t.p + q[j]) = op;
                if (t.mom) *reinterpret_cast<f32x4 *>(t.mom + q[j]) = om;
            }
        }
    }
    for (long e = b0 + 4 * nvec + threadIdx.x; e < b1; e += 256) {   // the tail, or everything when a pointer is unaligned
        float og, op, om = 0.f;
        update(t.g[e], t.p[e], t.mom ? t.mom[e] : 0.f, og, op, om);
        t.g[e] = og;
        t.p[e] = op;
        if (t.mom) t.mom[e] = om;
    }
}

// ---- the clipping alone (an iteration of a gradient-accumulation group that does not step: reference trainers.py:131-147) ----
__global__ __launch_bounds__(256) void clip_apply_kernel(const SgdTensor *tab, int nt, const float *partial, SgdHyper h, float *norms_out) {
    __shared__ float red[4];
    __shared__ int s_b0[SGD_LDS_TAB], s_grp[SGD_LDS_TAB];
    SgdIndex ix;
    ix.init(tab, nt, s_b0, s_grp);
    const SgdTensor t = tab[ix.find(tab, blockIdx.x)];
    float acc = 0.f;
    for (int b = threadIdx.x; b < h.nblocks; b += 256)
        if (ix.group_of_block(tab, b) == t.group) acc += partial[b];
    const float norm = sqrtf(sgd_block_sum(acc, red));
    const float mx = h.max_norm[t.group];
    const float coef = mx > 0.f ? fminf(mx / (norm + 1e-6f), 1.f) : 1.f;
    if (blockIdx.x == t.block0 && norms_out) {
        bool earlier = false;
        if (ix.lds) {
            for (int i = threadIdx.x; i < nt; i += 256) earlier |= ix.b0[i] < t.block0 && ix.grp[i] == t.group;
        } else if (threadIdx.x == 0) {
            for (int i = 0; i < nt && tab[i].block0 < t.block0; ++i) earlier |= tab[i].group == t.group;
        }
        if (!__syncthreads_or(earlier) && threadIdx.x == 0) {
            norms_out[t.group] = norm;
            // STICKY: the count of steps of this group that were skipped for a non-finite norm, never cleared by the library -- the next healthy
            // step overwrites the norm above, not this (one writer per group and launch, launches ordered by the stream: no atomic needed)
            if (mx > 0.f && !(norm < __builtin_inff())) norms_out[h.ngroups + t.group] += 1.f;
        }
    }
    if (coef == 1.f || (mx > 0.f && !(norm < __builtin_inff()))) return;    // torch multiplies by a clamped 1.0 too: the same bits; a non-finite norm is left for check_health
    const long b0 = (long)(blockIdx.x - t.block0) * SGD_CHUNK;
    const long b1 = min(t.n, b0 + SGD_CHUNK);
    for (long e = b0 + threadIdx.x; e < b1; e += 256) t.g[e] *= coef;
}

// ---- Adam (torch.optim.Adam, optionally AMSGrad) behind the same group-wise clipping -----------------------------------------
//   g *= coef (written back);  u = g + wd * p;  m += (u - m) (1 - beta1);  v = v beta2 + (1 - beta2) u u;
//   vmax = max(vmax, v) (amsgrad);  p -= step_size * m / (sqrt(vmax or v) / sqrt(1 - beta2^t) + eps),  step_size = lr / (1 - beta1^t)
// -- the element-wise sequence of torch's _single_tensor_adam; the scalars are formed on the host in double, as torch forms them.
struct AdamTensor {
    float *p, *g, *m, *v, *vmax;   // vmax: null without amsgrad
    long n;
    int group, block0;
};
struct AdamHyper {
    float max_norm[SGD_MAXGROUPS];
    float step_size, one_minus_beta1, beta2, one_minus_beta2, sqrt_bc2, eps, weight_decay;
    int ngroups, nblocks, any_clip;
};
__global__ __launch_bounds__(256) void adam_apply_kernel(const AdamTensor *tab, int nt, const float *partial, AdamHyper h, float *norms_out) {
    __shared__ float red[4];
    __shared__ int s_b0[SGD_LDS_TAB], s_grp[SGD_LDS_TAB];
    SgdIndex ix;
    ix.init(tab, nt, s_b0, s_grp);
    const AdamTensor t = tab[ix.find(tab, blockIdx.x)];
    float acc = 0.f;
    if (h.any_clip) {
        for (int b = threadIdx.x; b < h.nblocks; b += 256)
            if (ix.group_of_block(tab, b) == t.group) acc += partial[b];
    }
    const float norm = h.any_clip ? sqrtf(sgd_block_sum(acc, red)) : 0.f;
    const float mx = h.max_norm[t.group];
    const float coef = mx > 0.f ? fminf(mx / (norm + 1e-6f), 1.f) : 1.f;
    if (blockIdx.x == t.block0 && norms_out) {
        bool earlier = false;
        if (ix.lds) {
            for (int i = threadIdx.x; i < nt; i += 256) earlier |= ix.b0[i] < t.block0 && ix.grp[i] == t.group;
        } else if (threadIdx.x == 0) {
            for (int i = 0; i < nt && tab[i].block0 < t.block0; ++i) earlier |= tab[i].group == t.group;
        }
        if (!__syncthreads_or(earlier) && threadIdx.x == 0) {
            norms_out[t.group] = norm;
            // STICKY: the count of steps of this group that were skipped for a non-finite norm, never cleared by the library -- the next healthy
            // step overwrites the norm above, not this (one writer per group and launch, launches ordered by the stream: no atomic needed)
            if (mx > 0.f && !(norm < __builtin_inff())) norms_out[h.ngroups + t.group] += 1.f;
        }
    }
    if (mx > 0.f && !(norm < __builtin_inff())) return;   // non-finite norm of a clipped group: the step is not applied (see sgd_apply_kernel)
    const long b0 = (long)(blockIdx.x - t.block0) * SGD_CHUNK;
    const long b1 = min(t.n, b0 + SGD_CHUNK);
    // four elements per thread and round, all loads of a round issued before its arithmetic
    for (long base = b0; base < b1; base += 1024) {
        float g[4], p[4], m[4], v[4], vm[4];
        long e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = min(base + threadIdx.x + 256 * j, b1 - 1);          // clamped: loads are never under a branch
            g[j] = t.g[e[j]];
            p[j] = t.p[e[j]];
            m[j] = t.m[e[j]];
            v[j] = t.v[e[j]];
            vm[j] = t.vmax ? t.vmax[e[j]] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (base + threadIdx.x + 256 * j >= b1) continue;
            const float gc = g[j] * coef;
            const float u = h.weight_decay != 0.f ? gc + h.weight_decay * p[j] : gc;
            const float mn = m[j] + (u - m[j]) * h.one_minus_beta1;
            const float vn = v[j] * h.beta2 + h.one_minus_beta2 * u * u;
            const float vx = t.vmax ? fmaxf(vm[j], vn) : vn;
            const float denom = sqrtf(vx) / h.sqrt_bc2 + h.eps;
            t.g[e[j]] = gc;
            t.m[e[j]] = mn;
            t.v[e[j]] = vn;
            if (t.vmax) t.vmax[e[j]] = vx;
            t.p[e[j]] = p[j] - h.step_size * (mn / denom);
        }
    }
}
