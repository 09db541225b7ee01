// What does it cost to keep a run of same-level residual layers inside ONE launch on gfx950?  
//
// The coarse levels of the encoder are chains of short launches (cs_kernel, csrc/gemm_coarse_split.hpp: 16 rows per workgroup, the
// four waves split K, every wave streams its quarter of the layer's 288 KB of bf16 weight fragments from L2 straight into MFMA
// operand registers).  Between two layers of a level lies a kernel boundary; a persistent kernel would replace it by a
// point-to-point hand-over: tile i of layer l + 1 needs tiles i - s, i, i + s of layer l (s = dilation / 16, inside the video).
// This probe times that choice with a stand-in layer of cs_kernel's shape -- same grid (256 workgroups of 256 threads at
// T/8 of the bench batch), same loads (3 taps x 16 rows x 512 B of activations, 288 KB of weight fragments through an
// eight-deep register ring), same MFMA count (144 per wave), one LDS reduction, one 8 KB tile stored -- under four protocols:
//   launches   one launch per layer, plain loads and stores                              (what the library does)
//   wbl2       one launch; producer: plain stores, s_waitcnt, barrier, agent-scope RELEASE fence (buffer_wbl2), flag;
//              consumer: poll, agent-scope ACQUIRE fence (buffer_inv), plain loads        (r3's persistent kernel)
//   sc1        one launch; producer: write-through (sc0 sc1) stores, s_waitcnt vmcnt(0), barrier, sc1 flag store;
//              consumer: sc1 poll, barrier, sc0 sc1 loads of the three tiles              (the guide's publish-large row)
//   sc1+pf     sc1, and the first eight weight-fragment triples of the NEXT layer are requested before the wave starts to
//              wait for its producers (weights do not depend on activations)
//   xcd        (r6) sc1's flags and polls, but the tiles of a video are the workgroups with equal blockIdx % 8 -- ONE XCD under the observed round-robin
//              placement -- and the producer's stores are PLAIN: they stay in that XCD's L2, where the consumer's sc1 (L1-bypassing) loads find them.  Not a
//              placement-independent protocol: every workgroup records its HW_REG_XCC_ID and the host reports whether each video sat on one XCD; a
//              product kernel would have to fall back to sc1 stores when a consumer's XCD differs (flags carry the producer's XCC id).
//   xcd+pf     xcd with sc1+pf's weight request behind the publication
// Every protocol must produce the same bits as `launches` (the stand-in arithmetic is deterministic): a stale read shows up as
// a checksum mismatch.  Correctness never relies on workgroup -> XCD placement.
//
//   hipcc --offload-arch=gfx950 -O3 tools/coarse_protocol_probe.hip -o tools/build/coarse_protocol_probe && tools/build/coarse_protocol_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

constexpr int ROWS_PER_VIDEO = 512;      // T/8 of T = 4096
constexpr int VIDEOS = 8;
constexpr int TILES_PER_VIDEO = ROWS_PER_VIDEO / 16;
constexpr int NT = VIDEOS * TILES_PER_VIDEO;   // 256 workgroups
constexpr int WSTEP = 3 * 128 * 32;             // bf16 elements of one 32-deep k-step image: 3 planes x 128 channels x 32
constexpr int LAYER_ELEMS = 12 * WSTEP;         // 3 taps x 4 steps: 288 KB
constexpr int MAXL = 8;

struct Layer {
    const float *in;      // [rows][128]
    float *out;
    const uint16_t *W;    // LAYER_ELEMS
    int tap_tiles;        // dilation / 16
};
struct Args {
    Layer l[MAXL];
    int n;
    unsigned *flags;      // [MAXL][NT], zero before the launch
    unsigned seq;         // value a finished tile publishes
};

enum { P_LAUNCH = 0, P_WBL2 = 1, P_SC1 = 2, P_SC1_PF = 3, P_XCD = 4, P_XCD_PF = 5 };

__device__ __forceinline__ f32x4 load_plain(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 load_sc1(const float *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_sc1(float *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

template <int PROTO>
__device__ __forceinline__ void layer_body(const Layer &L, const int tile, const int video, unsigned *flags_prev, unsigned *flags_mine,
                                           const unsigned seq, const bool first, f32x4 *red, bf16x8 (&wf)[8][3], const bool ring_primed,
                                           const Layer *next) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    const uint16_t *w1 = L.W + (long)w * WSTEP + lane * 8;
    auto loadW = [&](const uint16_t *base, int i, int slot) {
        const uint16_t *src = base + (long)(i >> 3) * (4 * WSTEP) + (i & 7) * 512;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[slot][pl] = *reinterpret_cast<const bf16x8 *>(src + pl * (128 * 32));
    };
    // the first eight fragment triples: a launch requests them as it starts; the persistent protocols only once their inputs are
    // known to be there (a launch boundary gives nothing earlier either) -- except sc1+pf, where the previous layer's loop tail
    // already requested them (ring_primed)
    if (!ring_primed && (PROTO == P_LAUNCH || first)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) loadW(w1, i, i);
    }
    // ---- wait for the three producer tiles of the previous layer
    if (PROTO != P_LAUNCH && !first) {
        if (tid < 3) {
            const int t = (tile & (TILES_PER_VIDEO - 1)) + (tid - 1) * L.tap_tiles;
            if (t >= 0 && t < TILES_PER_VIDEO) {
                const unsigned *f = flags_prev + video * TILES_PER_VIDEO + t;
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(2);
            }
        }
        if (PROTO == P_WBL2) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!ring_primed) {
#pragma unroll
            for (int i = 0; i < 8; ++i) loadW(w1, i, i);
        }
    }
    // ---- activation slices: tap i, rows of tile (tile + (i - 1) s), channels 32 w + 8 g ..
    f32x4 ra[3][2];
    bool rok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int t = (tile & (TILES_PER_VIDEO - 1)) + (i - 1) * L.tap_tiles;
        rok[i] = t >= 0 && t < TILES_PER_VIDEO;
        const int tt = rok[i] ? t : (tile & (TILES_PER_VIDEO - 1));
        const float *src = L.in + ((long)(video * TILES_PER_VIDEO + tt) * 16 + c) * 128 + 32 * w + 8 * g;
        if (PROTO >= P_SC1 && !first) {
            ra[i][0] = load_sc1(src);
            ra[i][1] = load_sc1(src + 4);
        } else {
            ra[i][0] = load_plain(src);
            ra[i][1] = load_plain(src + 4);
        }
    }
    // residual operand: this workgroup's own tile of the previous layer, channel blocks 2 w, 2 w + 1
    f32x4 res[2];
    const long grow = ((long)(video * TILES_PER_VIDEO + (tile & (TILES_PER_VIDEO - 1))) * 16 + c) * 128 + 4 * g;
#pragma unroll
    for (int j = 0; j < 2; ++j) res[j] = (PROTO >= P_SC1 && !first) ? load_sc1(L.in + grow + 16 * (2 * w + j)) : load_plain(L.in + grow + 16 * (2 * w + j));
    if (PROTO >= P_SC1 && !first)   // (asm loads are invisible to the compiler's wait counting: the wait takes the loaded registers as operands, so nothing that uses them moves in front of it)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]), "+v"(res[0]), "+v"(res[1])::"memory");
    bf16x8 xa[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = e < 4 ? ra[i][0][e] : ra[i][1][e - 4];
            xa[i][e] = (__bf16)(rok[i] ? x : 0.f);
        }
    }
    f32x4 acc[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 24; ++i) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {          // six MFMAs per pair in the product kernel: two per plane here
            acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i & 7][pl], xa[i >> 3], acc[i & 7], 0, 0, 0);
            acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i & 7][pl], xa[(i >> 3) == 2 ? 0 : (i >> 3) + 1], acc[i & 7], 0, 0, 0);
        }
        if (i + 8 < 24) loadW(w1, i + 8, i & 7);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) red[(w * 8 + nb) * 64 + lane] = acc[nb];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nb = 2 * w + j;
        f32x4 x = ((red[(0 * 8 + nb) * 64 + lane] + red[(1 * 8 + nb) * 64 + lane]) + red[(2 * 8 + nb) * 64 + lane]) + red[(3 * 8 + nb) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = res[j][e] + 0.001f * fmaxf(x[e], 0.f);
        if (PROTO == P_SC1 || PROTO == P_SC1_PF) store_sc1(L.out + grow + 16 * nb, x);
        else *reinterpret_cast<f32x4 *>(L.out + grow + 16 * nb) = x;
    }
    // ---- publish
    if (PROTO != P_LAUNCH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its own stores
        __syncthreads();                                           // (also frees `red` for the next layer)
        if (tid == 0) {
            if (PROTO == P_WBL2) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(flags_mine + video * TILES_PER_VIDEO + (tile & (TILES_PER_VIDEO - 1)), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // sc1+pf: the next layer's first ring, requested behind the publication (in front of it the drain above would wait for it)
        if ((PROTO == P_SC1_PF || PROTO == P_XCD_PF) && next) {
            const uint16_t *wn = next->W + (long)w * WSTEP + lane * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) loadW(wn, i, i);
        }
    } else {
        __syncthreads();
    }
}

__device__ unsigned g_xcc[NT];
template <int PROTO>
__global__ __launch_bounds__(256) void probe_kernel(const Args a, const int only_layer) {
    __shared__ f32x4 red[4 * 8 * 64];
    bf16x8 wf[8][3];
    // xcd: video = blockIdx % 8 (the workgroups that share an XCD under round-robin placement), tile inside it = blockIdx / 8
    const int video = PROTO >= P_XCD ? (int)(blockIdx.x & 7) : (int)(blockIdx.x / TILES_PER_VIDEO);
    const int tile = PROTO >= P_XCD ? video * TILES_PER_VIDEO + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (PROTO >= P_XCD && threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
        g_xcc[tile] = x;
    }
    if (PROTO == P_LAUNCH) {
        layer_body<PROTO>(a.l[only_layer], tile, video, nullptr, nullptr, 0, true, red, wf, false, nullptr);
        return;
    }
    for (int l = 0; l < a.n; ++l)
        layer_body<PROTO>(a.l[l], tile, video, a.flags + (l ? l - 1 : 0) * NT, a.flags + l * NT, a.seq, l == 0, red, wf,
                          (PROTO == P_SC1_PF || PROTO == P_XCD_PF) && l > 0, l + 1 < a.n ? &a.l[l + 1] : nullptr);
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    int dev = 0;
    CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    printf("# %s, %d CUs; %d workgroups of 256 threads, 16 rows each (%d videos x %d rows), %d KB of weight fragments per layer\n", prop.name,
           prop.multiProcessorCount, NT, VIDEOS, ROWS_PER_VIDEO, LAYER_ELEMS * 2 / 1024);
    const size_t act_elems = (size_t)NT * 16 * 128;
    float *act[MAXL + 1];
    for (int i = 0; i <= MAXL; ++i) CHECK(hipMalloc(&act[i], act_elems * 4));
    uint16_t *W;
    CHECK(hipMalloc(&W, (size_t)MAXL * LAYER_ELEMS * 2));
    unsigned *flags;
    CHECK(hipMalloc(&flags, sizeof(unsigned) * MAXL * NT));
    CHECK(hipMemset(flags, 0, sizeof(unsigned) * MAXL * NT));
    {
        std::vector<float> h(act_elems);
        unsigned s = 12345;
        for (auto &x : h) {
            s = s * 1664525u + 1013904223u;
            x = ((s >> 8) & 0xffff) / 65536.f - 0.5f;
        }
        CHECK(hipMemcpy(act[0], h.data(), act_elems * 4, hipMemcpyHostToDevice));
        std::vector<uint16_t> hw((size_t)MAXL * LAYER_ELEMS);
        for (auto &x : hw) {
            s = s * 1664525u + 1013904223u;
            const float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f;
            uint32_t u;
            memcpy(&u, &f, 4);
            x = (uint16_t)(u >> 16);
        }
        CHECK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[6] = {"launches", "wbl2", "sc1", "sc1+pf", "xcd", "xcd+pf"};
    // runs of same-level layers as the bench shape has them: T/8 = layers 5..8 (dilations 32 .. 256 rows: 2, 4, 8, 16 tiles), and a
    // two-layer run (T/4-like spacing on this grid) for the short case
    const int runs[2][5] = {{4, 2, 4, 8, 16}, {2, 1, 2, 0, 0}};
    for (int r = 0; r < 2; ++r) {
        const int n = runs[r][0];
        Args a;
        memset(&a, 0, sizeof(a));
        a.n = n;
        a.flags = flags;
        for (int l = 0; l < n; ++l) a.l[l] = Layer{act[l], act[l + 1], W + (size_t)l * LAYER_ELEMS, runs[r][1 + l]};
        std::vector<float> ref(act_elems), got(act_elems);
        unsigned seq = 0;
        for (int proto = 0; proto < 6; ++proto) {
            CHECK(hipMemsetAsync(act[n], 0, act_elems * 4, st));
            auto once = [&]() {
                a.seq = ++seq;
                if (proto == P_LAUNCH) {
                    for (int l = 0; l < n; ++l) hipLaunchKernelGGL(probe_kernel<P_LAUNCH>, dim3(NT), dim3(256), 0, st, a, l);
                } else if (proto == P_WBL2) {
                    hipLaunchKernelGGL(probe_kernel<P_WBL2>, dim3(NT), dim3(256), 0, st, a, 0);
                } else if (proto == P_SC1) {
                    hipLaunchKernelGGL(probe_kernel<P_SC1>, dim3(NT), dim3(256), 0, st, a, 0);
                } else if (proto == P_SC1_PF) {
                    hipLaunchKernelGGL(probe_kernel<P_SC1_PF>, dim3(NT), dim3(256), 0, st, a, 0);
                } else if (proto == P_XCD) {
                    hipLaunchKernelGGL(probe_kernel<P_XCD>, dim3(NT), dim3(256), 0, st, a, 0);
                } else {
                    hipLaunchKernelGGL(probe_kernel<P_XCD_PF>, dim3(NT), dim3(256), 0, st, a, 0);
                }
            };
            for (int i = 0; i < 20; ++i) once();
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) once();
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(got.data(), act[n], act_elems * 4, hipMemcpyDeviceToHost));
            if (proto == 0) ref = got;
            size_t bad = 0;
            for (size_t i = 0; i < act_elems; ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
            printf("run of %d layers  %-9s %7.2f us per run  %6.2f us per layer   mismatching words vs launches: %zu", n, names[proto],
                   ms * 1e3 / reps, ms * 1e3 / reps / n, bad);
            if (proto >= P_XCD) {
                unsigned hx[NT];
                CHECK(hipMemcpyFromSymbol(hx, HIP_SYMBOL(g_xcc), sizeof(hx)));
                int split = 0;
                for (int v = 0; v < VIDEOS; ++v)
                    for (int t = 1; t < TILES_PER_VIDEO; ++t) split += hx[v * TILES_PER_VIDEO + t] != hx[v * TILES_PER_VIDEO];
                printf("   (last run: %d of %d tiles NOT on their video's XCD; video 0..7 on XCC %u %u %u %u %u %u %u %u)", split, NT, hx[0], hx[32], hx[64], hx[96], hx[128], hx[160], hx[192], hx[224]);
            }
            printf("\n");
        }
    }
    return 0;
}
