// MFMA implicit-GEMM kernels for gfx950: the dense contractions of the captioning hot path.
//
//   igemm_nt : Y[m][n] = epi( sum_k A(m,k) W[n][k] )   conv fwd / conv dgrad / fc fwd / fc dgrad / vocab
//   igemm_tn : dW[n][k] += sum_m dY[m][n] A(m,k)       conv + fc weight gradients (split over m)
//
// A(m,k) is gathered on the fly from an NHWC tensor (im2col never materialised).  Both kernels
// are templated on the storage type: bf16 -> v_mfma_f32_16x16x32_bf16, f32 -> the exact
// v_mfma_f32_16x16x4_f32 (reference precision).  64-lane waves, 4 waves per workgroup in a 2x2
// arrangement, register-staged double-buffered LDS tiles with 16-byte global loads; the output
// tile goes back through LDS so that every global store is a full 16-byte chunk of a row.
#include <stdlib.h>
#include <atomic>

#include <cxxabi.h>

#include "common.h"
#include "bn_merge.h"

// ------------------------------------------------------------------ kernel probe (capmi_kernel_probe_begin / _end, capmi.h)
// Which kernel would this call launch?  Between begin and end every launch site of this file RECORDS its kernel (the first
// one of a call: the GEMM itself, not the slab reduce behind it) instead of launching it, so the answer comes from the
// dispatch code itself -- bench.py / profiling.py label a launch with the exact symbol rocprofv3 prints for it, and no Python
// mirror of nt_cfg() / launch_glds() / launch_tn() can drift from them (round-3 review: profiling._nt_tile did).
struct KernelProbe {
    const void* fn;
    unsigned grid, block;
    int launches;
};
static thread_local KernelProbe* t_probe = nullptr;
static thread_local KernelProbe t_probe_store;
static inline void probe_note(const void* fn, dim3 grid, dim3 block) {
    if (t_probe->launches++ == 0) {
        t_probe->fn = fn;
        t_probe->grid = grid.x * grid.y * grid.z;
        t_probe->block = block.x * block.y * block.z;
    }
}
#define CAPMI_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                              \
    do {                                                                                                    \
        if (t_probe) probe_note(reinterpret_cast<const void*>(kernel), dim3(grid), dim3(block));            \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                           \
    } while (0)
#undef CAPMI_LAUNCH_CHECK
#define CAPMI_LAUNCH_CHECK(name)                                                           \
    do {                                                                                   \
        if (!t_probe) {                                                                    \
            hipError_t e__ = hipGetLastError();                                            \
            if (e__ != hipSuccess) {                                                       \
                capmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
                return 2;                                                                  \
            }                                                                              \
        }                                                                                  \
    } while (0)
extern "C" int capmi_kernel_probe_begin(void) {
    t_probe_store = KernelProbe{nullptr, 0u, 0u, 0};
    t_probe = &t_probe_store;
    return 0;
}
extern "C" int capmi_kernel_probe_end(char* symbol, int symbol_len, int* grid, int* block, int* launches) {
    KernelProbe* p = t_probe;
    t_probe = nullptr;
    CAPMI_CHECK(p, "capmi_kernel_probe_end: no probe is open on this thread");
    if (grid) *grid = (int)p->grid;
    if (block) *block = (int)p->block;
    if (launches) *launches = p->launches;
    if (symbol && symbol_len > 0) {
        symbol[0] = 0;
        if (p->fn) {
            const char* mangled = hipKernelNameRefByPtr(p->fn, nullptr);
            (void)hipGetLastError();
            if (!mangled) return 0;          // (a box without a HIP device: the selection ran, the runtime has no code object to name)
            int status = 0;
            char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);        // what rocprofv3 prints (mangled where this fails too)
            snprintf(symbol, (size_t)symbol_len, "%s", (status == 0 && dem) ? dem : mangled);
            free(dem);
        }
    }
    return 0;
}

// floor(n / d) as (n * ceil(2^40 / d)) >> 40, exact whenever n * d < 2^40 (checked by the launchers):
// one 64-bit multiply instead of a ~40-instruction integer division.
struct FastDiv {
    unsigned long long mul;
    int d;
};
static FastDiv fast_div(int d) {
    FastDiv f;
    f.d = d;
    f.mul = ((1ull << 40) + (unsigned long long)d - 1) / (unsigned long long)d;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> 40); }

struct IGemmArgs {
    const void* x;
    const void* w;
    void* y;
    const float* bias;
    const float* bn_mean;            // inference batch norm in the epilogue (capmi_igemm_nt_bn): acc -> bn_a * (acc - bn_mean) + bias
    const float* bn_a;
    // batch norm + activation of the INPUT operand, applied in the A-operand path (capmi_igemm_nt_bnact): x holds the RAW
    // output of the producing convolution, the kernel multiplies act(in_a * (x - in_mean) + in_off) -- bn_apply's formula
    const float* in_mean;
    const float* in_a;
    const float* in_off;
    int in_act;
    const void* addend;
    const void* ysaved;
    float* stats;
    int M, N, K;
    int ldw, ldy, ld_addend, ld_saved;
    capmi_conv_geom g;
    FastDiv fd_hw, fd_w;             // division by Ho*Wo and by Wo
    int act, dact, out_f32;
    int ksplit, kper;                // split-K over workgroups (capmi_igemm_nt_splitk): split s multiplies k in [s * kper, (s + 1) * kper) into f32 slab s
    // fused batch-norm backward reduction (data-gradient launches): for each of `nred` layers that take
    // this launch's OUTPUT as their dy, per-workgroup column sums of dz and dz*(x-mean)*invstd
    int nred;
    const float* stat_shift;         // EPI 8: per-column shift s (the layer's batch mean of the previous step): the sums are of (v - s) and (v - s)^2
    int stat_sums;                   // EPI 8 (capmi_igemm_nt_stat): stats = the layer's accumulator rows [4][2N] of (sum v, sum v^2), added with f32 atomics
    int red_mode;                    // 0: the RED template path (capmi_igemm_nt_bnred: per-tile parts, up to two targets); 1 / 2: EPI 7 (capmi_igemm_nt_bnsum), ONE
                                     // target, its sums as f32 atomics into the four accumulator rows rws[0][4][2N] (1) or as per-tile parts rws[0][tile][2][N] (2)
    const void* rx[2];               // the layer's conv output [rows][N], same row indexing as y
    const float* rmean[2];
    const float* rinv[2];
    float* rws[2];                   // [row block][2][N] partial sums
    // batch-norm finalize by the last-arriving workgroup (capmi_igemm_nt_bnfin): fin_cnt != NULL switches it on
    unsigned* fin_cnt;               // [tiles_n][fin_groups] group arrival counters, then [tiles_n] final counters (zero between launches)
    float* fin_merged;               // [fin_groups][N][2] merged parts (two-level form only)
    const float* fin_scale;
    float* fin_run_mean;
    float* fin_run_var;
    float* fin_mean;
    float* fin_invstd;
    float* fin_a;
    float fin_momentum, fin_eps;
    int fin_update, fin_k, fin_groups;     // fin_k = parts per group (0: one level, the parts are finalized directly)
#ifdef CAPMI_STAMPS
    unsigned long long* stamps;      // diagnostic build only: [workgroup][8] s_memtime stamps
#endif
};
#ifdef CAPMI_STAMPS
static unsigned long long* g_stamp_buffer = nullptr;
extern "C" void capmi_debug_set_stamp_buffer(void* p) { g_stamp_buffer = (unsigned long long*)p; }
#define STAMP(slot)                                                                                   \
    do {                                                                                              \
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

// ------------------------------------------------------------------ MFMA wrappers
template <typename T> struct Frag;
template <> struct Frag<bf16> {
    bf16x8 v;
    __device__ __forceinline__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
};
template <> struct Frag<float> {
    float v[8];
    __device__ __forceinline__ void load(const float* p) {
        f32x4 a = *reinterpret_cast<const f32x4*>(p);
        f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
};
// One 16x16 output tile, 32 reduction elements.  Lane (g = lane>>4, i = lane&15) holds the 8
// reduction elements 8g..8g+7 of row/col i.  For f32 the hardware's k-slot g of step kk is fed
// element 8g+kk of both operands: a permutation of the reduction index, which a dot product
// does not see.
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[kk], b.v[kk], acc, 0, 0, 0);
}

// ------------------------------------------------------------------ im2col addressing
// A row m = (b, ho, wo) is reduced once to (hb, wb, base): hb/wb = top-left input coordinate of
// its window, base = element offset of that (possibly out-of-range) pixel.  A k position is
// tracked incrementally as (c, r, q): channel inside the tap and the tap coordinates -- no
// division in the main loop.
struct RowPos {
    int hb, wb;
    int64_t base;     // ((b*Hi + hb)*Wi + wb) * ldx   (up == 1 only)
    int64_t pix;      // b*Hi*Wi                        (up  > 1)
    bool ok;
};
__device__ __forceinline__ RowPos row_pos(int m, int M, const capmi_conv_geom& g, const FastDiv& dhw, const FastDiv& dw) {
    RowPos r;
    r.ok = m < M;
    int b = fdiv(m, dhw);
    int rem = m - b * dhw.d;
    int ho = fdiv(rem, dw);
    int wo = rem - ho * dw.d;
    r.hb = ho * g.sd - g.pad;
    r.wb = wo * g.sd - g.pad;
    r.pix = (int64_t)b * g.Hi * g.Wi;
    r.base = (r.pix + (int64_t)r.hb * g.Wi + r.wb) * g.ldx;
    return r;
}
struct KPos {
    int k, c, r, q;
};
__device__ __forceinline__ KPos k_pos(int k, const capmi_conv_geom& g) {
    KPos p;
    p.k = k;
    int tap = k / g.Cin;
    p.c = k - tap * g.Cin;
    p.r = tap / g.kw;
    p.q = tap - p.r * g.kw;
    return p;
}
__device__ __forceinline__ void k_advance(KPos& p, int step, const capmi_conv_geom& g) {
    p.k += step;
    p.c += step;
    while (p.c >= g.Cin) {
        p.c -= g.Cin;
        if (++p.q == g.kw) { p.q = 0; ++p.r; }
    }
}
// element offset of A(row, k-chunk) inside x, or -1 when the tap is padding / out of range
__device__ __forceinline__ int64_t a_offset(const RowPos& rp, const KPos& kp, int K, const capmi_conv_geom& g) {
    if (!rp.ok || kp.k >= K) return -1;
    int hn = rp.hb + kp.r, wn = rp.wb + kp.q;
    if (hn < 0 || wn < 0) return -1;
    if (g.up == 1) {
        if (hn >= g.Hi || wn >= g.Wi) return -1;
        return rp.base + (int64_t)(kp.r * g.Wi + kp.q) * g.ldx + kp.c;
    }
    const int mask = g.up - 1;                      // up is a power of two (checked by the launcher)
    if ((hn & mask) || (wn & mask)) return -1;
    const int sh = 31 - __builtin_clz(g.up);
    hn >>= sh;
    wn >>= sh;
    if (hn >= g.Hi || wn >= g.Wi) return -1;
    return (rp.pix + (int64_t)hn * g.Wi + wn) * g.ldx + kp.c;
}

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&o)[8]) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&o)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = o;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// Workgroups are dealt round-robin over the 8 XCDs (each with a private L2).  Remap the launch
// index so that every XCD gets a CONTIGUOUS run of tiles: tiles that share input rows / write
// the two halves of the same output rows then meet in one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// TN consecutive output values of one row -> one vector store (bf16: 4/8 bytes, f32: 8/16 bytes)
template <typename O, int TN> __device__ __forceinline__ void store_run(O* p, const float (&v)[TN]) {
    if constexpr (TN == 8) {
        float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
        if constexpr (sizeof(O) == 2) {
            bf16x8 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3], (bf16)v[4], (bf16)v[5], (bf16)v[6], (bf16)v[7]};
            *reinterpret_cast<bf16x8*>(p) = o;
        } else {
            *reinterpret_cast<f32x4*>(p) = f32x4{lo[0], lo[1], lo[2], lo[3]};
            *reinterpret_cast<f32x4*>(p + 4) = f32x4{hi[0], hi[1], hi[2], hi[3]};
        }
    } else if constexpr (sizeof(O) == 2) {
        if constexpr (TN == 4) {
            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *reinterpret_cast<bf16x4*>(p) = o;
        } else {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            bf16x2 o = {(bf16)v[0], (bf16)v[1]};
            *reinterpret_cast<bf16x2*>(p) = o;
        }
    } else {
        if constexpr (TN == 4) *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        else {
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
        }
    }
}
template <typename O, int TN> __device__ __forceinline__ void load_run(const O* p, float (&v)[TN]) {
    if constexpr (TN == 8) {
        if constexpr (sizeof(O) == 2) {
            bf16x8 o = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)o[e];
        } else {
            f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
        }
    } else if constexpr (sizeof(O) == 2) {
        if constexpr (TN == 4) {
            bf16x4 o = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)o[e];
        } else {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            bf16x2 o = *reinterpret_cast<const bf16x2*>(p);
            v[0] = (float)o[0]; v[1] = (float)o[1];
        }
    } else {
        if constexpr (TN == 4) {
            f32x4 o = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[e];
        } else {
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            f32x2 o = *reinterpret_cast<const f32x2*>(p);
            v[0] = o[0]; v[1] = o[1];
        }
    }
}
template <int TN> __device__ __forceinline__ void act_run(float (&v)[TN], int act) {
    switch (act) {
        case CAPMI_ACT_RELU:
#pragma unroll
            for (int e = 0; e < TN; ++e) v[e] = fmaxf(v[e], 0.f);
            break;
        case CAPMI_ACT_RELU6:
#pragma unroll
            for (int e = 0; e < TN; ++e) v[e] = fminf(fmaxf(v[e], 0.f), 6.f);
            break;
        case CAPMI_ACT_TANH:
#pragma unroll
            for (int e = 0; e < TN; ++e) v[e] = tanhf_(v[e]);
            break;
        case CAPMI_ACT_SIGMOID:
#pragma unroll
            for (int e = 0; e < TN; ++e) v[e] = sigmoidf_(v[e]);
            break;
        default: break;
    }
}
// a mask byte carried in one element of an operand vector (the bit-mask form shares the registers of the saved-output form)
template <typename T> __device__ __forceinline__ T bits_to_elem(unsigned b);
template <> __device__ __forceinline__ bf16 bits_to_elem<bf16>(unsigned b) { return __builtin_bit_cast(bf16, (unsigned short)b); }
template <> __device__ __forceinline__ float bits_to_elem<float>(unsigned b) { return __builtin_bit_cast(float, b); }
template <typename T> __device__ __forceinline__ unsigned elem_to_bits(T e);
template <> __device__ __forceinline__ unsigned elem_to_bits<bf16>(bf16 e) { return (unsigned)__builtin_bit_cast(unsigned short, e); }
template <> __device__ __forceinline__ unsigned elem_to_bits<float>(float e) { return __builtin_bit_cast(unsigned, e); }
template <int TN> __device__ __forceinline__ void dact_run(float (&v)[TN], const float (&y)[TN], int dact) {
    switch (dact) {
        case CAPMI_ACT_RELU:
#pragma unroll
            for (int e = 0; e < TN; ++e) v[e] = y[e] > 0.f ? v[e] : 0.f;
            break;
        case CAPMI_ACT_RELU6:
#pragma unroll
            for (int e = 0; e < TN; ++e) v[e] = (y[e] > 0.f && y[e] < 6.f) ? v[e] : 0.f;
            break;
        case CAPMI_ACT_TANH:
#pragma unroll
            for (int e = 0; e < TN; ++e) v[e] *= 1.f - y[e] * y[e];
            break;
        case CAPMI_ACT_SIGMOID:
#pragma unroll
            for (int e = 0; e < TN; ++e) v[e] *= y[e] * (1.f - y[e]);
            break;
        default: break;
    }
}

// ------------------------------------------------------------------ batch norm + activation on 8 bf16 operand values
// The formula and rounding points of bn_apply_kernel (bn_ops.hip): act(a * (x - mean) + offset) in f32, rounded to bf16
// once -- a consumer that applies it to the RAW conv output multiplies the bits bn_apply would have stored.
// ca / mu / of: this lane's 8 consecutive channels.
constexpr int INBN_KMAX = 512;          // channels of the input tensor the coefficient table in LDS holds
__device__ __forceinline__ bf16x8 bn_act8(bf16x8 v, const f32x4& ca0, const f32x4& ca1, const f32x4& mu0, const f32x4& mu1,
                                          const f32x4& of0, const f32x4& of1, int act) {
    bf16x8 o;
    const float hi = act == CAPMI_ACT_RELU6 ? 6.f : __builtin_inff();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = e < 4 ? ca0[e & 3] : ca1[e & 3], m = e < 4 ? mu0[e & 3] : mu1[e & 3], b = e < 4 ? of0[e & 3] : of1[e & 3];
        const float f = a * ((float)v[e] - m) + b;
        o[e] = (bf16)__builtin_amdgcn_fmed3f(f, 0.f, hi);       // relu / relu6 as ONE instruction (median of f, 0, upper bound)
    }
    return o;
}
// the coefficient table [3][INBN_KMAX] f32 (a | mean | offset) of an input tensor with C channels: plain loads + LDS stores
// by the first C / 4 threads; the caller orders them before the first read (lgkmcnt(0) + a workgroup barrier)
template <int KC>
__device__ __forceinline__ void inbn_load_table(float* tab, const IGemmArgs& a, int C, int tid) {
    if (tid * 4 < C) {
        *reinterpret_cast<f32x4*>(tab + tid * 4) = *reinterpret_cast<const f32x4*>(a.in_a + tid * 4);
        *reinterpret_cast<f32x4*>(tab + KC + tid * 4) = *reinterpret_cast<const f32x4*>(a.in_mean + tid * 4);
        *reinterpret_cast<f32x4*>(tab + 2 * KC + tid * 4) = *reinterpret_cast<const f32x4*>(a.in_off + tid * 4);
    }
}
// this lane's 8 channels c0 .. c0 + 7 of the table: (a | mean | offset) as six 16-byte LDS reads
template <int KC>
__device__ __forceinline__ void inbn_coef(const float* tab, int c0, f32x4 (&cf)[6]) {
    const float* t = tab + c0;
    cf[0] = *reinterpret_cast<const f32x4*>(t);
    cf[1] = *reinterpret_cast<const f32x4*>(t + 4);
    cf[2] = *reinterpret_cast<const f32x4*>(t + KC);
    cf[3] = *reinterpret_cast<const f32x4*>(t + KC + 4);
    cf[4] = *reinterpret_cast<const f32x4*>(t + 2 * KC);
    cf[5] = *reinterpret_cast<const f32x4*>(t + 2 * KC + 4);
}

// ------------------------------------------------------------------ batch-norm finalize by the last-arriving workgroup
#ifndef CAPMI_FIN
#define CAPMI_FIN 0         // the tail is compiled OUT by default (lesson 48: its mere presence in the shared epilogue costs every NT kernel ~0.04 ms per step); -DCAPMI_FIN=1 builds it in, capmi_igemm_nt_bnfin is the two calls otherwise
#endif
// capmi_igemm_nt_bnfin: the statistics merge + finalize of capmi_bn_finalize WITHOUT its launch (a dependent ~10 us kernel
// behind every convolution of the forward chain).  Every workgroup has stored its part write-through; it drains its stores,
// adds to the arrival counter of its group of fin_k row blocks (per column tile) and leaves -- unless it is the LAST of the
// group: then it merges the group's parts (merge_parts, the fold order of bn_merge_kernel), stores the merged part
// write-through and adds to the column tile's final counter; the last GROUP to arrive finalizes the tile's channels
// (bn_finalize_body).  With <= 64 parts there is one level: the last workgroup of the column tile finalizes from the parts.
// Nobody waits for anybody (the launch drains in any arrival order); counters are reset by their last arriver; the
// arithmetic is capmi_bn_finalize's, bit for bit.  Runs on the 256 threads of the epilogue; sred: >= 6.2 KB of free LDS.
template <int BM, int BN>
__device__ __forceinline__ void nt_fin_tail(const IGemmArgs& a, int m0, int n0, float* sred) {
    const int tid = threadIdx.x;
    unsigned* flag = reinterpret_cast<unsigned*>(sred);
    double (*red)[64] = reinterpret_cast<double (*)[64]>(sred + 16);
    const int nparts = (a.M + BM - 1) / BM, p = m0 / BM, ntile = n0 / BN;
    const int groups = a.fin_k > 0 ? a.fin_groups : 1;
    const int grp = a.fin_k > 0 ? p / a.fin_k : 0;
    const int gsize = a.fin_k > 0 ? min(a.fin_k, nparts - grp * a.fin_k) : nparts;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores have left the chip's caches
    __syncthreads();
    if (tid == 0) {
        unsigned* c = a.fin_cnt + ntile * groups + grp;
        const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)gsize - 1u;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
        flag[0] = last ? 1u : 0u;
    }
    __syncthreads();
    if (!flag[0]) return;
    if (a.fin_k > 0) {
        const int p0 = grp * a.fin_k, p1 = min(nparts, p0 + a.fin_k);
#pragma unroll
        for (int cb = 0; cb < BN / 64; ++cb) {
            const int c = n0 + cb * 64 + (tid & 63);
            double mean, m2;
            merge_parts<true>(a.stats, BM, a.M, a.N, c, p0, p1, red, &mean, &m2);
            if ((tid >> 6) == 0 && c < a.N) {
                const unsigned long long v = (unsigned long long)__builtin_bit_cast(unsigned, (float)mean) |
                                             ((unsigned long long)__builtin_bit_cast(unsigned, (float)m2) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.fin_merged + ((int64_t)grp * a.N + c) * 2), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            unsigned* c = a.fin_cnt + ((a.N + BN - 1) / BN) * groups + ntile;
            const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old == (unsigned)groups - 1u;
            if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag[0] = last ? 1u : 0u;
        }
        __syncthreads();
        if (!flag[0]) return;
    }
    const float* src = a.fin_k > 0 ? a.fin_merged : a.stats;
    const int rows = a.fin_k > 0 ? BM * a.fin_k : BM;
#pragma unroll
    for (int cb = 0; cb < BN / 64; ++cb) {
        if (n0 + cb * 64 >= a.N) break;
        bn_finalize_body<true>(src, rows, a.M, a.N, (n0 + cb * 64) / 64, a.fin_scale, a.fin_run_mean, a.fin_run_var, a.fin_momentum, a.fin_eps,
                               a.fin_mean, a.fin_invstd, a.fin_a, a.fin_update, red);
    }
}

// ------------------------------------------------------------------ mask bytes of the epilogue, fetched in front of the main loop
// EPI 6 (data gradient with its ReLU mask as bits): the bytes a lane's rows need -- one per row, TM x 4 registers -- are loaded
// BEFORE the k loop, so the epilogue of a mask-only data gradient issues no load at all (its only memory round trip used to be
// the mask; a timing run without the mask bounds that at 0.15 ms per step, lesson 53).  Same row / column mapping as nt_epilogue.
template <typename T, int BM, int BN, int WMW, bool SCATTER = true>
__device__ __forceinline__ void nt_prefetch_mask(const IGemmArgs& a, int m0, int n0, unsigned (&pm)[BM / WMW / 16][4]) {
    constexpr int WNW = 4 / WMW, RW = BM / WMW, WN = BN / WNW, TM = RW / 16, TN = WN / 16;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNW, wn = wave % WNW;
    const int fr = lane & 15, fg = lane >> 4;
    const int wrow0 = m0 + wm * RW;
    const int wcnt = min(RW, a.M - wrow0);
    const int col0 = n0 + wn * WN + TN * fr;
    const uint8_t* bits = reinterpret_cast<const uint8_t*>(a.ysaved);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = i * 16 + fg * 4 + r;
            int64_t row = wrow0 + rl;
            const bool valid = rl < wcnt && col0 < a.N;
            if (SCATTER && a.g.os > 1 && valid) {
                const int mm = (int)row, bb = fdiv(mm, a.fd_hw), rem = mm - bb * a.fd_hw.d;
                const int ii = fdiv(rem, a.fd_w), jj = rem - ii * a.fd_w.d;
                row = ((int64_t)bb * a.g.Hof + ii * a.g.os + a.g.oh0) * a.g.Wof + jj * a.g.os + a.g.ow0;
            }
            pm[i][r] = valid ? (unsigned)bits[(row * a.ld_saved + col0) >> 3] : 0u;
        }
}

// ------------------------------------------------------------------ shared epilogue of the tiled NT kernels
// DENSE: the host guarantees N and every row pitch the epilogue touches are multiples of 8 (nt_dense): every lane stores whole
// runs, and the element-wise store path -- HALF of each kernel's instructions, all of them unrolled copies that only ragged
// shapes ever execute -- is compiled out.  The LDS-DMA kernel families are DENSE-only (ragged launches take the register-staged
// kernel): the same step runs 0.10 ms faster for it, lesson 54 -- an epilogue of 9 000 - 17 000 instructions does not fit the
// instruction cache two CUs share next to the other lane's kernels.
// EPI = 1 / 4 (nt_conv_class, chosen on the host): the epilogue of a training convolution (1: plain stores + the fused
// statistics) or of its data gradient (4: addend, relu / relu6 derivative mask, output scatter; no statistics) -- no bias, no
// inference batch norm, no output activation, storage-type output.  The other paths (five activation forms per stored run, f32
// slabs) are compiled out of those instantiations: code size again.
// EPI = 2 (nt_fc_class): the decoder's fully connected layers and their data gradients -- bias and addend as in the general form,
// activation / activation derivative tanh or none, storage-type output.
// EPI = 3 (nt_inf_class): a convolution of the inference graph (capmi_igemm_nt_bn) -- batch norm on the accumulator, residual
// addend, relu / relu6 or nothing, storage-type output; no statistics, no derivative mask.
// EPI = 6: the data-gradient form (4) whose ReLU mask comes as BITS (dact | CAPMI_DACT_BITMASK: a byte per 8 channels that
// capmi_bn_apply_mask wrote in the forward pass) instead of the saved activation itself -- 1/16 of the mask's bytes.
// EPI = 7 (capmi_igemm_nt_bnsum): EPI 6 + the batch-norm BACKWARD sums of the layer whose output gradient this launch completes
// (sum dz and sum dz * (x - mean) * invstd per column, x = that layer's conv output): the epilogue loads the x run of every row
// next to the addend, adds the masked, bf16-rounded gradient -- the very values bn_bwd_apply will read back -- into 2 x TN
// per-lane sums, folds them over the wave (row4_sum), over the four waves in LDS, and sends 2 x BN f32 atomics per workgroup to
// the layer's accumulator rows (capmi_bn_bwd_reduce_spread's layout; deterministic mode: plain stores of per-tile parts).  The
// streaming reduction of that layer (a launch + two tensor reads on the dependency chain) is gone.  One target, no statistics,
// mask bits always prefetched; instantiated only for the launches that carry it (lesson 54).
// EPI = 8 (capmi_igemm_nt_stat): the forward form (1) whose batch statistics leave the workgroup as SUMS -- sum v and sum v^2 per
// column from the f32 accumulators, folded over the wave and the four waves like EPI 7's, 2 x BN f32 atomics per workgroup into
// the layer's four accumulator rows stats[4][2N] -- instead of one exact (mean, M2) part per row block.  capmi_bn_stat_apply
// forms mean / invstd from the rows in its prologue: the merge + finalize launch behind every convolution of the forward chain
// (41 x ~10 us at cfg 2) is gone.  Not bit-reproducible (atomic order) and one-pass in f32: bf16 engines outside deterministic
// mode only -- the f32 engine and deterministic mode keep the exact parts.
// EPI = 5 (nt_f32_class): a plain product with an f32 output and at most a bias -- the vocabulary projection's logits
// (model_adaAttention_aic.py:25), once per train step and once per decode step.
template <typename T, int BM, int BN, int WMW, bool RED = false, bool DENSE = false, int EPI = 0>
__device__ __forceinline__ void nt_epilogue(const IGemmArgs& a, f32x4 (&acc)[BM / WMW / 16][BN / (4 / WMW) / 16], int m0, int n0, float* sred,
                                            int64_t slab_off = 0,        // f32 elements added to y (split-K: this split's slab)
                                            const unsigned (*pmask)[4] = nullptr) {      // EPI 6: mask bytes fetched in front of the main loop (nt_prefetch_mask)
    constexpr int WNW = 4 / WMW, RW = BM / WMW, WN = BN / WNW, TM = RW / 16, TN = WN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNW, wn = wave % WNW;
    const int fr = lane & 15, fg = lane >> 4;
    // ---- epilogue (registers only).  acc[i][j][r] = output (row wrow0 + 16i + 4fg + r, column col0 + j).
    const int wrow0 = m0 + wm * RW;                            // first row of this wave's sub-tile
    const int wcnt = min(RW, a.M - wrow0);                     // valid rows in it (<= 0: none)
    const int col0 = n0 + wn * WN + TN * fr;                   // this lane's TN consecutive columns
    const bool rows_full = wcnt == RW;
    if ((EPI == 0 || EPI == 3) && a.bn_a) {       // (EPI 1: neither; EPI 2: bias only) the formula of bn_apply (mean subtracted before scaling), on the f32 accumulator
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bool ok = col0 + j < a.N;
            const float ca = ok ? a.bn_a[col0 + j] : 0.f, mu = ok ? a.bn_mean[col0 + j] : 0.f, off = ok ? a.bias[col0 + j] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = ca * (acc[i][j][r] - mu) + off;
        }
    } else if ((EPI == 0 || EPI == 2 || EPI == 5) && a.bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bias = col0 + j < a.N ? a.bias[col0 + j] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += bias;
        }
    }
    const T* addend = (const T*)a.addend;
    const T* ysaved = (const T*)a.ysaved;
    const bool vec_ok = DENSE || ((a.ldy % TN == 0) && (!addend || a.ld_addend % TN == 0) && (!a.dact || a.ld_saved % TN == 0) && col0 + TN <= a.N &&
                                  (!a.nred || a.N % TN == 0));
    typedef T __attribute__((ext_vector_type(TN))) RunT;      // TN consecutive values of one row, as loaded
    // Fused batch-norm backward reduction (capmi_igemm_nt_bnred): this launch's output is the dy of up to
    // two BN layers.  Target 0's sums are taken while the rows are stored; target 1 (rare) in a later pass.
    const T* rx0 = (const T*)a.rx[0];
    float mu0[TN], s1[TN], s2[TN];
    if constexpr (RED || EPI == 7) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bool ok = col0 + j < a.N;
            mu0[j] = ok ? a.rmean[0][col0 + j] : 0.f;
            s1[j] = 0.f; s2[j] = 0.f;
        }
    }
    // EPI 7: the batch-norm operand (the layer's conv output) of EVERY row of the lane, requested up front -- the fragment
    // registers of the main loop are dead by now -- so the epilogue pays one memory latency for them, not one per row batch
    // (measured: +10.7 us per launch with the loads inside the batches, 2 rows at a time)
    // The sums are taken in a SECOND pass over the accumulators (which then hold the stored values), behind all the stores:
    // nothing in the store pass waits for these loads.  With an addend (whose own loads come first) they are issued row block
    // by row block behind that block's stores, as its addend registers fall free.
    RunT pxa[EPI == 7 ? TM : 1][4];
    auto load_px = [&](int i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = i * 16 + fg * 4 + r;
            if (rows_full || rl < wcnt) pxa[i][r] = *reinterpret_cast<const RunT*>(rx0 + (int64_t)(wrow0 + rl) * a.N + col0);
        }
    };
    if constexpr (EPI == 7) {
        if (col0 < a.N && !a.addend) {
#pragma unroll
            for (int i = 0; i < TM; ++i) load_px(i);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // EPI 8: the shift of column n0 + tid, requested before the stores and consumed at the very end of the epilogue (loads and
    // stores share vmcnt: by then the stores have long been acknowledged -- a use right behind them waited for every one)
    float sh8 = 0.f;
    if constexpr (EPI == 8) {
        if (tid < BN && n0 + tid < a.N) sh8 = a.stat_shift[n0 + tid];
    }
    // output row (scattered for a parity class of a strided data gradient) of the lane's row r of row block i
    auto out_row = [&](int i, int r, bool& valid) -> int64_t {
        const int rl = i * 16 + fg * 4 + r;
        valid = rows_full || rl < wcnt;
        int64_t row = wrow0 + rl;
        if (EPI != 1 && EPI != 8 && EPI != 7 && a.g.os > 1 && valid) {  // scatter: GEMM row (b,i,j) -> pixel (b, i*os+oh0, j*os+ow0) of [B,Hof,Wof]
            const int mm = (int)row, bb = fdiv(mm, a.fd_hw), rem = mm - bb * a.fd_hw.d;
            const int ii = fdiv(rem, a.fd_w), jj = rem - ii * a.fd_w.d;
            row = ((int64_t)bb * a.g.Hof + ii * a.g.os + a.g.oh0) * a.g.Wof + jj * a.g.os + a.g.ow0;
        }
        return row;
    };
    // EPI 3 / 6 (inference convolution with its residual; data gradient with an addend and the mask already in registers): the
    // addend runs of ALL rows up front as well -- these epilogues have no other load, and the batches of four rows cost one
    // exposed memory latency each (TM of them per workgroup)
    constexpr bool PA_ALL = DENSE && (EPI == 3 || EPI == 6 || EPI == 7);
    RunT paa[PA_ALL ? TM : 1][4];
    if constexpr (PA_ALL) {
        if (addend && col0 < a.N) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bool ok;
                    const int64_t row = out_row(i, r, ok);
                    if (ok) paa[i][r] = *reinterpret_cast<const RunT*>(addend + row * a.ld_addend + col0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (col0 < a.N) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int64_t rows[4];
            bool valid[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) rows[r] = out_row(i, r, valid[r]);
            if (DENSE || vec_ok) {
                // all loads of the four rows first (the output may alias the addend, so the compiler cannot
                // hoist a row's loads over the previous row's store by itself: one memory latency, not four)
                constexpr int RB = RED ? 2 : 4;      // rows per load batch (register budget)
#pragma unroll
                for (int h = 0; h < 4; h += RB) {
                RunT pa[4], py[4], px[4];
#pragma unroll
                for (int r = h; r < h + RB; ++r) {
                    if (!PA_ALL && EPI != 1 && EPI != 8 && EPI != 5 && addend && valid[r]) pa[r] = *reinterpret_cast<const RunT*>(addend + rows[r] * a.ld_addend + col0);
                    if constexpr (EPI == 6) {
                        if (!pmask && valid[r]) py[r][0] = bits_to_elem<T>(reinterpret_cast<const uint8_t*>(a.ysaved)[(rows[r] * a.ld_saved + col0) >> 3]);
                    } else if (EPI != 7 && EPI != 1 && EPI != 8 && EPI != 3 && EPI != 5 && a.dact && valid[r]) py[r] = *reinterpret_cast<const RunT*>(ysaved + rows[r] * a.ld_saved + col0);
                    if (RED && valid[r]) px[r] = *reinterpret_cast<const RunT*>(rx0 + rows[r] * a.N + col0);
                }
#pragma unroll
                for (int r = h; r < h + RB; ++r) {
                    if (!valid[r]) continue;
                    float v[TN], t[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
                    if (EPI != 1 && EPI != 8 && EPI != 5 && addend) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) v[j] += PA_ALL ? (float)paa[i][r][j] : (float)pa[r][j];
                    }
                    if constexpr (EPI == 0) act_run<TN>(v, a.act);
                    if constexpr (EPI == 2) {
                        if (a.act) {
#pragma unroll
                            for (int j = 0; j < TN; ++j) v[j] = tanhf_(v[j]);
                        }
                    }
                    if constexpr (EPI == 3) {
                        if (a.act) {
                            const float hi = a.act == CAPMI_ACT_RELU6 ? 6.f : __builtin_inff();
#pragma unroll
                            for (int j = 0; j < TN; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], 0.f, hi);
                        }
                    }
                    if constexpr (EPI == 6) {
                        const unsigned m = (pmask ? pmask[i][r] : elem_to_bits<T>(py[r][0])) >> (col0 & 7);
#pragma unroll
                        for (int j = 0; j < TN; ++j) v[j] = ((m >> j) & 1u) ? v[j] : 0.f;
                    } else if constexpr (EPI == 7) {
                        const unsigned m = pmask[i][r] >> (col0 & 7);
#pragma unroll
                        for (int j = 0; j < TN; ++j) v[j] = ((m >> j) & 1u) ? v[j] : 0.f;
                    } else if (EPI != 1 && EPI != 8 && EPI != 3 && EPI != 5 && a.dact) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) t[j] = (float)py[r][j];
                        if constexpr (EPI == 0) dact_run<TN>(v, t, a.dact);
                        else if constexpr (EPI == 2) {
#pragma unroll
                            for (int j = 0; j < TN; ++j) v[j] *= 1.f - t[j] * t[j];
                        } else {
                            const float hi = a.dact == CAPMI_ACT_RELU6 ? 6.f : __builtin_inff();      // relu / relu6 (nt_conv_class)
#pragma unroll
                            for (int j = 0; j < TN; ++j) v[j] = (t[j] > 0.f && t[j] < hi) ? v[j] : 0.f;
                        }
                    }
                    if (EPI == 5 || (EPI == 0 && a.out_f32)) store_run<float, TN>((float*)a.y + slab_off + rows[r] * a.ldy + col0, v);
                    else store_run<T, TN>((T*)a.y + rows[r] * a.ldy + col0, v);
                    if constexpr (RED) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float dz = (a.out_f32 || sizeof(T) == 4) ? v[j] : to_f32(from_f32<T>(v[j]));   // the value the BN apply pass reads
                            acc[i][j][r] = dz;
                            s1[j] += dz;
                            s2[j] += dz * ((float)px[r][j] - mu0[j]);      // * invstd when the part is written
                        }
                    }
                    if constexpr (EPI == 7) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float dz = to_f32(from_f32<T>(v[j]));     // the value the BN apply pass reads back
                            acc[i][j][r] = dz;
                            s1[j] += dz;
                        }
                    }
                }
                }
                if constexpr (EPI == 7) {
                    if (a.addend) load_px(i);
                }
            } else if constexpr (!DENSE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!valid[r]) continue;
                    const int64_t row = rows[r];
                    float v[TN], t[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
                    const int nv = min(TN, a.N - col0);
                    if (addend) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) v[j] += j < nv ? to_f32(addend[row * a.ld_addend + col0 + j]) : 0.f;
                    }
                    act_run<TN>(v, a.act);
                    if (a.dact) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) t[j] = j < nv ? to_f32(ysaved[row * a.ld_saved + col0 + j]) : 0.f;
                        dact_run<TN>(v, t, a.dact);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (j < nv) {
                            if (a.out_f32) ((float*)a.y)[slab_off + row * a.ldy + col0 + j] = v[j];
                            else ((T*)a.y)[row * a.ldy + col0 + j] = from_f32<T>(v[j]);
                        }
                    if constexpr (RED) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float dz = j < nv ? ((a.out_f32 || sizeof(T) == 4) ? v[j] : to_f32(from_f32<T>(v[j]))) : 0.f;
                            acc[i][j][r] = dz;
                            s1[j] += dz;
                            if (j < nv) s2[j] += dz * (to_f32(rx0[row * a.N + col0 + j]) - mu0[j]);
                        }
                    }
                }
            }
        }
    }
    if constexpr (EPI == 8) {
        // Per-tile sums of v and v^2 over the tile's BM rows -- packed f32 (v_pk_add_f32 / v_pk_fma_f32: two rows per instruction,
        // 10 VALU per column where the shifted, per-row-tested form took 24; rows past M hold zeros and add nothing) -- folded over
        // the wave and the WMW waves, and only THEN shifted: with n valid rows, sum (v - s) = t1 - n s and sum (v - s)^2 =
        // t2 - s (t1 + (t1 - n s)).  What the atomics add up over the M / BM tiles are the shifted sums, so with the shift near the
        // column's mean (the previous step's batch mean) the one-pass variance E[d^2] - E[d]^2 loses nothing to cancellation over
        // the tensor; inside a tile (128 rows, a register tree) the unshifted sums are good to ~1e-7 of t2.
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        lds_barrier();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x2 p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x2 lo = {acc[i][j][0], acc[i][j][1]}, hi = {acc[i][j][2], acc[i][j][3]};
                p1 += lo;
                p2 = __builtin_elementwise_fma(lo, lo, p2);
                p1 += hi;
                p2 = __builtin_elementwise_fma(hi, hi, p2);
            }
            const float r1 = row4_sum(p1[0] + p1[1]), r2 = row4_sum(p2[0] + p2[1]);
            if (fg == 0) {
                const int c = wn * WN + TN * fr + j;
                sred[(wm * BN + c) * 2 + 0] = r1;
                sred[(wm * BN + c) * 2 + 1] = r2;
            }
        }
        lds_barrier();
        if (tid < BN && n0 + tid < a.N && m0 < a.M) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) { t1 += sred[(w * BN + tid) * 2]; t2 += sred[(w * BN + tid) * 2 + 1]; }
            const float nrows = (float)min(BM, a.M - m0);
            const float u1 = __builtin_fmaf(-nrows, sh8, t1);               // sum (v - s)
            const float u2 = __builtin_fmaf(-sh8, t1 + u1, t2);             // sum (v - s)^2
            float* row = a.stats + (int64_t)((m0 / BM) & 3) * 2 * a.N + n0 + tid;
            __hip_atomic_fetch_add(row, u1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(row + a.N, u2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (EPI <= 1 && a.stats) {      // (EPI 2 / 3 / 4 never carry statistics)
        // (after the output stores have been issued: the statistics -- two LDS round trips -- then run while the stores
        // drain; in front of them they cost +42 % on the write-bound 64 -> 256 1x1 layer, tools/nt_ablate.hip)
        // Fused batch-norm statistics: exact (mean, M2 = sum (v-mean)^2) per column of this workgroup's
        // BM-row block from the f32 accumulators.  Each wave reduces its RW rows in registers
        // (two passes, no cancellation); the WMW wave results meet in LDS and are merged with Chan's
        // formula; ONE part per workgroup row block is stored (plain stores, one producer per
        // (part, column): deterministic).  bn_finalize (bn_ops.hip) merges the parts in f64.
        // sred: [WMW][BN][2] floats of LDS scratch (the staging tiles are free now)
        // (LDS-only barriers: __syncthreads() is also `s_waitcnt vmcnt(0)`, i.e. it made every workgroup wait for its output
        // tile's stores to be acknowledged before it could start on the statistics)
        lds_barrier();
        const float inv_cnt = wcnt > 0 ? 1.f / (float)wcnt : 0.f;       // one division per lane, not one per column (exact for full 16 / 32-row blocks)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rows_full || i * 16 + fg * 4 + r < wcnt) s1 += acc[i][j][r];
            const float mean = row4_sum(s1) * inv_cnt;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[i][j][r] - mean;
                    if (rows_full || i * 16 + fg * 4 + r < wcnt) m2 += d * d;
                }
            m2 = row4_sum(m2);
            if (fg == 0) {
                const int c = wn * WN + TN * fr + j;
                sred[(wm * BN + c) * 2 + 0] = mean;
                sred[(wm * BN + c) * 2 + 1] = m2;
            }
        }
        lds_barrier();
        if (tid < BN && n0 + tid < a.N && m0 < a.M) {
            float ntot = 0.f, msum = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) {
                const float nw = (float)max(0, min(RW, a.M - (m0 + w * RW)));
                ntot += nw;
                msum += nw * sred[(w * BN + tid) * 2];
            }
            const float mean = msum / ntot;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) {
                const float nw = (float)max(0, min(RW, a.M - (m0 + w * RW)));
                const float d = sred[(w * BN + tid) * 2] - mean;
                m2 += sred[(w * BN + tid) * 2 + 1] + nw * d * d;
            }
            float* w = a.stats + ((int64_t)(m0 / BM) * a.N + n0 + tid) * 2;
            if (CAPMI_FIN && a.fin_cnt) {        // write-through: another workgroup of this launch (the last to arrive) reads it
                const unsigned long long v = (unsigned long long)__builtin_bit_cast(unsigned, mean) | ((unsigned long long)__builtin_bit_cast(unsigned, m2) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(w), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                w[0] = mean;
                w[1] = m2;
            }
        }
        if (CAPMI_FIN && a.fin_cnt) nt_fin_tail<BM, BN>(a, m0, n0, sred);
    }
    if constexpr (EPI == 7) {
        if (col0 < a.N) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!(rows_full || i * 16 + fg * 4 + r < wcnt)) continue;
#pragma unroll
                    for (int j = 0; j < TN; ++j) s2[j] += acc[i][j][r] * ((float)pxa[i][r][j] - mu0[j]);  // * invstd below, once per column
                }
        }
        // (first barrier: every wave is done with whatever lived at the start of the staging area -- the k-group exchange)
        lds_barrier();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float r1 = row4_sum(s1[j]), r2 = row4_sum(s2[j]);
            if (fg == 0) {
                const int c = wn * WN + TN * fr + j;
                sred[(wm * BN + c) * 2 + 0] = r1;
                sred[(wm * BN + c) * 2 + 1] = r2;
            }
        }
        lds_barrier();
        if (tid < BN && n0 + tid < a.N && m0 < a.M) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) { t1 += sred[(w * BN + tid) * 2]; t2 += sred[(w * BN + tid) * 2 + 1]; }
            t2 *= a.rinv[0][n0 + tid];
            if (a.red_mode == 1) {      // four accumulator rows [4][2N]: a quarter of the row blocks adds to any one address
                float* row = a.rws[0] + (int64_t)((m0 / BM) & 3) * 2 * a.N + n0 + tid;
                __hip_atomic_fetch_add(row, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(row + a.N, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {                    // deterministic mode: one part per row block, summed in a fixed order by bn_bwd_reduce_final
                float* w = a.rws[0] + (int64_t)(m0 / BM) * 2 * a.N + n0 + tid;
                w[0] = t1;
                w[a.N] = t2;
            }
        }
    }
    if constexpr (RED)
    for (int q = 0; q < a.nred; ++q) {
        if (q > 0) {        // second target: another pass over the stored values (acc holds them)
            const T* rx = (const T*)a.rx[q];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const bool ok = col0 + j < a.N;
                mu0[j] = ok ? a.rmean[q][col0 + j] : 0.f;
                s1[j] = 0.f; s2[j] = 0.f;
            }
            if (col0 < a.N) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rl = i * 16 + fg * 4 + r;
                        if (!rows_full && rl >= wcnt) continue;
                        int64_t row = wrow0 + rl;
                        if (a.g.os > 1) {
                            const int mm = (int)row, bb = fdiv(mm, a.fd_hw), rem = mm - bb * a.fd_hw.d;
                            const int ii = fdiv(rem, a.fd_w), jj = rem - ii * a.fd_w.d;
                            row = ((int64_t)bb * a.g.Hof + ii * a.g.os + a.g.oh0) * a.g.Wof + jj * a.g.os + a.g.ow0;
                        }
                        float t[TN];
                        if (vec_ok) load_run<T, TN>(rx + row * a.N + col0, t);
                        else {
#pragma unroll
                            for (int j = 0; j < TN; ++j) t[j] = col0 + j < a.N ? to_f32(rx[row * a.N + col0 + j]) : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            s1[j] += acc[i][j][r];
                            s2[j] += acc[i][j][r] * (t[j] - mu0[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float r1 = row4_sum(s1[j]), r2 = row4_sum(s2[j]);
            if (fg == 0) {
                const int c = wn * WN + TN * fr + j;
                sred[(wm * BN + c) * 2 + 0] = r1;
                sred[(wm * BN + c) * 2 + 1] = r2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.N && m0 < a.M) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) { t1 += sred[(w * BN + tid) * 2]; t2 += sred[(w * BN + tid) * 2 + 1]; }
            float* w = a.rws[q] + (int64_t)(m0 / BM) * 2 * a.N + n0 + tid;
            w[0] = t1;
            w[a.N] = t2 * a.rinv[q][n0 + tid];
        }
    }
}

// ------------------------------------------------------------------ NT kernel
// Column permutation: the weight row fed to MFMA tile j, lane fr of a wave is column TN*fr + j of
// the wave's BN/2-column range (the permutation is applied when the W tile is written to LDS, so
// fragment reads keep their conflict-free addresses).  Each lane then owns TN CONSECUTIVE output
// columns of every row it holds: the epilogue stores straight from registers -- a wave instruction
// writes 4 rows x (16 lanes x TN values) = whole 128-byte lines (bf16, TN = 4), no LDS round trip.
template <typename T, int BM, int BN, int WMW, bool RED = false>
__global__ __launch_bounds__(256) void igemm_nt_kernel(IGemmArgs a) {
    // WMW x (4/WMW) waves.  BK = 64 with ONE LDS stage + one stage in registers: the next tile's 16-byte loads (8 per
    // thread for a 128x128 tile) are in flight during the whole compute phase.  The small-K convs
    // are bound by memory concurrency (bytes in flight per CU), not by MFMA or LDS.
    constexpr int BK = 64;
    constexpr int VEC = Vec<T>::N;
    constexpr int CPR = BK / VEC;          // 16-byte chunks per tile row
    constexpr int LD = BK + VEC;           // padded LDS row, elements (16-byte aligned rows)
    constexpr int RSTEP = 256 / CPR;
    constexpr int ACH = BM / RSTEP, BCH = BN / RSTEP;
    constexpr int WNW = 4 / WMW;           // waves along N
    constexpr int RW = BM / WMW;           // rows per wave
    constexpr int WN = BN / WNW;           // columns per wave
    constexpr int TM = RW / 16, TN = WN / 16;
    static_assert(TN == 2 || TN == 4 || TN == 8, "lane owns 2, 4 or 8 consecutive columns");
    __shared__ __attribute__((aligned(16))) T As[BM * LD];
    __shared__ __attribute__((aligned(16))) T Bs[BN * LD];

    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ W = (const T*)a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNW, wn = wave % WNW;
    const int tiles_n = (a.N + BN - 1) / BN;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int kc = tid % CPR, r0 = tid / CPR;

    RowPos rp[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) rp[i] = row_pos(m0 + r0 + i * RSTEP, a.M, a.g, a.fd_hw, a.fd_w);
    KPos kp = k_pos(kc * VEC, a.g);
    int brow[BCH];                         // permuted LDS row of each W row this thread stages
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int nl = r0 + i * RSTEP, rem = nl % WN;
        brow[i] = (nl / WN) * WN + (rem % TN) * 16 + rem / TN;
    }

    Vec<T> ra[ACH], rb[BCH];
    auto load_tile = [&]() {          // loads the tile at the current kp, then advances kp by BK
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            int64_t off = a_offset(rp[i], kp, a.K, a.g);
            ra[i] = off >= 0 ? vload<T>(X + off) : vzero<T>();
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            int n = n0 + r0 + i * RSTEP;
            rb[i] = (n < a.N && kp.k < a.K) ? vload<T>(W + (int64_t)n * a.ldw + kp.k) : vzero<T>();
        }
        k_advance(kp, BK, a.g);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < ACH; ++i) vstore<T>(&As[(r0 + i * RSTEP) * LD + kc * VEC], ra[i]);
#pragma unroll
        for (int i = 0; i < BCH; ++i) vstore<T>(&Bs[brow[i] * LD + kc * VEC], rb[i]);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (a.K + BK - 1) / BK;
    const int fr = lane & 15, fg = lane >> 4;
    STAMP(0);
    load_tile();
    STAMP(1);
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt) __syncthreads();                 // every wave is done reading the previous tile
        store_tile();
        if (kt == 0) STAMP(2);
        __syncthreads();
        if (kt + 1 < nkt) load_tile();           // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            Frag<T> af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i].load(&As[(wm * RW + i * 16 + fr) * LD + ks * 32 + fg * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j].load(&Bs[(wn * WN + j * 16 + fr) * LD + ks * 32 + fg * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma16(acc[i][j], af[i], bf[j]);
        }
    }

    STAMP(3);
    nt_epilogue<T, BM, BN, WMW, RED>(a, acc, m0, n0, reinterpret_cast<float*>(As));
    STAMP(5);
#ifdef CAPMI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(6);
#endif
}

// ------------------------------------------------------------------ NT kernel, LDS-DMA pipeline (bf16, deep K)
// 128x128x32 tiles, a ring of 4 LDS stages filled by global_load_lds (no staging VGPRs): three
// stages (48 KB per workgroup) are in flight while one is being multiplied.  A wave-instruction of
// the DMA writes 1 KiB of LDS linearly (wave-uniform base + lane*16), so the tile rows are stored
// unpadded (64 B) and the ds_read_b128 bank conflicts are removed by an XOR swizzle applied on the
// SOURCE side: LDS slot (row, p) receives global chunk p ^ G((row >> 2) & 3) (lds_swz4), and fragment reads use
// the same involution.  Padding taps / out-of-range rows read a 64-byte zero page.  Counted
// s_waitcnt vmcnt(8) (two younger stages stay in flight) + ONE raw s_barrier per k-tile; no
// __syncthreads() in the loop (it would drain the DMA queue).
// G(q) for q = (row >> 2) & 3: ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,
// 28-31}, ... -- MI355X_MICROARCH.md, LDS): every group holds rows 0..15 once, rows 4-11 with the neighbouring k-chunk
// of rows 0-3 / 12-15, and its sixteen 16-byte slots (row % 4) * 4 + p must differ.  G = (0, 2, 3, 1) does that for all
// four groups; the identity (G(q) = q, which is conflict-free for contiguous 16-lane groups) was a 2-way conflict on
// every fragment read: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 50 % in every GEMM kernel.
__device__ __forceinline__ int lds_swz4(int q) { return (0x1320 >> (4 * (q & 3))) & 3; }
__device__ __attribute__((aligned(64))) unsigned int capmi_zero_page[16];

#ifndef CAPMI_NT_ABL
#define CAPMI_NT_ABL 0
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KG = 2 / 4: 8 / 16 waves in KG k-groups -- group g takes the k-steps s with s % KG == g (its own part of every ring slot)
// and the groups' accumulators meet in LDS before the epilogue.  Same tile, same bytes, twice the waves: for
// deep-K layers whose grid is below ~2 workgroups per CU the loop is bound by per-wave latency, not by MFMA.
// INBN (LIN = 1 only: a 1x1 convolution, no padding): batch norm + ReLU of the input tensor applied to the A fragments
// after the LDS read -- x is the producing convolution's RAW output, its normalised / activated form is never stored
// (MobileNetV2.py:88-121: conv -> batch_norm -> relu is ONE unit of the reference graph; the unit boundary moves from
// the producer's output to the consumer's operand).  With 4x1 waves every A row belongs to one wave, so the transform runs
// once per staged element; per-channel coefficients come from a table in LDS.
template <int BM, int BN, int NST, bool RED, int LIN, int KG = 1, int INBN_KC = 0, int EPI = 0>        // INBN_KC: channels the coefficient table holds (0: off)
__device__ __forceinline__ void nt_glds_body(const IGemmArgs& a, int block, int nblocks) {
    typedef bf16 T;
    constexpr bool INBN = INBN_KC > 0;
    static_assert(!INBN || (LIN == 1 && KG == 1 && !RED), "operand-path batch norm: 1x1 convolutions on the plain kernel");
    constexpr int BK = 32, WMW = 4;
    constexpr int TM = BM / 64, TN = BN / 16;           // 4x1 waves: BM/4 rows x BN columns each
    constexpr int AOPB = BM * BK * 2, BOPB = BN * BK * 2;   // bytes of the A / B operand tiles
    constexpr int STB = AOPB + BOPB;                    // bytes per stage
    constexpr int ACNT = BM / 64, BCNT = BN / 64;       // DMA instructions per thread per stage
    constexpr int NGL = ACNT + BCNT;
    __shared__ __attribute__((aligned(1024))) char smem[NST * KG * STB < 4096 ? 4096 : NST * KG * STB];
    __shared__ __attribute__((aligned(16))) float inbn_tab[INBN ? 3 * INBN_KC : 4];

    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ W = (const T*)a.w;
    const int grp = KG == 1 ? 0 : (int)(threadIdx.x >> 8);       // k-group of this wave
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // position inside the group
    const int tiles_n = (a.N + BN - 1) / BN;
    int split = 0, kbeg = 0, kend = a.K;
    if (LIN == 1 && a.ksplit > 1) {     // split-K over workgroups: blocks [s * tiles, (s + 1) * tiles) work k in [s * kper, (s + 1) * kper)
        nblocks /= a.ksplit;
        split = block / nblocks;
        block -= split * nblocks;
        kbeg = split * a.kper;
        kend = min(a.K, kbeg + a.kper);
    }
    const int tile = xcd_swizzle(block, nblocks);
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    if (m0 >= a.M) return;              // padding block of a grouped launch

    // this thread's DMA slots: rows (tid>>2) [+64], LDS chunk position tid&3, i.e. global chunk
    // (tid&3) ^ G((tid>>4)&3) of that row ((row>>2)&3 is the same for row and row+64)
    const int chunk = (tid & 3) ^ lds_swz4(tid >> 4);
    RowPos rp[ACNT];
    const T* wrow[BCNT];
    bool wok[BCNT];
#pragma unroll
    for (int i = 0; i < ACNT; ++i) rp[i] = row_pos(m0 + i * 64 + (tid >> 2), a.M, a.g, a.fd_hw, a.fd_w);
#pragma unroll
    for (int i = 0; i < BCNT; ++i) {
        const int l = i * 64 + (tid >> 2);
        const int n = n0 + TN * (l & 15) + (l >> 4);    // LDS row j*16+fr holds weight column TN*fr + j
        wok[i] = n < a.N;
        wrow[i] = W + (int64_t)(wok[i] ? n : 0) * a.ldw;
    }
    KPos kp = k_pos(chunk * 8 + grp * BK, a.g);
    if constexpr (LIN == 1) kp.k += kbeg;
    const T* zero = reinterpret_cast<const T*>(capmi_zero_page);
    // addressing mode LIN: 1 = 1x1 / no padding (any stride): A(m, k) = x[base(m) + k], no tap arithmetic in the
    // loop; 2 = up == 1 and Cin >= BK: taps tracked with selects, no branches; 0 = general (a_offset)
    const T* arow[ACNT];
#pragma unroll
    for (int i = 0; i < ACNT; ++i) arow[i] = rp[i].ok ? X + rp[i].base : nullptr;
    const int Hi = a.g.Hi, Wi = a.g.Wi, ldx = a.g.ldx, Cin = a.g.Cin, kw = a.g.kw;
    auto issue_stage = [&](int st) {    // DMA of the tile at the current kp into ring slot st; advances kp
        char* base = smem + (st * KG + grp) * STB + wave * 1024;
#pragma unroll
        for (int i = 0; i < ACNT; ++i) {
            const T* src;
            if constexpr (LIN == 1) {
                src = (arow[i] && kp.k < kend) ? arow[i] + kp.k : zero;
            } else if constexpr (LIN == 2) {
                const int hn = rp[i].hb + kp.r, wn = rp[i].wb + kp.q;
                const bool ok = arow[i] && kp.k < a.K && (unsigned)hn < (unsigned)Hi && (unsigned)wn < (unsigned)Wi;
                src = arow[i] + ((kp.r * Wi + kp.q) * ldx + kp.c);
                src = ok ? src : zero;
            } else {
                const int64_t off = a_offset(rp[i], kp, a.K, a.g);
                src = off >= 0 ? X + off : zero;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BCNT; ++i) {
            const T* src = (wok[i] && kp.k < (LIN == 1 ? kend : a.K)) ? wrow[i] + kp.k : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + AOPB + i * 4096), 16, 0, 0);
        }
        if constexpr (LIN == 1) kp.k += KG * BK;
        else if constexpr (LIN == 2) {       // Cin >= KG*BK: one channel wrap at most per step
            kp.k += KG * BK;
            kp.c += KG * BK;
            const bool wc = kp.c >= Cin;
            kp.c -= wc ? Cin : 0;
            kp.q += wc ? 1 : 0;
            const bool wq = kp.q == kw;
            kp.q = wq ? 0 : kp.q;
            kp.r += wq ? 1 : 0;
        } else k_advance(kp, KG * BK, a.g);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (kend - kbeg + KG * BK - 1) / (KG * BK);
    const int fr = lane & 15, fg = lane >> 4;
    // fragment byte offsets inside a stage (swizzled chunk position)
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wave * (BM / 4) + i * 16 + fr;
        aoff[i] = row * 64 + ((fg ^ lds_swz4(row >> 2)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = j * 16 + fr;
        boff[j] = AOPB + row * 64 + ((fg ^ lds_swz4(row >> 2)) << 4);
    }

    // EPI 6: the epilogue's mask bytes, loaded now (in FRONT of the DMA issues: vector-memory operations retire in order, so the
    // loop's counted waits stay exact) and consumed after the loop
    unsigned pmask[TM][4];
    if constexpr (EPI == 6 || EPI == 7) {
        if (grp == 0) nt_prefetch_mask<T, BM, BN, WMW, EPI == 6>(a, m0, n0, pmask);
    }
    // CAPMI_NT_ABL (tools/nt_ablate.hip only, 0 in the library): 1 = no MFMAs, 2 = no LDS reads either, 4 = no DMA
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (!(CAPMI_NT_ABL & 4)) issue_stage(p);
    if constexpr (INBN) {       // behind the prologue's DMA issues: the table's own load latency hides under them; published by the
        inbn_load_table<INBN ? INBN_KC : 4>(inbn_tab, a, a.K, tid);      // first k-step's barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    int slot = 0;                                            // ring slot of stage kt
    for (int kt = 0; kt < nkt; ++kt) {
        // this thread's part of stage kt has landed (the (NST-2)*NGL younger DMAs may still be in flight)
        wait_vmcnt<(NST - 2) * NGL>();
        __builtin_amdgcn_s_barrier();                        // ... and everyone else's; all waves left stage kt-1
        asm volatile("" ::: "memory");
        if (!(CAPMI_NT_ABL & 4)) issue_stage(slot == 0 ? NST - 1 : slot - 1);         // refill the slot that was read in iteration kt-1
        const char* st = smem + (slot * KG + grp) * STB;
        slot = slot + 1 == NST ? 0 : slot + 1;
        if (CAPMI_NT_ABL & 2) continue;
        Frag<T> af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i].load(reinterpret_cast<const T*>(st + aoff[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j].load(reinterpret_cast<const T*>(st + boff[j]));
        f32x4 cf[INBN ? 6 : 1];
        if constexpr (INBN)     // this lane's 8 channels of the k-step: kt * 32 + 8 fg .. + 7 (16-lane groups read the same words: broadcast)
            inbn_coef<INBN ? INBN_KC : 4>(inbn_tab, kt * BK + fg * 8, cf);
        // every fragment read is issued before the first MFMA: one exposed LDS latency per k-step instead of one
        // per pair of MFMAs (the scheduler otherwise recycles two fragment registers; grids below ~2 waves per
        // SIMD have nobody to hide that behind)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (INBN) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i].v = bn_act8(af[i].v, cf[0], cf[1], cf[2], cf[3], cf[4], cf[5], a.in_act);
        }
        if (CAPMI_NT_ABL & 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0][0] += (float)af[i].v[0];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[0][j][1] += (float)bf[j].v[7];
            continue;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma16(acc[i][j], af[i], bf[j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // retire the (zero-page) tail stages before LDS reuse
    __syncthreads();
    if constexpr (KG > 1) {             // groups 1.. hand their accumulators over and leave; group 0 runs the epilogue
        f32x4* xch = reinterpret_cast<f32x4*>(smem);
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) xch[((grp - 1) * TM * TN + i * TN + j) * 256 + tid] = acc[i][j];
        }
        __syncthreads();
        if (grp > 0) {                  // keep the barrier count of the epilogue's statistics / batch-norm-sums path (two), then leave
            if (a.stats || EPI == 7 || EPI == 8) { lds_barrier(); lds_barrier(); }
            return;
        }
#pragma unroll
        for (int g = 0; g < KG - 1; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] += xch[(g * TM * TN + i * TN + j) * 256 + tid];
    }
    nt_epilogue<T, BM, BN, WMW, RED, true, EPI>(a, acc, m0, n0, reinterpret_cast<float*>(smem), (int64_t)split * a.M * a.ldy, (EPI == 6 || EPI == 7) ? pmask : nullptr);
}

template <int BM, int BN, int NST, bool RED = false, int LIN = 0, int EPI = 0>
__global__ __launch_bounds__(256, BM == 64 ? (NST == 3 ? 4 : 3) : (NST == 3 ? 3 : 2)) void igemm_nt_glds_kernel(IGemmArgs a) {
    nt_glds_body<BM, BN, NST, RED, LIN, 1, 0, EPI>(a, blockIdx.x, gridDim.x);
}

// 1x1 convolution whose input is the producer's RAW output: batch norm + ReLU in the A-operand path (nt_glds_body, INBN)
// (KC = 128 / 256 / 512: the table is sized to the layer -- 1.5 / 3 / 6 KB next to a 36 KB ring keeps 64-row tiles at the
// plain kernel's 4 workgroups per CU up to 256 input channels; one workgroup per CU fewer cost a whole round on the 14x14 layers)
template <int BM, int BN, int KC>
__global__ __launch_bounds__(256, BM == 64 ? (KC <= 256 ? 4 : 3) : 2) void igemm_nt_glds_inbn_kernel(IGemmArgs a) {
    nt_glds_body<BM, BN, 3, false, 1, 1, KC, 1>(a, blockIdx.x, gridDim.x);      // (always a training convolution's forward form)
}

template <int BM, int BN, int NST, int LIN, int KG, int EPI = 0>
__global__ __launch_bounds__(256 * KG, (KG == 2 && BM == 64) ? 2 : 1) void igemm_nt_glds_kg_kernel(IGemmArgs a) {
    nt_glds_body<BM, BN, NST, false, LIN, KG, 0, EPI>(a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ 3x3 / stride 1 / pad 1: input tile staged ONCE per channel chunk
// The LDS-DMA kernels above are bound by the bytes they stage (MFMA sits at 25-35 % of a CU's rate, the DMA path at its
// 40-60 GB/s per CU): per k-step a BM x 32 piece of the im2col matrix + a BN x 32 piece of the filter.  For a 3x3
// convolution nine consecutive k-steps of one channel chunk re-stage THE SAME input pixels shifted by one tap.  Here the
// loop runs (channel chunk, tap): the BM output pixels m0..m0+BM-1 of a tile are consecutive in memory (NHWC, stride 1),
// so ONE halo tile -- pixels m0 - (W+1) .. m0 + BM - 1 + (W+1), 32 channels, one 64-byte LDS row each -- serves all nine
// taps: tap (dr, dq) of output row i is LDS row i + (W+1) + dr*W + dq.  Rows whose tap falls outside the image (padding,
// or the neighbouring image row / image the linear window runs into) are zeroed per LANE after the fragment read: an MFMA
// A-fragment lane holds data of exactly one row.  Staged bytes per nine taps at 128 x 128: 9 x 8 KB of filter + ~11-16 KB
// of input instead of 9 x 16 KB (64 -> ~110 flop per staged byte); at BN = 64 the input was two thirds of the bytes.
//   pipeline: filter tiles in a 3-slot ring (as above), halo tiles double-buffered, the next chunk's halo issued at tap 0;
//   the DMA returns in issue order, so `vmcnt` only has to leave the issues of LATER steps outstanding (see the loop).
// Same weight-row permutation, accumulator layout and epilogue as the kernels above.
// INBN: the input is the producing convolution's RAW output; batch norm + ReLU (bn_apply's formula, bn_act8) is applied to
// the halo tile IN LDS, once per staged element: every thread transforms the 16-byte pieces its own DMA wrote (it knows
// they have landed from its own vmcnt; nobody else touches them) -- chunk 0's in the prologue, chunk c + 1's spread over
// taps 2.. of chunk c, under that chunk's MFMAs, into the buffer nobody reads yet.  Padding / out-of-image rows are zeroed per
// lane AFTER the fragment read (vmask), so what the transform makes of their zero-page bytes never reaches an MFMA.
template <int BM, int BN, bool INBN = false, int EPI = 0>
__global__ __launch_bounds__(256, BM == 64 ? 3 : 2) void igemm_nt_halo3_kernel(IGemmArgs a) {
    typedef bf16 T;
    constexpr int BK = 32, WMW = 4;
    constexpr int TM = BM / 64, TN = BN / 16;
    constexpr int WMAX = 56;                                    // widest image row the halo buffer is sized for
    constexpr int HR = (BM + 2 * (WMAX + 1) + 63) / 64 * 64;    // halo rows allocated (a multiple of 64: 16 rows x 4 waves per round)
    constexpr int ACNT = HR / 64, BCNT = BN / 64;               // DMA instructions per thread: halo tile / filter tile
    constexpr int AHB = HR * 64, BOPB = BN * BK * 2, NSTB = 3;
    __shared__ __attribute__((aligned(1024))) char smem[2 * AHB + NSTB * BOPB];
    __shared__ __attribute__((aligned(16))) float inbn_tab[INBN ? 3 * INBN_KMAX : 4];
    char* const bring = smem + 2 * AHB;

    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ W = (const T*)a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (a.N + BN - 1) / BN;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;
    const int Wi = a.g.Wi, Hi = a.g.Hi, Cin = a.g.Cin, ldx = a.g.ldx;
    const int npix = a.g.B * Hi * Wi;
    const int hrows = BM + 2 * (Wi + 1);                        // halo rows in use
    const T* zero = reinterpret_cast<const T*>(capmi_zero_page);

    // ---- DMA slots of this thread.  Halo: round i covers LDS rows 64 i + 16 wave + (lane >> 2), chunk position lane & 3
    const T* asrc[ACNT];
#pragma unroll
    for (int i = 0; i < ACNT; ++i) {
        const int j = 64 * i + 16 * wave + (lane >> 2);
        const int chunk = (lane & 3) ^ lds_swz4(j >> 2);
        const int pix = m0 - (Wi + 1) + j;
        asrc[i] = (j < hrows && pix >= 0 && pix < npix) ? X + (int64_t)pix * ldx + chunk * 8 : nullptr;
    }
    const int bchunk = (tid & 3) ^ lds_swz4(tid >> 4);
    const T* wrow[BCNT];
#pragma unroll
    for (int i = 0; i < BCNT; ++i) {
        const int l = i * 64 + (tid >> 2);
        const int n = n0 + TN * (l & 15) + (l >> 4);            // LDS row j*16+fr holds weight column TN*fr + j
        wrow[i] = n < a.N ? W + (int64_t)n * a.ldw + bchunk * 8 : nullptr;
    }
    auto issue_A = [&](int buf, int cc, bool live) {
        char* base = smem + buf * AHB + wave * 1024;
#pragma unroll
        for (int i = 0; i < ACNT; ++i) {
            const T* src = (live && asrc[i]) ? asrc[i] + cc * BK : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
        }
    };
    auto issue_B = [&](int slot, int koff, bool live) {         // filter columns koff .. koff + 31 of this tile's rows
        char* base = bring + slot * BOPB + wave * 1024;
#pragma unroll
        for (int i = 0; i < BCNT; ++i) {
            const T* src = (live && wrow[i]) ? wrow[i] + koff : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
        }
    };

    // ---- fragment addressing.  Output row of (wave, tm, lane): wave * (BM/4) + 16 tm + fr
    const int fr = lane & 15, fg = lane >> 4;
    int arow[TM];                                               // LDS row of the centre tap
    unsigned vmask[TM];                                         // bit (3 (dr+1) + (dq+1)): that tap reads a pixel of the image
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wave * (BM / 4) + i * 16 + fr;
        arow[i] = row + Wi + 1;
        const int m = m0 + row;
        const int b = fdiv(m, a.fd_hw), rem = m - b * a.fd_hw.d;
        const int h = fdiv(rem, a.fd_w), w = rem - h * a.fd_w.d;
        unsigned bits = 0;
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
            for (int dq = -1; dq <= 1; ++dq)
                if (m < a.M && (unsigned)(h + dr) < (unsigned)Hi && (unsigned)(w + dq) < (unsigned)Wi) bits |= 1u << (3 * (dr + 1) + (dq + 1));
        vmask[i] = bits;
    }
    int boff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = j * 16 + fr;
        boff[j] = row * 64 + ((fg ^ lds_swz4(row >> 2)) << 4);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nch = Cin / BK;
    // operand-path batch norm: piece i of this thread (LDS row 64 i + 16 wave + (lane >> 2), position lane & 3) holds the
    // global chunk (lane & 3) ^ G((lane >> 4) & 3) of its row -- the same for every i -- i.e. channels 32 cc + 8 gch .. + 7
    const int gch = (lane & 3) ^ lds_swz4(lane >> 4);
    auto piece_ptr = [&](int buf, int i) { return reinterpret_cast<bf16x8*>(smem + buf * AHB + wave * 1024 + i * 4096 + lane * 16); };
    auto transform_piece = [&](int buf, int i, const f32x4 (&cf)[6]) {
        bf16x8* ptr = piece_ptr(buf, i);
        *ptr = bn_act8(*ptr, cf[0], cf[1], cf[2], cf[3], cf[4], cf[5], a.in_act);
    };
    auto load_coef = [&](int cc, f32x4 (&cf)[6]) { inbn_coef<INBN_KMAX>(inbn_tab, cc * BK + gch * 8, cf); };
    if constexpr (INBN) {
        inbn_load_table<INBN_KMAX>(inbn_tab, a, Cin, tid);
        __syncthreads();
    }
    unsigned pmask[TM][4];                                      // EPI 6: the epilogue's mask bytes, in front of the DMA issues (see nt_glds_body)
    if constexpr (EPI == 6 || EPI == 7) nt_prefetch_mask<T, BM, BN, WMW, false>(a, m0, n0, pmask);      // (the halo kernel never scatters)
    // prologue: halo of chunk 0, filter tiles of steps 0 and 1 (taps 0 and 1 of chunk 0)
    issue_A(0, 0, true);
    issue_B(0, 0, true);
    issue_B(1, Cin, true);
    f32x4 cfn[INBN ? 6 : 1];                                    // coefficients of the chunk being transformed
    if constexpr (INBN) {
        wait_vmcnt<0>();                                        // this thread's pieces of chunk 0 have landed
        load_coef(0, cfn);
#pragma unroll
        for (int i = 0; i < ACNT; ++i) transform_piece(0, i, cfn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // ... and are rewritten before the first step's barrier
    }
    int slot = 0;                                               // ring slot of the current step's filter tile
    for (int cc = 0; cc < nch; ++cc) {
        const char* abuf = smem + (cc & 1) * AHB;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // This step's filter tile has landed (and, in issue order, everything older: this chunk's halo).  Only the
            // order BETWEEN steps is relied on -- inside a step the compiler is free to emit the filter and the halo
            // issues in any order (it put the halo first: with vmcnt(BCNT + ACNT) at tap 2 the filter tile issued
            // next to the halo at tap 0 could still be in flight -- a few wrong 16-row strips per launch, found by the
            // in-situ layer test).  Tap 1 waits for a tile issued one step BEFORE the halo (the halo + the next tile may
            // stay in flight); tap 2 for one issued WITH it (only the next step's tile may); the very first step for
            // the whole prologue.
#ifdef CAPMI_HALO_SAFE
            wait_vmcnt<0>();
#else
            if (tap == 1) wait_vmcnt<BCNT + ACNT>();
            else if (tap == 0 && cc == 0) wait_vmcnt<0>();
            else wait_vmcnt<BCNT>();
#endif
            __builtin_amdgcn_s_barrier();                       // ... everyone's; all waves have left the previous step
            asm volatile("" ::: "memory");
            {   // refill the slot read in the previous step with the filter tile of step + 2; at tap 0 the next chunk's halo
                const int tap2 = (tap + 2) % 9, cc2 = cc + (tap + 2) / 9;
                issue_B(slot == 0 ? NSTB - 1 : slot - 1, tap2 * Cin + cc2 * BK, cc2 < nch);
                if (tap == 0) issue_A((cc + 1) & 1, cc + 1, cc + 1 < nch);
            }
            const char* bst = bring + slot * BOPB;
            slot = slot + 1 == NSTB ? 0 : slot + 1;
            const int shift = (tap / 3 - 1) * Wi + (tap % 3 - 1);
            Frag<T> af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int j = arow[i] + shift;
                af[i].load(reinterpret_cast<const T*>(abuf + j * 64 + ((fg ^ lds_swz4(j >> 2)) << 4)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j].load(reinterpret_cast<const T*>(bst + boff[j]));
            // INBN: the next chunk's halo tile (issued at tap 0 of this chunk; from tap 2 on this thread's vmcnt wait has
            // covered it) is transformed one piece per tap, in the buffer nobody reads before the next chunk's tap 0.  The
            // piece is read WITH the fragments (one LDS latency for all of them), its ~36 VALU instructions are dealt into the
            // gaps of this step's MFMAs (an MFMA holds the SIMD's issue for 8 of its 16 cycles: two or three plain VALU fit
            // behind each) and it is written back behind them.  No wait for that write here: the next step's fragment reads
            // retire it (LDS operations complete in order) long before the barrier in front of the first read of this buffer.
            const bool XF = INBN && tap >= 2 && tap < 2 + ACNT;        // (the tap loop is fully unrolled: a constant per copy)
            const bool xf_live = XF && cc + 1 < nch;
            bf16x8 piece = {};
            if constexpr (INBN) {
                if (XF && tap == 2 && xf_live) load_coef(cc + 1, cfn);
                if (XF && xf_live) piece = *piece_ptr((cc + 1) & 1, tap - 2);
            }
            __builtin_amdgcn_sched_barrier(0);                  // every fragment read in flight before the first MFMA
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (!((vmask[i] >> tap) & 1u)) af[i] = Frag<T>{};       // padding / beyond the image: this lane's row contributes zero
            if constexpr (INBN) {
                if (XF) piece = bn_act8(piece, cfn[0], cfn[1], cfn[2], cfn[3], cfn[4], cfn[5], a.in_act);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma16(acc[i][j], af[i], bf[j]);
            if constexpr (INBN) {
                if (XF) {
#pragma unroll
                    for (int q = 0; q < TM * TN; ++q) {         // 1 MFMA, then up to 3 VALU, ...: the transform rides in the MFMA shadow
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, (40 + TM * TN - 1) / (TM * TN), 0);
                    }
                    if (xf_live) *piece_ptr((cc + 1) & 1, tap - 2) = piece;
                }
            }
            // The tap loop is unrolled, so without this the scheduler sinks the MFMAs of a step -- and with them the
            // lgkmcnt wait for its fragment reads -- below the NEXT step's barrier and DMA issue: the refill of the slot
            // those reads come from was then in flight while the reads were not yet known to have returned (WAR on LDS;
            // a few wrong 16-row strips per launch whenever LDS traffic of a co-resident workgroup delayed the reads).
            // Pinned here, every wave has its fragments in registers before it arrives at the barrier.
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // retire the (zero-page) tail issues before LDS reuse
    __syncthreads();
    nt_epilogue<T, BM, BN, WMW, false, true, EPI>(a, acc, m0, n0, reinterpret_cast<float*>(smem), 0, (EPI == 6 || EPI == 7) ? pmask : nullptr);
}

// Several independent problems (the parity classes of a strided data gradient) in ONE launch: the
// classes are small GEMMs that under-fill the chip one by one.  Block ranges start at multiples of 8 so
// that the XCD-contiguous tile order holds inside every problem.
constexpr int NT_GROUP_MAX = 4;
struct NtGroup {
    IGemmArgs a[NT_GROUP_MAX];
    int first[NT_GROUP_MAX + 1];
    int count;
};
template <int BM, int BN, int NST, int LIN, int EPI = 0>
__global__ __launch_bounds__(256, BM == 64 ? (NST == 3 ? 4 : 3) : (NST == 3 ? 3 : 2)) void igemm_nt_glds_group_kernel(NtGroup g) {
    const int b = blockIdx.x;
    int p = 0;
    while (p + 1 < g.count && b >= g.first[p + 1]) ++p;
    nt_glds_body<BM, BN, NST, false, LIN, 1, 0, EPI>(g.a[p], b - g.first[p], g.first[p + 1] - g.first[p]);
}

// ------------------------------------------------------------------ skinny NT kernel (M <= 64)
// The decoder's recurrent GEMMs ([B,H]x[H,4H] forward, [B,4H]x[4H,H] in BPTT; B = 64) are chains of
// tiny dependent launches: with the tiled kernel they run on 4-16 workgroups and pay one memory
// latency per k-tile.  Here a workgroup owns a 64x32 output tile, its 4 waves split K four ways,
// every wave streams its fragments straight from global memory (L2-resident) into MFMA operand
// layout -- no LDS, no barrier in the main loop, loads several k-steps ahead -- and the four partial
// tiles meet in LDS once.  Plain row-major A only (no im2col).
template <typename T>
__global__ __launch_bounds__(256) void igemm_nt_skinny_kernel(IGemmArgs a) {
    constexpr int TM = 4, TN = 2;                      // 64 x 32 tile per workgroup
    __shared__ __attribute__((aligned(16))) float red[4][TM * TN][64][4];
    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ W = (const T*)a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 32;
    const int kper = a.K / 4;                          // launcher guarantees K % 128 == 0
    const int kbeg = wave * kper;
    const int ldx = a.g.ldx;

    const T* arow[TM];
    bool aok[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = i * 16 + fr;
        aok[i] = m < a.M;
        arow[i] = X + (int64_t)(aok[i] ? m : 0) * ldx + kbeg + fg * 8;
    }
    const T* brow[TN];
    bool bok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + TN * fr + j;                // lane owns TN consecutive columns (see igemm_nt_kernel)
        bok[j] = n < a.N;
        brow[j] = W + (int64_t)(bok[j] ? n : 0) * a.ldw + kbeg + fg * 8;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 4
    for (int k = 0; k < kper; k += 32) {
        Frag<T> af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            af[i].load(arow[i] + k);
            if (!aok[i]) af[i] = Frag<T>{};
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf[j].load(brow[j] + k);
            if (!bok[j]) bf[j] = Frag<T>{};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma16(acc[i][j], af[i], bf[j]);
    }
    // cross-wave reduction: wave w finishes rows 16w..16w+15 (tiles (w, 0..TN-1))
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(&red[wave][i * TN + j][lane][0]) = acc[i][j];
    __syncthreads();
    float v[4][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x4 s = *reinterpret_cast<const f32x4*>(&red[0][wave * TN + j][lane][0]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            f32x4 t = *reinterpret_cast<const f32x4*>(&red[w][wave * TN + j][lane][0]);
            s += t;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r][j] = s[r];
    }
    const int col0 = n0 + TN * fr;
    if (col0 >= a.N) return;
    const T* addend = (const T*)a.addend;
    const T* ysaved = (const T*)a.ysaved;
    const bool vec_ok = (a.ldy % TN == 0) && (!addend || a.ld_addend % TN == 0) && (!a.dact || a.ld_saved % TN == 0) && col0 + TN <= a.N;
    const int nv = min(TN, a.N - col0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = wave * 16 + fg * 4 + r;
        if (row >= a.M) continue;
        float o[TN], t[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) o[j] = v[r][j] + ((a.bias && j < nv) ? a.bias[col0 + j] : 0.f);
        if (addend) {
            if (vec_ok) load_run<T, TN>(addend + row * a.ld_addend + col0, t);
            else
#pragma unroll
                for (int j = 0; j < TN; ++j) t[j] = j < nv ? to_f32(addend[row * a.ld_addend + col0 + j]) : 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) o[j] += t[j];
        }
        act_run<TN>(o, a.act);
        if (a.dact) {
            if (vec_ok) load_run<T, TN>(ysaved + row * a.ld_saved + col0, t);
            else
#pragma unroll
                for (int j = 0; j < TN; ++j) t[j] = j < nv ? to_f32(ysaved[row * a.ld_saved + col0 + j]) : 0.f;
            dact_run<TN>(o, t, a.dact);
        }
        if (vec_ok) {
            if (a.out_f32) store_run<float, TN>((float*)a.y + row * a.ldy + col0, o);
            else store_run<T, TN>((T*)a.y + row * a.ldy + col0, o);
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (j < nv) {
                    if (a.out_f32) ((float*)a.y)[row * a.ldy + col0 + j] = o[j];
                    else ((T*)a.y)[row * a.ldy + col0 + j] = from_f32<T>(o[j]);
                }
        }
    }
}

// Tile / wave-grid selection shared by the launcher and the statistics-workspace query.
//   bf16, N > 64 : 4x1 waves, each lane owns 8 consecutive columns (16-byte stores);
//                  128x128 (2 workgroups/CU) when the reduction is deep (K >= 512) and the grid is
//                  large, else 64x128 (32 accumulator registers -> 4 workgroups/CU): the small-K
//                  convs are bound by memory concurrency, not by MFMA
//   bf16, N <= 64: 128x64 / 64x64, 4x1 waves (8-byte stores)
//   f32          : 2x2 waves, tiles sized for 64 KiB of static LDS
// ------------------------------------------------------------------ fused LSTM steps (skinny GEMM + pointwise cell)
// The recurrence is 2T dependent launches on a nearly idle chip; each fused kernel is one skinny product (64 x 32
// tile per workgroup, K split over the four waves, fragments straight from L2) with the lstm_unit cell as its
// epilogue, so a time step is ONE launch in each direction.
template <typename T>
__device__ __forceinline__ void skinny_product(const T* const (&arow)[4], const bool (&aok)[4], const T* const (&brow)[2], int kper,
                                               float (*red)[8][64][4], float (&v)[4][2]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < kper; k += 32) {
        Frag<T> af[4], bf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i].load(arow[i] + k);
            if (!aok[i]) af[i] = Frag<T>{};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j].load(brow[j] + k);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma16(acc[i][j], af[i], bf[j]);
    }
    // cross-wave reduction: wave w finishes rows 16w..16w+15
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&red[wave][i * 2 + j][lane][0]) = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x4 sm = *reinterpret_cast<const f32x4*>(&red[0][wave * 2 + j][lane][0]);
#pragma unroll
        for (int w = 1; w < 4; ++w) sm += *reinterpret_cast<const f32x4*>(&red[w][wave * 2 + j][lane][0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r][j] = sm[r];
    }
}

// forward step: gates[b][g*H+u] (in: input part x_t.Wx + bias) += h_prev . Wh^T, then i,f,o,g -> c, h.
// Workgroup = 8 hidden units x 4 gates (tile column 8g + u).
template <typename T>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(const T* __restrict__ h_prev, const T* __restrict__ wh, int ldw, T* gates,
                                                            const T* __restrict__ c_prev, T* h, T* c, int B, int H) {
    __shared__ __attribute__((aligned(16))) float red[4][8][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int u0 = blockIdx.x * 8;
    const int kper = H / 4, kbeg = wave * kper;
    const T* arow[4];
    bool aok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = i * 16 + fr;
        aok[i] = m < B;
        arow[i] = h_prev + (int64_t)(aok[i] ? m : 0) * H + kbeg + fg * 8;
    }
    const T* brow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cj = 2 * fr + j;
        brow[j] = wh + (int64_t)((cj >> 3) * H + u0 + (cj & 7)) * ldw + kbeg + fg * 8;
    }
    float v[4][2];
    skinny_product<T>(arow, aok, brow, kper, red, v);
    __syncthreads();                                   // every wave has read the partial tiles
    float* tile = &red[0][0][0][0];                    // [64][33]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) tile[(wave * 16 + fg * 4 + r) * 33 + 2 * fr + j] = v[r][j];
    __syncthreads();
    const int b = tid & 63;
    if (b >= B) return;
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
        const int u = (tid >> 6) * 2 + uu;
        T* gr = gates + (int64_t)b * 4 * H + u0 + u;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const T stored = from_f32<T>(tile[b * 33 + 8 * g + u] + to_f32(gr[g * H]));
            gr[g * H] = stored;                        // the pre-activation BPTT reads back
            pre[g] = to_f32(stored);
        }
        const float i_ = sigmoid_for<T>(pre[0]), f_ = sigmoid_for<T>(pre[1]), o_ = sigmoid_for<T>(pre[2]), g_ = tanh_for<T>(pre[3]);
        const float cp = c_prev ? to_f32(c_prev[(int64_t)b * H + u0 + u]) : 0.f;
        const float cn = f_ * cp + i_ * g_;
        c[(int64_t)b * H + u0 + u] = from_f32<T>(cn);
        h[(int64_t)b * H + u0 + u] = from_f32<T>(o_ * tanh_for<T>(cn));
    }
}

// backward step: dh_{t-1} = dh_in + dG_t . Wh (whT rows = hidden units, reduction over the 4H gates), then the
// cell backward of step t-1 on the thread's own 4 x 2 outputs: dG_{t-1}, dc_{t-2}.
template <typename T>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const T* __restrict__ dg_t, const T* __restrict__ whT, int ldwT, const T* __restrict__ dh_in,
                                                            const T* __restrict__ gates_p, const T* __restrict__ c_pp, const T* __restrict__ c_p,
                                                            const T* __restrict__ dc_in, T* dg_p, T* dc_prev, int dc_prev_acc, int B, int H) {
    __shared__ __attribute__((aligned(16))) float red[4][8][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 32;
    const int kper = H, kbeg = wave * kper;            // reduction length 4H, a quarter per wave
    const T* arow[4];
    bool aok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = i * 16 + fr;
        aok[i] = m < B;
        arow[i] = dg_t + (int64_t)(aok[i] ? m : 0) * 4 * H + kbeg + fg * 8;
    }
    const T* brow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) brow[j] = whT + (int64_t)(n0 + 2 * fr + j) * ldwT + kbeg + fg * 8;
    // the cell's operands do not depend on the product: load them first, their latency hides under it
    typedef T __attribute__((ext_vector_type(2))) Pair;
    Pair pg[4][4], pc[4], pcp[4], pdc[4], pdh[4];
    const int col0 = n0 + 2 * fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + fg * 4 + r;
        const bool ok = row < B;
        const int64_t e = (int64_t)(ok ? row : 0) * H + col0, ge = (int64_t)(ok ? row : 0) * 4 * H + col0;
#pragma unroll
        for (int g = 0; g < 4; ++g) pg[r][g] = *reinterpret_cast<const Pair*>(gates_p + ge + g * H);
        pc[r] = *reinterpret_cast<const Pair*>(c_p + e);
        pcp[r] = c_pp ? *reinterpret_cast<const Pair*>(c_pp + e) : Pair{};
        pdc[r] = dc_in ? *reinterpret_cast<const Pair*>(dc_in + e) : Pair{};
        pdh[r] = dh_in ? *reinterpret_cast<const Pair*>(dh_in + e) : Pair{};
    }
    float v[4][2];
    skinny_product<T>(arow, aok, brow, kper, red, v);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + fg * 4 + r;
        if (row >= B) continue;
        const int64_t e = (int64_t)row * H + col0, ge = (int64_t)row * 4 * H + col0;
        Pair odg[4], odc;
        const Pair old_dc = (dc_prev && dc_prev_acc) ? *reinterpret_cast<const Pair*>(dc_prev + e) : Pair{};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float dhh = to_f32(from_f32<T>(v[r][j] + (float)pdh[r][j]));
            const float i_ = sigmoid_for<T>((float)pg[r][0][j]), f_ = sigmoid_for<T>((float)pg[r][1][j]);
            const float o_ = sigmoid_for<T>((float)pg[r][2][j]), g_ = tanh_for<T>((float)pg[r][3][j]);
            const float tc = tanh_for<T>((float)pc[r][j]);
            const float dct = (float)pdc[r][j] + dhh * o_ * (1.f - tc * tc);
            const float cpv = (float)pcp[r][j];
            odg[0][j] = from_f32<T>(dct * g_ * i_ * (1.f - i_));
            odg[1][j] = from_f32<T>(dct * cpv * f_ * (1.f - f_));
            odg[2][j] = from_f32<T>(dhh * tc * o_ * (1.f - o_));
            odg[3][j] = from_f32<T>(dct * i_ * (1.f - g_ * g_));
            odc[j] = from_f32<T>(dct * f_ + (float)old_dc[j]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<Pair*>(dg_p + ge + g * H) = odg[g];
        if (dc_prev) *reinterpret_cast<Pair*>(dc_prev + e) = odc;
    }
}

// ------------------------------------------------------------------ the whole recurrence of one LSTM layer in ONE launch
// T dependent steps were 2T launches per direction on a nearly idle chip (10.9 us per forward step, 15 + 6 us per
// backward step at B = 64, H = 512).  Here the grid stays resident for all T steps: every wave keeps its slice of the
// recurrent weights in registers for the whole sequence, and the steps are separated by a grid barrier instead of a
// kernel boundary.  Same k-split, MFMA order and rounding points as lstm_step_{fwd,bwd}_kernel, so the results equal
// the per-step launches up to the contraction of a*b + c*d in the cell (forward: bit for bit in bf16).
//
// Hand-off between steps (cdna_hip_programming.md, Guideline 16; MI355X_MICROARCH.md, inter-workgroup visibility:
// per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores).  The bytes one workgroup writes and
// the others read -- h_t forward, dG_t backward -- are stored WRITE-THROUGH (sc1, 4 or 8 bytes per store) and loaded
// with sc1 loads (16 bytes, L1 bypassed); every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets
// at its barrier, ONE lane adds to ONE arrival counter (agent scope) and polls it with sc1 loads until step *
// workgroups have arrived; the other waves load behind the workgroup barrier that lane then joins.  No fence: a release
// (buffer_wbl2) costs 2-6 us per step with freshly dirtied lines, an acquire (buffer_inv) 1.7 us -- measured 13-15 us
// per step with them, against 10.9 for a kernel boundary.  `fences` != 0 adds them back (agent-scope release before the
// arrival, acquire behind the poll; CAPMI_LSTM_SEQ=2).  Everything else a workgroup touches is either written before the
// launch or written and re-read by the same thread.  Every handed-off row block is written exactly once per launch
// before it is read.  Grids are <= 128 workgroups of 256 threads: they fit the chip at once, but they are co-resident only
// once the side lane's (finite) weight-gradient kernels have let them in -- progress rests on those kernels draining, not on
// an occupancy guarantee.  Every spin is bounded: after
// SEQ_SPIN_LIMIT polls a workgroup sets sync[1] and runs on (later barriers see the flag and do not wait), so the launch
// always drains.  sync[0] = arrival counter and sync[1] = "somebody gave up in THIS launch" are zeroed by capmi_lstm_seq_*
// before every launch; sync[2] is STICKY: set together with sync[1], never cleared by the library -- the host reads and
// clears it (DecoderRunner.check_sync), so a time-out in step N is still visible after steps N+1.. have been enqueued.
constexpr unsigned SEQ_SPIN_LIMIT = 1u << 21;
typedef unsigned int seq_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void seq_grid_barrier(unsigned* sync, unsigned target, int fences) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's (write-through) stores have left the chip's caches
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fences) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (kept: ROCm 7.2 can drop the fence's own wait)
        }
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SEQ_SPIN_LIMIT || __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sticky: only the host clears it
                break;
            }
        }
        if (fences) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the invalidate has completed before the barrier opens
        }
    }
    __syncthreads();
}
// 8 consecutive elements at byte offset `off` of the buffer, L1 bypassed
__device__ __forceinline__ void load_frag_sc1(Frag<bf16>& f, __amdgpu_buffer_rsrc_t r, unsigned off) {
    const seq_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
    f.v = __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void load_frag_sc1(Frag<float>& f, __amdgpu_buffer_rsrc_t r, unsigned off) {
    const seq_u32x4 ua = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16), ub = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 16);
    const f32x4 a = __builtin_bit_cast(f32x4, ua), b = __builtin_bit_cast(f32x4, ub);      // (whole vectors: a bit_cast of ONE element read element 0 every time)
#pragma unroll
    for (int i = 0; i < 4; ++i) { f.v[i] = a[i]; f.v[4 + i] = b[i]; }
}
// two adjacent elements as ONE write-through store
__device__ __forceinline__ void store_pair_sc1(bf16* p, float x, float y) {
    const bf16 a = (bf16)x, b = (bf16)y;
    const unsigned v = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_pair_sc1(float* p, float x, float y) {
    const unsigned long long v = (unsigned long long)__builtin_bit_cast(unsigned, x) | ((unsigned long long)__builtin_bit_cast(unsigned, y) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// forward: blocks 1..T of hbuf / cbuf from block 0 (the zero state); gates [T][B][4H] in: input part + bias, out: the
// full pre-activations BPTT reads back.  Workgroup = 8 hidden units x 4 gates x all rows, as lstm_step_fwd_kernel.
template <typename T, int KS>                                   // KS = H / 128: k-steps of 32 per wave
__global__ __launch_bounds__(256) void lstm_seq_fwd_kernel(T* hbuf, const T* __restrict__ wh, int ldw, T* gates, T* cbuf,
                                                           int B, int H, int Tn, unsigned* sync, int fences) {
    __shared__ __attribute__((aligned(16))) float red[4][8][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int u0 = blockIdx.x * 8;
    const int kbeg = wave * (H / 4);
    Frag<T> wf[2][KS];                                          // this wave's recurrent weights, resident for the whole sequence
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cj = 2 * fr + j;
        const T* row = wh + (int64_t)((cj >> 3) * H + u0 + (cj & 7)) * ldw + kbeg + fg * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[j][ks].load(row + ks * 32);
    }
    const __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)hbuf, 0, (unsigned)((size_t)(Tn + 1) * B * H * sizeof(T)), 0x00020000);
    bool aok[4];
    unsigned aoff[4];                                           // byte offsets inside one row block of hbuf
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = i * 16 + fr;
        aok[i] = m < B;
        aoff[i] = (unsigned)(((aok[i] ? m : 0) * H + kbeg + fg * 8) * sizeof(T));
    }
    // the cell: row b, units u0 + up, u0 + up + 1 -- four neighbouring lanes cover the workgroup's 8 units of one row (16
    // contiguous bytes in bf16), a wave 16 rows: a gate / state access of a wave touches 16 lines, not 64 (with the row
    // in the lane index the forward kernel took 11.6 us per step against 5.7 for the backward one)
    const int b = tid >> 2, up = (tid & 3) * 2;
    float* tile = &red[0][0][0][0];                             // [64][33]
    const unsigned blk = (unsigned)((size_t)B * H * sizeof(T));
    typedef T __attribute__((ext_vector_type(2))) Pair;
    for (int t = 0; t < Tn; ++t) {
        T* gt = gates + (int64_t)t * B * 4 * H;
        T* cp = cbuf + (int64_t)t * B * H;
        T* hn = hbuf + (int64_t)(t + 1) * B * H;
        // the cell's own operands do not depend on the other workgroups: fetched before the barrier
        Pair pg[4], pcv;
        if (b < B) {
#pragma unroll
            for (int g = 0; g < 4; ++g) pg[g] = *reinterpret_cast<const Pair*>(gt + (int64_t)b * 4 * H + g * H + u0 + up);
            pcv = *reinterpret_cast<const Pair*>(cp + (int64_t)b * H + u0 + up);
        }
        if (t > 0) {
            seq_grid_barrier(sync, (unsigned)t * gridDim.x, fences);    // every workgroup has written its part of h_{t-1} (block t)
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<T> af[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    load_frag_sc1(af[i], hres, (unsigned)t * blk + aoff[i] + ks * 32 * (unsigned)sizeof(T));
                    if (!aok[i]) af[i] = Frag<T>{};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma16(acc[i][j], af[i], wf[j][ks]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&red[wave][i * 2 + j][lane][0]) = acc[i][j];
            __syncthreads();
            float v[4][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 sm = *reinterpret_cast<const f32x4*>(&red[0][wave * 2 + j][lane][0]);
#pragma unroll
                for (int w = 1; w < 4; ++w) sm += *reinterpret_cast<const f32x4*>(&red[w][wave * 2 + j][lane][0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r][j] = sm[r];
            }
            __syncthreads();                                    // every wave has read the partial tiles
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) tile[(wave * 16 + fg * 4 + r) * 33 + 2 * fr + j] = v[r][j];
            __syncthreads();
        }
        if (b < B) {
            float hv[2];
            Pair og[4], oc;
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int u = up + uu;
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (t > 0) {
                        og[g][uu] = from_f32<T>(tile[b * 33 + 8 * g + u] + to_f32(pg[g][uu]));     // the pre-activation BPTT reads back
                        pre[g] = to_f32(og[g][uu]);
                    } else {
                        pre[g] = to_f32(pg[g][uu]);             // h_{-1} = 0: the input part is the pre-activation
                    }
                }
                float i_, f_, o_, g_, cn;
                const float cpv = to_f32(pcv[uu]);
                if (t > 0) {        // the arithmetic of lstm_step_fwd_kernel ...
                    i_ = sigmoid_for<T>(pre[0]), f_ = sigmoid_for<T>(pre[1]), o_ = sigmoid_for<T>(pre[2]), g_ = tanh_for<T>(pre[3]);
                    cn = f_ * cpv + i_ * g_;
                    hv[uu] = o_ * tanh_for<T>(cn);
                } else {            // ... and of capmi_lstm_cell_fwd, which runs step 0 of the per-step plan
                    i_ = sigmoidf_(pre[0]), f_ = sigmoidf_(pre[1]), o_ = sigmoidf_(pre[2]), g_ = tanhf_(pre[3]);
                    cn = f_ * cpv + i_ * g_;
                    hv[uu] = o_ * tanhf_(cn);
                }
                oc[uu] = from_f32<T>(cn);
            }
            if (t > 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<Pair*>(gt + (int64_t)b * 4 * H + g * H + u0 + up) = og[g];
            }
            *reinterpret_cast<Pair*>(cp + (int64_t)B * H + (int64_t)b * H + u0 + up) = oc;       // read back by this thread only
            store_pair_sc1(hn + (int64_t)b * H + u0 + up, hv[0], hv[1]);                        // read by every workgroup in step t + 1
        }
    }
}

// One cell backward on the thread's row x 2 units (the arithmetic of lstm_step_bwd_kernel; EXACT = the ocml functions
// of capmi_lstm_cell_bwd, which runs the last time step of the per-step plan).  dG goes out write-through.
template <typename T, bool EXACT>
__device__ __forceinline__ void seq_cell_bwd(const float (&v)[2], bool have_product, const T* dh_in, const T* gates_p, const T* c_pp,
                                             const T* c_p, const T* dc_in, T* dg_p, T* dc_prev, int dc_prev_acc, int H, int row, int col0) {
    typedef T __attribute__((ext_vector_type(2))) Pair;
    const int64_t e = (int64_t)row * H + col0, ge = (int64_t)row * 4 * H + col0;
    Pair pg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) pg[g] = *reinterpret_cast<const Pair*>(gates_p + ge + g * H);
    const Pair pc = *reinterpret_cast<const Pair*>(c_p + e);
    const Pair pcp = c_pp ? *reinterpret_cast<const Pair*>(c_pp + e) : Pair{};
    const Pair pdc = dc_in ? *reinterpret_cast<const Pair*>(dc_in + e) : Pair{};
    const Pair pdh = *reinterpret_cast<const Pair*>(dh_in + e);
    const Pair old_dc = (dc_prev && dc_prev_acc) ? *reinterpret_cast<const Pair*>(dc_prev + e) : Pair{};
    float odg[4][2];
    Pair odc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float dhh = have_product ? to_f32(from_f32<T>(v[j] + (float)pdh[j])) : (float)pdh[j];
        float i_, f_, o_, g_, tc;
        if constexpr (EXACT) {
            i_ = sigmoidf_((float)pg[0][j]), f_ = sigmoidf_((float)pg[1][j]), o_ = sigmoidf_((float)pg[2][j]), g_ = tanhf_((float)pg[3][j]);
            tc = tanhf_((float)pc[j]);
        } else {
            i_ = sigmoid_for<T>((float)pg[0][j]), f_ = sigmoid_for<T>((float)pg[1][j]), o_ = sigmoid_for<T>((float)pg[2][j]), g_ = tanh_for<T>((float)pg[3][j]);
            tc = tanh_for<T>((float)pc[j]);
        }
        const float dct = (float)pdc[j] + dhh * o_ * (1.f - tc * tc);
        const float cpv = (float)pcp[j];
        odg[0][j] = dct * g_ * i_ * (1.f - i_);
        odg[1][j] = dct * cpv * f_ * (1.f - f_);
        odg[2][j] = dhh * tc * o_ * (1.f - o_);
        odg[3][j] = dct * i_ * (1.f - g_ * g_);
        odc[j] = from_f32<T>(dct * f_ + (float)old_dc[j]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) store_pair_sc1(dg_p + ge + g * H, odg[g][0], odg[g][1]);
    if (dc_prev) *reinterpret_cast<Pair*>(dc_prev + e) = odc;
}

// backward (BPTT of one layer): gates / cbuf as the forward left them, dhbuf blocks 1..T = d loss / d h_t from everything
// but the recurrence, dcbuf blocks 1..T = d loss / d c_t from outside (dc_outside != 0: the top layer, where the sentinel
// reads c_t; the cell's own d c_{t-1} is then ADDED into block t-1; otherwise dcbuf is scratch the kernel writes).
// Writes dgates [T][B][4H].  Workgroup = 16 rows x 32 hidden units (grid H/32 x ceil(B/16)): every workgroup reads only
// its 16 rows of dG_t (64 KB at H = 512) per step, its waves split the 4H reduction four ways.  whT rows = hidden units.
template <typename T, int KS, int NW>                           // NW waves split the 4H reduction; KS = 4H / NW / 32 k-steps per wave
__global__ __launch_bounds__(64 * NW) void lstm_seq_bwd_kernel(const T* gates, const T* cbuf, const T* __restrict__ whT, int ldwT, const T* dhbuf,
                                                           T* dcbuf, T* dgates, int dc_outside, int B, int H, int Tn, unsigned* sync, int fences) {
    __shared__ __attribute__((aligned(16))) float red[NW][2][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 32, col0 = n0 + 2 * fr;
    const int m0 = blockIdx.y * 16;
    const int kbeg = wave * (4 * H / NW);
    const unsigned nwg = gridDim.x * gridDim.y;
    Frag<T> wf[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const T* row = whT + (int64_t)(n0 + 2 * fr + j) * ldwT + kbeg + fg * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[j][ks].load(row + ks * 32);
    }
    const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc((void*)dgates, 0, (unsigned)((size_t)Tn * B * 4 * H * sizeof(T)), 0x00020000);
    const bool aok = m0 + fr < B;
    const unsigned aoff = (unsigned)(((aok ? m0 + fr : 0) * 4 * H + kbeg + fg * 8) * sizeof(T));
    const unsigned blk = (unsigned)((size_t)B * 4 * H * sizeof(T));
    const int row = m0 + fg * 4 + (wave & 3);                   // the cell (waves 0..3): this row, units col0, col0 + 1
    const bool rok = row < B && wave < 4;
    const int64_t BH = (int64_t)B * H;
    float v[2] = {0.f, 0.f};
    if (rok) {   // step T-1: no later step feeds it
        const int t = Tn - 1;
        seq_cell_bwd<T, true>(v, false, dhbuf + (t + 1) * BH, gates + t * 4 * BH, cbuf + t * BH, cbuf + (t + 1) * BH,
                              dc_outside ? dcbuf + (t + 1) * BH : nullptr, dgates + t * 4 * BH, t > 0 ? dcbuf + t * BH : nullptr,
                              dc_outside, H, row, col0);
    }
    for (int t = Tn - 1; t > 0; --t) {
        seq_grid_barrier(sync, (unsigned)(Tn - t) * nwg, fences);       // every workgroup has written its part of dG_t
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Frag<T> af;
            load_frag_sc1(af, gres, (unsigned)t * blk + aoff + ks * 32 * (unsigned)sizeof(T));
            if (!aok) af = Frag<T>{};
            mma16(acc[0], af, wf[0][ks]);
            mma16(acc[1], af, wf[1][ks]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&red[wave][j][lane][0]) = acc[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float sm = red[0][j][lane][wave & 3];
#pragma unroll
            for (int w = 1; w < NW; ++w) sm += red[w][j][lane][wave & 3];
            v[j] = sm;
        }
        // d h_{t-1} = its outside gradient (block t) + dG_t . Wh, then the cell backward of step t-1
        if (rok)
            seq_cell_bwd<T, false>(v, true, dhbuf + t * BH, gates + (t - 1) * 4 * BH, cbuf + (t - 1) * BH, cbuf + t * BH,
                                   dcbuf + t * BH, dgates + (t - 1) * 4 * BH, t > 1 ? dcbuf + (t - 1) * BH : nullptr, dc_outside, H, row, col0);
        // (red is rewritten only behind the next barrier's workgroup barrier)
    }
}

extern "C" int capmi_lstm_seq_supported(int B, int H, int T, int dtype) {
    if (B < 1 || B > 64 || T < 1) return 0;
    if (dtype == CAPMI_BF16) return (H == 256 || H == 384 || H == 512 || H == 768 || H == 1024) ? 1 : 0;   // a wave's recurrent weights fit its registers
    if (dtype == CAPMI_F32) return H == 256 ? 1 : 0;
    return 0;
}
static int seq_sync_reset(void* sync, hipStream_t st, const char* who) {
    if (t_probe) return 0;
    hipError_t e = hipMemsetAsync(sync, 0, 8, st);          // arrival counter + this launch's gave-up flag; word 2 (sticky) stays
    if (e != hipSuccess) {
        capmi_set_error("%s: hipMemsetAsync: %s", who, hipGetErrorString(e));
        return 1;
    }
    return 0;
}
// CAPMI_LSTM_SEQ=2: agent-scope release / acquire fences around the arrival counter as well.  The fence-free hand-off
// (write-through stores + sc1 loads) is outside the HIP memory model: it rests on gfx950's sc1 semantics
// (MI355X_MICROARCH.md, inter-workgroup visibility), so it is the default ONLY where the current device reports gfx950;
// anywhere else the fences stay in.
static int seq_fences() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CAPMI_LSTM_SEQ");
        v = (e && e[0] == '2') ? 1 : 0;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) v = 1;
    }
    return v;
}
extern "C" int capmi_lstm_seq_fwd(void* hbuf, const void* wh, int ldw, void* gates, void* cbuf, int B, int H, int T, void* sync, int dtype, void* stream) {
    CAPMI_CHECK(hbuf && wh && gates && cbuf && sync, "capmi_lstm_seq_fwd: null pointer");
    CAPMI_CHECK(capmi_lstm_seq_supported(B, H, T, dtype), "capmi_lstm_seq_fwd: B=%d H=%d T=%d dtype=%d outside the persistent kernel", B, H, T, dtype);
    const int vec = dtype == CAPMI_F32 ? 4 : 8;
    CAPMI_CHECK(ldw % vec == 0, "capmi_lstm_seq_fwd: ldw=%d must be a multiple of %d", ldw, vec);
    CAPMI_CHECK((long long)(T + 1) * B * H * 4 < (1ll << 32), "capmi_lstm_seq_fwd: state buffer beyond 4 GiB");
    hipStream_t st = (hipStream_t)stream;
    if (seq_sync_reset(sync, st, "capmi_lstm_seq_fwd")) return 1;
    const int fences = seq_fences();
#define CAPMI_SEQ_FWD(TT, KS_) CAPMI_KLAUNCH((lstm_seq_fwd_kernel<TT, KS_>), dim3(H / 8), dim3(256), 0, st, (TT*)hbuf, (const TT*)wh, ldw, (TT*)gates, (TT*)cbuf, B, H, T, (unsigned*)sync, fences)
    if (dtype == CAPMI_BF16) {
        if (H == 256) CAPMI_SEQ_FWD(bf16, 2);
        else if (H == 384) CAPMI_SEQ_FWD(bf16, 3);
        else if (H == 512) CAPMI_SEQ_FWD(bf16, 4);
        else if (H == 768) CAPMI_SEQ_FWD(bf16, 6);
        else CAPMI_SEQ_FWD(bf16, 8);
    } else {
        CAPMI_SEQ_FWD(float, 2);
    }
#undef CAPMI_SEQ_FWD
    CAPMI_LAUNCH_CHECK("capmi_lstm_seq_fwd");
    return 0;
}
extern "C" int capmi_lstm_seq_bwd(const void* gates, const void* cbuf, const void* whT, int ldwT, const void* dhbuf, void* dcbuf, void* dgates,
                                  int dc_outside, int B, int H, int T, void* sync, int dtype, void* stream) {
    CAPMI_CHECK(gates && cbuf && whT && dhbuf && dcbuf && dgates && sync, "capmi_lstm_seq_bwd: null pointer");
    CAPMI_CHECK(capmi_lstm_seq_supported(B, H, T, dtype), "capmi_lstm_seq_bwd: B=%d H=%d T=%d dtype=%d outside the persistent kernel", B, H, T, dtype);
    const int vec = dtype == CAPMI_F32 ? 4 : 8;
    CAPMI_CHECK(ldwT % vec == 0, "capmi_lstm_seq_bwd: ldwT=%d must be a multiple of %d", ldwT, vec);
    CAPMI_CHECK((long long)T * B * 4 * H * 4 < (1ll << 32), "capmi_lstm_seq_bwd: gate-gradient buffer beyond 4 GiB");
    hipStream_t st = (hipStream_t)stream;
    if (seq_sync_reset(sync, st, "capmi_lstm_seq_bwd")) return 1;
    const int fences = seq_fences();
    const dim3 grid(H / 32, (B + 15) / 16);
#define CAPMI_SEQ_BWD(TT, KS_, NW_) CAPMI_KLAUNCH((lstm_seq_bwd_kernel<TT, KS_, NW_>), grid, dim3(64 * NW_), 0, st, (const TT*)gates, (const TT*)cbuf, (const TT*)whT, ldwT, (const TT*)dhbuf, (TT*)dcbuf, (TT*)dgates, dc_outside, B, H, T, (unsigned*)sync, fences)
    if (dtype == CAPMI_BF16) {
        if (H == 256) CAPMI_SEQ_BWD(bf16, 8, 4);
        else if (H == 384) CAPMI_SEQ_BWD(bf16, 12, 4);
        else if (H == 512) CAPMI_SEQ_BWD(bf16, 16, 4);
        else if (H == 768) CAPMI_SEQ_BWD(bf16, 12, 8);          // eight waves: a wave's weights stay within 128 registers
        else CAPMI_SEQ_BWD(bf16, 16, 8);
    } else {
        CAPMI_SEQ_BWD(float, 8, 4);
    }
#undef CAPMI_SEQ_BWD
    CAPMI_LAUNCH_CHECK("capmi_lstm_seq_bwd");
    return 0;
}

extern "C" int capmi_lstm_step_supported(int B, int H, int dtype) {
    return (B >= 1 && B <= 64 && H >= 256 && H % 128 == 0 && (dtype == CAPMI_BF16 || dtype == CAPMI_F32)) ? 1 : 0;
}
extern "C" int capmi_lstm_step_fwd(const void* h_prev, const void* wh, int ldw, void* gates, const void* c_prev, void* h, void* c,
                                   int B, int H, int dtype, void* stream) {
    CAPMI_CHECK(h_prev && wh && gates && h && c, "capmi_lstm_step_fwd: null pointer");
    CAPMI_CHECK(capmi_lstm_step_supported(B, H, dtype), "capmi_lstm_step_fwd: B=%d H=%d outside the fused kernel (B <= 64, H %% 128 == 0, H >= 256)", B, H);
    const int vec = dtype == CAPMI_F32 ? 4 : 8;
    CAPMI_CHECK(ldw % vec == 0, "capmi_lstm_step_fwd: ldw=%d must be a multiple of %d", ldw, vec);
    if (dtype == CAPMI_BF16)
        CAPMI_KLAUNCH(lstm_step_fwd_kernel<bf16>, dim3(H / 8), dim3(256), 0, (hipStream_t)stream, (const bf16*)h_prev, (const bf16*)wh, ldw,
                           (bf16*)gates, (const bf16*)c_prev, (bf16*)h, (bf16*)c, B, H);
    else
        CAPMI_KLAUNCH(lstm_step_fwd_kernel<float>, dim3(H / 8), dim3(256), 0, (hipStream_t)stream, (const float*)h_prev, (const float*)wh, ldw,
                           (float*)gates, (const float*)c_prev, (float*)h, (float*)c, B, H);
    CAPMI_LAUNCH_CHECK("capmi_lstm_step_fwd");
    return 0;
}
extern "C" int capmi_lstm_step_bwd(const void* dgates_t, const void* whT, int ldwT, const void* dh_in, const void* gates_prev,
                                   const void* c_prev, const void* c, const void* dc_in, void* dgates_prev, void* dc_prev,
                                   int dc_prev_accumulate, int B, int H, int dtype, void* stream) {
    CAPMI_CHECK(dgates_t && whT && gates_prev && c && dgates_prev, "capmi_lstm_step_bwd: null pointer");
    CAPMI_CHECK(capmi_lstm_step_supported(B, H, dtype), "capmi_lstm_step_bwd: B=%d H=%d outside the fused kernel", B, H);
    const int vec = dtype == CAPMI_F32 ? 4 : 8;
    CAPMI_CHECK(ldwT % vec == 0, "capmi_lstm_step_bwd: ldwT=%d must be a multiple of %d", ldwT, vec);
    if (dtype == CAPMI_BF16)
        CAPMI_KLAUNCH(lstm_step_bwd_kernel<bf16>, dim3(H / 32), dim3(256), 0, (hipStream_t)stream, (const bf16*)dgates_t, (const bf16*)whT, ldwT,
                           (const bf16*)dh_in, (const bf16*)gates_prev, (const bf16*)c_prev, (const bf16*)c, (const bf16*)dc_in, (bf16*)dgates_prev,
                           (bf16*)dc_prev, dc_prev_accumulate, B, H);
    else
        CAPMI_KLAUNCH(lstm_step_bwd_kernel<float>, dim3(H / 32), dim3(256), 0, (hipStream_t)stream, (const float*)dgates_t, (const float*)whT, ldwT,
                           (const float*)dh_in, (const float*)gates_prev, (const float*)c_prev, (const float*)c, (const float*)dc_in, (float*)dgates_prev,
                           (float*)dc_prev, dc_prev_accumulate, B, H);
    CAPMI_LAUNCH_CHECK("capmi_lstm_step_bwd");
    return 0;
}

// The LDS-DMA kernel families store whole runs only (nt_epilogue DENSE): N and every row pitch of the epilogue a multiple of 8.
static bool nt_dense(const IGemmArgs& a) {
    return a.N % 8 == 0 && a.ldy % 8 == 0 && (!a.addend || a.ld_addend % 8 == 0) && (!a.dact || a.ld_saved % 8 == 0);
}

struct NtCfg { int bm, bn, wmw; };
static NtCfg nt_cfg(int M, int N, int K, int dtype) {
    const bool wide = N > 64;
    if (dtype == CAPMI_BF16) {
        if (wide) {
            // 128-row tiles from K = 64 on (512 until the epilogues shrank, lesson 54: with ~1 000 instructions of prologue + epilogue per
            // wave the K <= 256 layers are bound by instruction issue, and a 128-row tile pays the prologue and the statistics once
            // per 128 rows: train step 8.31 -> 8.26 ms with 128; 64 leaves the train step where it is and takes the beam-5 decode at
            // batch 128 from 21 300 to 22 100 captions/s).  CAPMI_NT_BIGK overrides.
            static const int big_k = getenv("CAPMI_NT_BIGK") ? atoi(getenv("CAPMI_NT_BIGK")) : 64;
            static const int big_tiles = getenv("CAPMI_NT_BIGTILES") ? atoi(getenv("CAPMI_NT_BIGTILES")) : 384;      // experiment knob
            const bool big = K >= big_k && (int64_t)cdiv(M, 128) * cdiv(N, 128) >= big_tiles;   // LDS-DMA pipeline kernel: full grid
            const int64_t t128 = (int64_t)cdiv(M, 128) * cdiv(N, 128);
            if (!big && K >= 1024 && t128 >= 160 && t128 <= 256) return NtCfg{128, 128, 4};      // one round of 128x128 tiles, two k-groups each
            if (!big && (int64_t)cdiv(M, 64) * cdiv(N, 128) < 256) return NtCfg{64, 64, 5};      // under-filled grid: 64x64 LDS-DMA tiles (wmw 5 = marker)
            return NtCfg{big ? 128 : 64, 128, 4};
        }
        // tall grids of narrow outputs (the 64-channel 56 x 56 layers): 128 x 64 tiles pay the prologue, the statistics and (3 x 3)
        // the halo once per 128 rows (wmw 6 = marker).  CAPMI_NT_TALL64 = minimum number of 128-row blocks, 0 = never.
        // (8.235 -> 8.195 ms per step at cfg 2 with 1 024; lesson 54.)
        static const int tall64 = getenv("CAPMI_NT_TALL64") ? atoi(getenv("CAPMI_NT_TALL64")) : 1024;
        if (N >= 32 && tall64 > 0 && cdiv(M, 128) >= tall64) return NtCfg{128, 64, 6};
        if (N >= 32) return NtCfg{64, 64, 5};          // LDS-DMA 64x64 tiles
        const bool tall = cdiv(M, 128) >= 512;
        return NtCfg{tall ? 128 : 64, 64, 4};
    }
    const bool tall = (int64_t)cdiv(M, 128) * cdiv(N, 64) >= 256;
    return wide ? NtCfg{64, 128, 2} : NtCfg{tall ? 128 : 64, 64, 2};
}

extern "C" int capmi_igemm_nt_stats_part_rows(int M, int N, int K, int dtype) {
    return nt_cfg(M, N, K, dtype).bm;
}

template <typename T, int BM, int BN, int WMW>
static int launch_nt(const IGemmArgs& a, hipStream_t st) {
    int64_t tiles = (int64_t)cdiv(a.M, BM) * cdiv(a.N, BN);
    if (tiles <= 0) return 0;
    CAPMI_CHECK(tiles < (1ll << 31), "capmi_igemm_nt: grid too large");
    if (a.nred) CAPMI_KLAUNCH((igemm_nt_kernel<T, BM, BN, WMW, true>), dim3((unsigned)tiles), dim3(256), 0, st, a);
    else CAPMI_KLAUNCH((igemm_nt_kernel<T, BM, BN, WMW>), dim3((unsigned)tiles), dim3(256), 0, st, a);
    CAPMI_LAUNCH_CHECK("capmi_igemm_nt");
    return 0;
}

struct BnRedTarget { const void* x; const float* mean; const float* invstd; float* ws; };

static bool nt_uses_skinny(const capmi_conv_geom* g, int M, int K, bool stats, int dtype) {
    const bool plain = g->kh == 1 && g->kw == 1 && g->sd == 1 && g->up == 1 && g->pad == 0 && g->Hi == 1 && g->Wi == 1 && g->Ho == 1 && g->Wo == 1;
    return plain && g->os <= 1 && !stats && M <= 64 && K % 128 == 0 && K >= 256 && (dtype == CAPMI_BF16 || dtype == CAPMI_F32);
}

static int nt_prepare(IGemmArgs& a, const void* x, const void* w, void* y, const capmi_conv_geom* g,
                      int N, int ldw, int ldy, const float* bias, const void* addend,
                      int ld_addend, const void* ysaved, int ld_saved, float* stats,
                      int act, int dact, int out_f32, int nred, const BnRedTarget* red, int dtype) {
    CAPMI_CHECK(x && w && y && g, "capmi_igemm_nt: null pointer");
    const int vec = dtype == CAPMI_F32 ? 4 : 8;
    CAPMI_CHECK(g->Cin % vec == 0 && g->ldx % vec == 0 && ldw % vec == 0,
                "capmi_igemm_nt: Cin=%d ldx=%d ldw=%d must be multiples of %d", g->Cin, g->ldx, ldw, vec);
    CAPMI_CHECK(g->up >= 1 && (g->up & (g->up - 1)) == 0 && g->sd >= 1 && g->kh >= 1 && g->kw >= 1,
                "capmi_igemm_nt: bad geometry (up must be a power of two)");
    CAPMI_CHECK(!dact || ysaved, "capmi_igemm_nt: dact needs ysaved");
    CAPMI_CHECK(!(stats && addend), "capmi_igemm_nt: fused statistics and addend are mutually exclusive");
    CAPMI_CHECK(g->os <= 1 || (!stats && g->Hof > 0 && g->Wof > 0 && (g->Ho - 1) * g->os + g->oh0 < g->Hof && (g->Wo - 1) * g->os + g->ow0 < g->Wof),
                "capmi_igemm_nt: bad output-scatter geometry");
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.bn_mean = nullptr; a.bn_a = nullptr; a.addend = addend; a.ysaved = ysaved; a.stats = stats;
    a.in_mean = nullptr; a.in_a = nullptr; a.in_off = nullptr; a.in_act = 0; a.ksplit = 1; a.kper = 0;
    a.fin_cnt = nullptr; a.fin_merged = nullptr; a.fin_k = 0; a.fin_groups = 1;
    a.M = g->B * g->Ho * g->Wo; a.N = N; a.K = g->kh * g->kw * g->Cin;
    a.ldw = ldw; a.ldy = ldy; a.ld_addend = ld_addend; a.ld_saved = ld_saved;
    a.g = *g; a.act = act; a.dact = dact; a.out_f32 = out_f32;
    if (dact & CAPMI_DACT_BITMASK) {
        const int base = dact & (CAPMI_DACT_BITMASK - 1);
        CAPMI_CHECK(dtype == CAPMI_BF16 && (base == CAPMI_ACT_RELU || base == CAPMI_ACT_RELU6) && N % 8 == 0 && ld_saved % 8 == 0 && ldy % 8 == 0 &&
                        (!addend || ld_addend % 8 == 0) && !nred && !bias && !act && !out_f32 && !stats && !(g->Hi == 1 && g->Wi == 1) && N >= 32,
                    "capmi_igemm_nt: CAPMI_DACT_BITMASK is for bf16 convolution data gradients (relu / relu6; N, ldy, ld_saved, ld_addend multiples of 8; no bias / activation / statistics / f32 output)");
    }
    a.nred = nred;
    a.red_mode = 0;
    a.stat_sums = 0;
    a.stat_shift = nullptr;
    for (int q = 0; q < 2; ++q) {
        a.rx[q] = q < nred ? red[q].x : nullptr; a.rmean[q] = q < nred ? red[q].mean : nullptr;
        a.rinv[q] = q < nred ? red[q].invstd : nullptr; a.rws[q] = q < nred ? red[q].ws : nullptr;
    }
    CAPMI_CHECK((long long)(a.M + 256) * g->Ho * g->Wo < (1ll << 40), "capmi_igemm_nt: M * Ho*Wo outside the fast-division range");
    a.fd_hw = fast_div(g->Ho * g->Wo); a.fd_w = fast_div(g->Wo);
#ifdef CAPMI_STAMPS
    a.stamps = g_stamp_buffer;
#endif
    CAPMI_CHECK(ldw >= a.K, "capmi_igemm_nt: ldw=%d < K=%d", ldw, a.K);
    return 0;
}

template <int BM, int BN, int EPI>
static int launch_glds_epi(const IGemmArgs& a, const capmi_conv_geom* g, bool lin, bool conv1, hipStream_t st) {
    const int64_t tiles = (int64_t)cdiv(a.M, BM) * cdiv(a.N, BN);
    CAPMI_CHECK(tiles < (1ll << 31), "capmi_igemm_nt: grid too large");
    const dim3 grid((unsigned)tiles);
    if constexpr (BM == 128) {
        if ((!a.nred || EPI == 7) && tiles <= 256 && a.K >= 1024 && (lin || (conv1 && g->Cin >= 64))) {
            if (lin) CAPMI_KLAUNCH((igemm_nt_glds_kg_kernel<BM, BN, 3, 1, 2, EPI>), grid, dim3(512), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_glds_kg_kernel<BM, BN, 3, 2, 2, EPI>), grid, dim3(512), 0, st, a);
            CAPMI_LAUNCH_CHECK("capmi_igemm_nt(glds 128, k-groups)");
            return 0;
        }
    }
    if constexpr (BM == 64) {
        // deep K on an under-filled grid: 2 or 4 k-groups per workgroup (more waves, same tile)
        const bool k2 = (!a.nred || EPI == 7) && tiles < 512 && a.K >= 1024 && (lin || (conv1 && g->Cin >= 64));
        const bool k4 = k2 && tiles <= 256 && a.K >= 2048 && (lin || g->Cin >= 128);
        if (k2) {
            if (k4 && lin) CAPMI_KLAUNCH((igemm_nt_glds_kg_kernel<BM, BN, 3, 1, 4, EPI>), grid, dim3(1024), 0, st, a);
            else if (k4) CAPMI_KLAUNCH((igemm_nt_glds_kg_kernel<BM, BN, 3, 2, 4, EPI>), grid, dim3(1024), 0, st, a);
            else if (lin) CAPMI_KLAUNCH((igemm_nt_glds_kg_kernel<BM, BN, 3, 1, 2, EPI>), grid, dim3(512), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_glds_kg_kernel<BM, BN, 3, 2, 2, EPI>), grid, dim3(512), 0, st, a);
            CAPMI_LAUNCH_CHECK("capmi_igemm_nt(glds, k-groups)");
            return 0;
        }
    }
    if constexpr (EPI == 0 || EPI == 4) {       // (the fused batch-norm backward sums ride on data gradients only)
        if (a.nred) {
            CAPMI_KLAUNCH((igemm_nt_glds_kernel<BM, BN, 3, true, 0, EPI>), grid, dim3(256), 0, st, a);
            CAPMI_LAUNCH_CHECK("capmi_igemm_nt(glds)");
            return 0;
        }
    }
    if (lin) CAPMI_KLAUNCH((igemm_nt_glds_kernel<BM, BN, 3, false, 1, EPI>), grid, dim3(256), 0, st, a);
    else if (conv1) CAPMI_KLAUNCH((igemm_nt_glds_kernel<BM, BN, 3, false, 2, EPI>), grid, dim3(256), 0, st, a);
    else CAPMI_KLAUNCH((igemm_nt_glds_kernel<BM, BN, 3, false, 0, EPI>), grid, dim3(256), 0, st, a);
    CAPMI_LAUNCH_CHECK("capmi_igemm_nt(glds)");
    return 0;
}

// The epilogue class of a launch (nt_epilogue EPI): 1 = a training convolution / its data gradient.
static bool nt_conv_class(const IGemmArgs& a) {
    return !capmi_general_epilogue() && !a.bias && !a.bn_a && a.act == CAPMI_ACT_NONE && !a.out_f32 && a.ksplit <= 1 &&
           (a.dact == CAPMI_ACT_NONE || a.dact == CAPMI_ACT_RELU || a.dact == CAPMI_ACT_RELU6);
}
// the data-gradient form with its mask as bits (the flag forces the class: no other epilogue reads bits)
static bool nt_bits_class(const IGemmArgs& a) { return (a.dact & CAPMI_DACT_BITMASK) != 0; }
// ... and with the batch-norm backward sums of the completed tensor's layer in the same epilogue (capmi_igemm_nt_bnsum)
static bool nt_bnsum_class(const IGemmArgs& a) { return nt_bits_class(a) && a.red_mode != 0; }
// conv class: 1 = the forward form (nothing but stores and statistics), 4 = the data-gradient form (no statistics)
static bool nt_conv_fwd(const IGemmArgs& a) { return !a.addend && !a.dact && a.g.os <= 1 && !a.nred; }
// 2 = a fully connected layer of the decoder / its data gradient (bias, addend; tanh or nothing on either side)
static bool nt_fc_class(const IGemmArgs& a) {
    return !capmi_general_epilogue() && !a.bn_a && !a.out_f32 && a.ksplit <= 1 && !a.stats && !a.nred && (a.act == CAPMI_ACT_NONE || a.act == CAPMI_ACT_TANH) &&
           (a.dact == CAPMI_ACT_NONE || a.dact == CAPMI_ACT_TANH);
}
// 3 = a convolution of the inference graph (capmi_igemm_nt_bn)
static bool nt_inf_class(const IGemmArgs& a) {
    return !capmi_general_epilogue() && a.bn_a && !a.out_f32 && a.ksplit <= 1 && !a.stats && !a.nred && !a.dact &&
           (a.act == CAPMI_ACT_NONE || a.act == CAPMI_ACT_RELU || a.act == CAPMI_ACT_RELU6);
}
// 5 = a plain product with an f32 output (+ bias): the logits of the vocabulary projection
static bool nt_f32_class(const IGemmArgs& a) {
    return !capmi_general_epilogue() && a.out_f32 && !a.bn_a && a.ksplit <= 1 && !a.stats && !a.nred && !a.addend && a.act == CAPMI_ACT_NONE && a.dact == CAPMI_ACT_NONE;
}
template <int BM, int BN>
static int launch_glds(const IGemmArgs& a, const capmi_conv_geom* g, bool lin, bool conv1, hipStream_t st) {
    if (nt_bnsum_class(a)) return launch_glds_epi<BM, BN, 7>(a, g, lin, conv1, st);
    if (nt_bits_class(a)) return launch_glds_epi<BM, BN, 6>(a, g, lin, conv1, st);
    if (nt_conv_class(a) && nt_conv_fwd(a) && a.stat_sums) return launch_glds_epi<BM, BN, 8>(a, g, lin, conv1, st);
    if (nt_conv_class(a)) return nt_conv_fwd(a) ? launch_glds_epi<BM, BN, 1>(a, g, lin, conv1, st) : launch_glds_epi<BM, BN, 4>(a, g, lin, conv1, st);
    if (nt_fc_class(a)) return launch_glds_epi<BM, BN, 2>(a, g, lin, conv1, st);
    if (nt_inf_class(a)) return launch_glds_epi<BM, BN, 3>(a, g, lin, conv1, st);
    if (nt_f32_class(a)) return launch_glds_epi<BM, BN, 5>(a, g, lin, conv1, st);
    return launch_glds_epi<BM, BN, 0>(a, g, lin, conv1, st);
}

// 3x3 / stride 1 / pad 1 "same" convolutions (forward, and the data gradient of one) on rows of <= 56 pixels, channels
// in chunks of 32: the halo-staged kernel, ON by default (CAPMI_HALO3=0 restores per-tap staging; 2 / 3: forward /
// data-gradient calls only).  History (DESIGN.md lesson 29): alone it was bit-exact from the start, inside the two-lane
// train step its 56x56 launches came out with a few 16-row strips wrong, differently every run.  Root cause: an LDS
// write-after-read the source did not show -- the tap loop is unrolled and s_barrier is IntrNoMem, so the scheduler sank
// the MFMAs of step s (and with them the lgkmcnt wait that retires the step's fragment reads) below the barrier and the
// DMA issue of step s + 1, i.e. the ring slot was being refilled while its reads were not known to have returned.  Fix: the
// sched_barrier(0) behind the last MFMA of every tap (see the kernel).  The fix depends on the compiler keeping the
// lgkmcnt wait in FRONT of s_barrier, so two gates must stay green on every toolchain bump: tools/lds_war_audit.py (static
// screen of the device assembly for an LDS-DMA issue behind a barrier with unretired ds_reads; tests/test_host.py runs
// it when hipcc is present) and the in-situ check of all 53 / 104 convolutions inside the two-lane step
// (tests/test_gpu_fullsize.py).
static bool nt_halo3_ok(const IGemmArgs& a, const capmi_conv_geom* g, int nred) {
    static const int mode = getenv("CAPMI_HALO3") ? atoi(getenv("CAPMI_HALO3")) : 1;      // 0: per-tap staging; 2 / 3: forward / data-gradient calls only
    if (mode == 0 || (mode == 2 && !a.stats) || (mode == 3 && a.stats)) return false;
    // experiment knob: grids of at most this many 128 x 128 tiles take the k-group LDS-DMA kernel instead (8 waves per workgroup)
    static const int min_tiles = getenv("CAPMI_HALO3_MINTILES") ? atoi(getenv("CAPMI_HALO3_MINTILES")) : 0;
    if (min_tiles > 0 && (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) <= min_tiles) return false;
    return (!nred || a.red_mode != 0) && g->kh == 3 && g->kw == 3 && g->sd == 1 && g->up == 1 && g->pad == 1 && g->Hi == g->Ho && g->Wi == g->Wo &&
           g->os <= 1 && g->Wi <= 56 && g->Wi * g->Hi > 64 && g->Cin % 32 == 0 && a.K == 9 * g->Cin && a.N >= 32;       // (7 x 7: the k-group kernel is faster)
}

// Which kernel family applies batch norm + ReLU in the operand path for this convolution: 1 = the halo-staged 3x3 kernel
// (transform of the halo tile in LDS), 2 = the LDS-DMA kernel on a 1x1 convolution (transform of the A fragments), 0 = none.
// Same tile selection as the plain call, so the statistics parts (capmi_igemm_nt_stats_part_rows) keep their height.
static int nt_inbn_kind(const IGemmArgs& a, const capmi_conv_geom* g, const NtCfg& c, int nred, int dtype) {
    if (dtype != CAPMI_BF16 || nred || g->Cin > INBN_KMAX || g->Cin % 32 != 0 || g->ldx != g->Cin || !nt_dense(a)) return 0;
    if (nt_halo3_ok(a, g, nred)) return 1;
    const bool lin = g->kh == 1 && g->kw == 1 && g->up == 1 && g->pad == 0 && (g->Ho - 1) * g->sd < g->Hi && (g->Wo - 1) * g->sd < g->Wi;
    const bool glds = c.wmw == 5 || c.bn == 128;
    if (lin && glds && g->os <= 1 && a.K == g->Cin) return 2;
    return 0;
}

static int nt_dispatch(const IGemmArgs& a, const capmi_conv_geom* g, int N, float* stats, int nred, int dtype, hipStream_t st) {
    if (nt_uses_skinny(g, a.M, a.K, stats != nullptr, dtype)) {
        CAPMI_CHECK(nred == 0, "capmi_igemm_nt_bnred: not available for M <= 64 plain products (see capmi_igemm_nt_bnred_part_rows)");
        // decoder recurrence and other M <= 64 products: skinny kernel (64x32 tiles, K split over waves)
        if (dtype == CAPMI_BF16) CAPMI_KLAUNCH(igemm_nt_skinny_kernel<bf16>, dim3(cdiv(N, 32)), dim3(256), 0, st, a);
        else CAPMI_KLAUNCH(igemm_nt_skinny_kernel<float>, dim3(cdiv(N, 32)), dim3(256), 0, st, a);
        CAPMI_LAUNCH_CHECK("capmi_igemm_nt(skinny)");
        return 0;
    }
    const NtCfg c = nt_cfg(a.M, N, a.K, dtype);
    if (dtype == CAPMI_BF16 && !nt_dense(a)) {
        // ragged N or an odd pitch: the register-staged kernel (element-wise stores), same row-block height as the tile it replaces
        // (capmi_igemm_nt_stats_part_rows stays valid)
        CAPMI_CHECK(!a.in_a, "capmi_igemm_nt_bnact: N, ldy must be multiples of 8");
        if (c.bm == 128) return launch_nt<bf16, 128, 64, 4>(a, st);
        return launch_nt<bf16, 64, 64, 4>(a, st);
    }
    if (a.in_a) {               // operand-path batch norm (capmi_igemm_nt_bnact): the two kernel families that carry it
        const int kind = nt_inbn_kind(a, g, c, nred, dtype);
        CAPMI_CHECK(kind != 0, "capmi_igemm_nt_bnact: this convolution has no operand-path batch-norm kernel (capmi_igemm_nt_bnact_supported)");
        const int bn = kind == 1 ? (N <= 64 ? 64 : 128) : (c.wmw == 5 ? 64 : 128);
        const int bm = kind == 1 ? c.bm : (c.wmw == 5 ? 64 : c.bm);
        const int64_t tiles = (int64_t)cdiv(a.M, bm) * cdiv(N, bn);
        CAPMI_CHECK(tiles < (1ll << 31), "capmi_igemm_nt_bnact: grid too large");
        const dim3 grid((unsigned)tiles);
        if (kind == 1) {
            if (bm == 128 && bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 128, true, 1>), grid, dim3(256), 0, st, a);
            else if (bm == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 64, true, 1>), grid, dim3(256), 0, st, a);
            else if (bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 128, true, 1>), grid, dim3(256), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 64, true, 1>), grid, dim3(256), 0, st, a);
        } else {
#define CAPMI_INBN_LAUNCH(BM_, BN_)                                                                                              \
            do {                                                                                                                 \
                if (a.K <= 128) CAPMI_KLAUNCH((igemm_nt_glds_inbn_kernel<BM_, BN_, 128>), grid, dim3(256), 0, st, a);       \
                else if (a.K <= 256) CAPMI_KLAUNCH((igemm_nt_glds_inbn_kernel<BM_, BN_, 256>), grid, dim3(256), 0, st, a);  \
                else CAPMI_KLAUNCH((igemm_nt_glds_inbn_kernel<BM_, BN_, INBN_KMAX>), grid, dim3(256), 0, st, a);            \
            } while (0)
            if (bm == 128) CAPMI_INBN_LAUNCH(128, 128);
            else if (bn == 128) CAPMI_INBN_LAUNCH(64, 128);
            else CAPMI_INBN_LAUNCH(64, 64);
#undef CAPMI_INBN_LAUNCH
        }
        CAPMI_LAUNCH_CHECK("capmi_igemm_nt_bnact");
        return 0;
    }
    if (dtype == CAPMI_BF16 && nt_halo3_ok(a, g, nred)) {
        // same row-block height as the kernel it replaces: capmi_igemm_nt_stats_part_rows (one statistics part per BM rows) stays valid
        const int bn = N <= 64 ? 64 : 128;
        const int64_t tiles = (int64_t)cdiv(a.M, c.bm) * cdiv(N, bn);
        CAPMI_CHECK(tiles < (1ll << 31), "capmi_igemm_nt: grid too large");
        const dim3 grid((unsigned)tiles);
        if (nt_bnsum_class(a)) {
            if (c.bm == 128 && bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 128, false, 7>), grid, dim3(256), 0, st, a);
            else if (c.bm == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 64, false, 7>), grid, dim3(256), 0, st, a);
            else if (bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 128, false, 7>), grid, dim3(256), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 64, false, 7>), grid, dim3(256), 0, st, a);
        } else if (nt_bits_class(a)) {
            if (c.bm == 128 && bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 128, false, 6>), grid, dim3(256), 0, st, a);
            else if (c.bm == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 64, false, 6>), grid, dim3(256), 0, st, a);
            else if (bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 128, false, 6>), grid, dim3(256), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 64, false, 6>), grid, dim3(256), 0, st, a);
        } else if (nt_conv_class(a) && nt_conv_fwd(a) && a.stat_sums) {
            if (c.bm == 128 && bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 128, false, 8>), grid, dim3(256), 0, st, a);
            else if (c.bm == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 64, false, 8>), grid, dim3(256), 0, st, a);
            else if (bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 128, false, 8>), grid, dim3(256), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 64, false, 8>), grid, dim3(256), 0, st, a);
        } else if (nt_conv_class(a) && nt_conv_fwd(a)) {
            if (c.bm == 128 && bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 128, false, 1>), grid, dim3(256), 0, st, a);
            else if (c.bm == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 64, false, 1>), grid, dim3(256), 0, st, a);
            else if (bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 128, false, 1>), grid, dim3(256), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 64, false, 1>), grid, dim3(256), 0, st, a);
        } else if (nt_conv_class(a) && !a.stats) {
            if (c.bm == 128 && bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 128, false, 4>), grid, dim3(256), 0, st, a);
            else if (c.bm == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 64, false, 4>), grid, dim3(256), 0, st, a);
            else if (bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 128, false, 4>), grid, dim3(256), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 64, false, 4>), grid, dim3(256), 0, st, a);
        } else if (nt_inf_class(a)) {
            if (c.bm == 128 && bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 128, false, 3>), grid, dim3(256), 0, st, a);
            else if (c.bm == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 64, false, 3>), grid, dim3(256), 0, st, a);
            else if (bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 128, false, 3>), grid, dim3(256), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 64, false, 3>), grid, dim3(256), 0, st, a);
        } else {
            if (c.bm == 128 && bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 128>), grid, dim3(256), 0, st, a);
            else if (c.bm == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<128, 64>), grid, dim3(256), 0, st, a);
            else if (bn == 128) CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 128>), grid, dim3(256), 0, st, a);
            else CAPMI_KLAUNCH((igemm_nt_halo3_kernel<64, 64>), grid, dim3(256), 0, st, a);
        }
        CAPMI_LAUNCH_CHECK("capmi_igemm_nt(halo 3x3)");
        return 0;
    }
    if (dtype == CAPMI_BF16) {
        const bool lin = g->kh == 1 && g->kw == 1 && g->up == 1 && g->pad == 0 && (g->Ho - 1) * g->sd < g->Hi && (g->Wo - 1) * g->sd < g->Wi;
        const bool conv1 = !lin && g->up == 1 && g->Cin >= 32;
        if (c.wmw == 5) return launch_glds<64, 64>(a, g, lin, conv1, st);           // 64x64 LDS-DMA tiles
        if (c.wmw == 6) return launch_glds<128, 64>(a, g, lin, conv1, st);          // 128x64 LDS-DMA tiles
        if (c.bn == 128 && c.bm == 128) return launch_glds<128, 128>(a, g, lin, conv1, st);
        if (c.bn == 128) return launch_glds<64, 128>(a, g, lin, conv1, st);
        if (c.bm == 128 && c.bn == 64) return launch_nt<bf16, 128, 64, 4>(a, st);
        return launch_nt<bf16, 64, 64, 4>(a, st);
    } else if (dtype == CAPMI_F32) {
        if (c.bm == 64 && c.bn == 128) return launch_nt<float, 64, 128, 2>(a, st);
        if (c.bm == 128 && c.bn == 64) return launch_nt<float, 128, 64, 2>(a, st);
        return launch_nt<float, 64, 64, 2>(a, st);
    }
    capmi_set_error("capmi_igemm_nt: bad dtype %d", dtype);
    return 1;
}


static int igemm_nt_impl(const void* x, const void* w, void* y, const capmi_conv_geom* g,
                         int N, int ldw, int ldy, const float* bias, const void* addend,
                         int ld_addend, const void* ysaved, int ld_saved, float* stats,
                         int act, int dact, int out_f32, int nred, const BnRedTarget* red, int dtype, void* stream) {
    IGemmArgs a;
    if (nt_prepare(a, x, w, y, g, N, ldw, ldy, bias, addend, ld_addend, ysaved, ld_saved, stats, act, dact, out_f32, nred, red, dtype)) return 1;
    return nt_dispatch(a, g, N, stats, nred, dtype, (hipStream_t)stream);
}

extern "C" int capmi_igemm_nt(const void* x, const void* w, void* y, const capmi_conv_geom* g,
                              int N, int ldw, int ldy, const float* bias, const void* addend,
                              int ld_addend, const void* ysaved, int ld_saved, float* stats,
                              int act, int dact, int out_f32, int dtype, void* stream) {
    return igemm_nt_impl(x, w, y, g, N, ldw, ldy, bias, addend, ld_addend, ysaved, ld_saved, stats, act, dact, out_f32, 0, nullptr, dtype, stream);
}

// ------------------------------------------------------------------ convolution + statistics + finalize in one launch
// Arrival counters of nt_fin_tail: a pool of 128 sets per DEVICE, handed out round-robin per launch and zero again when a
// launch ends (the contract of capmi_bn_finalize's pool, capmi.h: at most 127 later fused launches in flight next to an
// unfinished one; the engine joins its lanes at the end of every step).
constexpr int FIN_SET = 1024;
__device__ unsigned nt_fin_counters[128 * FIN_SET];
static unsigned* next_fin_counters() {
    constexpr int MAXDEV = 64;
    static std::atomic<unsigned> next[MAXDEV];
    static std::atomic<unsigned*> base[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    unsigned* b = base[dev].load(std::memory_order_acquire);
    if (!b) {
        if (hipGetSymbolAddress((void**)&b, HIP_SYMBOL(nt_fin_counters)) != hipSuccess || !b) return nullptr;
        base[dev].store(b, std::memory_order_release);
    }
    return b + (next[dev].fetch_add(1) % 128u) * FIN_SET;
}

/* capmi_igemm_nt with fused statistics (stats != NULL, no bias / addend / activation) followed by capmi_bn_finalize on those
 * statistics -- conv2d -> batch_norm's statistics of MobileNetV2.py:88-121 conv_bn_layer (train mode) -- as ONE launch
 * wherever the convolution's grid is small enough for the last-arriver tail to cost less than the dependent launch it
 * replaces (nt_fin_tail; <= CAPMI_BNFIN_MAX_WGS workgroups, default 1024), and as the two calls otherwise (large grids, f32,
 * captured streams).  Results are bit-identical either way.  stats: the workspace of the two-call form (parts + room for
 * the merged parts, capmi.h). */
extern "C" int capmi_igemm_nt_bnfin(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy, float* stats,
                                    const float* scale, float* run_mean, float* run_var, float momentum, float eps, float* saved_mean,
                                    float* saved_invstd, float* coef_a, int update_running, int dtype, void* stream) {
    CAPMI_CHECK(stats && scale && saved_mean && saved_invstd && coef_a, "capmi_igemm_nt_bnfin: null pointer");
    CAPMI_CHECK(!update_running || (run_mean && run_var), "capmi_igemm_nt_bnfin: running stats missing");
    IGemmArgs a;
    if (nt_prepare(a, x, w, y, g, N, ldw, ldy, nullptr, nullptr, 0, nullptr, 0, stats, 0, 0, 0, 0, nullptr, dtype)) return 1;
    static const int max_wgs = getenv("CAPMI_BNFIN_MAX_WGS") ? atoi(getenv("CAPMI_BNFIN_MAX_WGS")) : 1024;
    const NtCfg c = nt_cfg(a.M, N, a.K, dtype);
    const int part_rows = c.bm;
    bool fused = CAPMI_FIN && dtype == CAPMI_BF16 && max_wgs > 0 && !nt_uses_skinny(g, a.M, a.K, true, dtype);
    int bn_tile = 0;
    if (fused) {
        // the column tile of the kernel nt_dispatch will pick (the counters are per column tile)
        if (nt_halo3_ok(a, g, 0)) bn_tile = N <= 64 ? 64 : 128;
        else if (c.wmw == 5) bn_tile = 64;
        else if (c.bn == 128) bn_tile = 128;
        else bn_tile = 64;
        const int64_t tiles = (int64_t)cdiv(a.M, c.bm) * cdiv(N, bn_tile);
        const int nparts = cdiv(a.M, part_rows);
        const int k = nparts > 2 * CAPMI_BN_MERGE_GROUPS ? cdiv(nparts, N <= 128 ? 2 * CAPMI_BN_MERGE_GROUPS : CAPMI_BN_MERGE_GROUPS) : 0;
        const int groups = k > 0 ? cdiv(nparts, k) : 1;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cap);       // a captured launch would freeze its counter set into the graph
        fused = tiles <= max_wgs && cdiv(N, bn_tile) * (groups + 1) <= FIN_SET && cap == hipStreamCaptureStatusNone;
        if (fused) {
            a.fin_cnt = next_fin_counters();
            fused = a.fin_cnt != nullptr;
            a.fin_k = k;
            a.fin_groups = groups;
            a.fin_merged = stats + (int64_t)nparts * N * 2;
            a.fin_scale = scale; a.fin_run_mean = run_mean; a.fin_run_var = run_var; a.fin_mean = saved_mean; a.fin_invstd = saved_invstd;
            a.fin_a = coef_a; a.fin_momentum = momentum; a.fin_eps = eps; a.fin_update = update_running;
        }
        if (!fused) a.fin_cnt = nullptr;
    }
    if (nt_dispatch(a, g, N, stats, 0, dtype, (hipStream_t)stream)) return 1;
    if (fused || t_probe) return 0;
    return capmi_bn_finalize(stats, part_rows, a.M, N, scale, run_mean, run_var, momentum, eps, saved_mean, saved_invstd, coef_a, update_running, stream);
}

// ------------------------------------------------------------------ split-K over workgroups (deep K, few output tiles)
// y[m][n] = sum_s slab[s][m][n], rounded to T once: 16-byte slab loads, fixed summation order (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void nt_splitk_reduce_kernel(const float* __restrict__ slab, int S, int64_t MN, int N, int ldy, T* y) {
    const int64_t n4 = MN / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 acc = reinterpret_cast<const f32x4*>(slab)[i];
        for (int s = 1; s < S; ++s) acc += reinterpret_cast<const f32x4*>(slab + (int64_t)s * MN)[i];
        const int64_t e = i * 4, row = e / N;
        const int col = (int)(e - row * N);
        T* o = y + row * ldy + col;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = from_f32<T>(acc[k]);
    }
}
static int nt_splitk_plan(int M, int N, int K, int* kper) {      // number of splits (1: not worth it) and the k range of one
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    if (K < 4096 || tiles > 128 || N % 8 != 0) return 1;
    int S = 256 / tiles;
    if (S > cdiv(K, 1024)) S = cdiv(K, 1024);                    // >= 32 k-steps per split
    if (S < 2) return 1;
    const int per = cdiv(cdiv(K, S), 32) * 32;
    *kper = per;
    return cdiv(K, per);
}
extern "C" long long capmi_igemm_nt_splitk_ws_bytes(int M, int N, int K, int dtype) {
    int kper = 0;
    const int S = dtype == CAPMI_BF16 ? nt_splitk_plan(M, N, K, &kper) : 1;
    return S > 1 ? (long long)S * M * N * 4 : 0;
}
/* A plain product y[M][N] = x[M][K] . w[N][K]^T (no bias / activation / statistics) whose reduction is long and whose output
 * is small -- the tied projection's data gradient, [T*B][V] x [V][E] (model_adaAttention_aic.py:25 backward): 40 output
 * tiles cannot fill 256 CUs however many waves work each.  K is split over workgroups into f32 slabs (ws, at least
 * capmi_igemm_nt_splitk_ws_bytes) that a second launch adds up in a fixed order; shapes the split does not pay for (or a
 * missing workspace) run as capmi_igemm_nt. */
extern "C" int capmi_igemm_nt_splitk(const void* x, const void* w, void* y, int M, int K, int ldx, int N, int ldw, int ldy,
                                     float* ws, long long ws_bytes, int dtype, void* stream) {
    const capmi_conv_geom g = {M, 1, 1, K, 1, 1, 1, 1, 1, 1, 0, ldx, 0, 0, 0, 0, 0};
    int kper = 0;
    const int S = dtype == CAPMI_BF16 ? nt_splitk_plan(M, N, K, &kper) : 1;
    if (S <= 1 || !ws || ws_bytes < (long long)S * M * N * 4)
        return igemm_nt_impl(x, w, y, &g, N, ldw, ldy, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, 0, nullptr, dtype, stream);
    IGemmArgs a;
    if (nt_prepare(a, x, w, ws, &g, N, ldw, N, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, 0, 1, 0, nullptr, dtype)) return 1;
    a.ksplit = S;
    a.kper = kper;
    const int64_t tiles = (int64_t)cdiv(M, 128) * cdiv(N, 128);
    hipStream_t st = (hipStream_t)stream;
    CAPMI_KLAUNCH((igemm_nt_glds_kernel<128, 128, 3, false, 1>), dim3((unsigned)(tiles * S)), dim3(256), 0, st, a);
    const int64_t MN = (int64_t)M * N;
    CAPMI_KLAUNCH(nt_splitk_reduce_kernel<bf16>, dim3(ew_grid(MN / 4)), dim3(256), 0, st, ws, S, MN, N, ldy, (bf16*)y);
    CAPMI_LAUNCH_CHECK("capmi_igemm_nt_splitk");
    return 0;
}

/* Convolution + INFERENCE batch norm + residual + activation in one launch (the exported model of infer.py: every
 * batch_norm is_test): y = act(coef_a * (conv - mean) + offset (+ res)), the formula of capmi_bn_apply on the f32
 * accumulator -- the conv output never goes to memory.  mean / coef_a from capmi_bn_inference_coef. */
extern "C" int capmi_igemm_nt_bn(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                                 const float* mean, const float* coef_a, const float* offset, const void* res, int ld_res,
                                 int act, int dtype, void* stream) {
    CAPMI_CHECK(mean && coef_a && offset, "capmi_igemm_nt_bn: null batch-norm vector");
    IGemmArgs a;
    if (nt_prepare(a, x, w, y, g, N, ldw, ldy, offset, res, ld_res, nullptr, 0, nullptr, act, 0, 0, 0, nullptr, dtype)) return 1;
    CAPMI_CHECK(!nt_uses_skinny(g, a.M, a.K, false, dtype), "capmi_igemm_nt_bn: plain products of <= 64 rows are not convolutions");
    a.bn_mean = mean;
    a.bn_a = coef_a;
    return nt_dispatch(a, g, N, nullptr, 0, dtype, (hipStream_t)stream);
}

/* Convolution whose INPUT is the raw output of the producing convolution: act(coef_a * (x - mean) + offset) -- bn_apply's
 * formula and rounding -- is applied in the A-operand path, so the producer's normalised / activated tensor is never
 * materialised for this consumer (IC/model/MobileNetV2.py:88-121: the conv1 -> conv2 -> conv3 links of a unit chain).
 * stats: fused batch statistics of THIS convolution's output, as capmi_igemm_nt. */
extern "C" int capmi_igemm_nt_bnact_supported(const capmi_conv_geom* g, int N, int dtype) {
    if (!g || dtype != CAPMI_BF16) return 0;
    IGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.M = g->B * g->Ho * g->Wo; a.N = N; a.K = g->kh * g->kw * g->Cin;
    a.stats = reinterpret_cast<float*>(1);        // (a forward call: CAPMI_HALO3=2 / 3 look at it)
    if (nt_uses_skinny(g, a.M, a.K, true, dtype)) return 0;
    return nt_inbn_kind(a, g, nt_cfg(a.M, N, a.K, dtype), 0, dtype);
}
extern "C" int capmi_igemm_nt_bnact(const void* x_raw, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                                    const float* in_mean, const float* in_coef_a, const float* in_offset, int in_act,
                                    float* stats, int dtype, void* stream) {
    CAPMI_CHECK(in_mean && in_coef_a && in_offset, "capmi_igemm_nt_bnact: null batch-norm vector");
    CAPMI_CHECK(in_act == CAPMI_ACT_RELU || in_act == CAPMI_ACT_RELU6, "capmi_igemm_nt_bnact: the input activation must be relu or relu6 (got %d)", in_act);
    CAPMI_CHECK(((uintptr_t)in_mean | (uintptr_t)in_coef_a | (uintptr_t)in_offset) % 16 == 0, "capmi_igemm_nt_bnact: batch-norm vectors must be 16-byte aligned");
    IGemmArgs a;
    if (nt_prepare(a, x_raw, w, y, g, N, ldw, ldy, nullptr, nullptr, 0, nullptr, 0, stats, 0, 0, 0, 0, nullptr, dtype)) return 1;
    a.in_mean = in_mean; a.in_a = in_coef_a; a.in_off = in_offset; a.in_act = in_act;
    return nt_dispatch(a, g, N, stats, 0, dtype, (hipStream_t)stream);
}

extern "C" int capmi_igemm_nt_bnred_part_rows(const capmi_conv_geom* g, int N, int dtype) {
    if (!g) return 0;
    const int M = g->B * g->Ho * g->Wo, K = g->kh * g->kw * g->Cin;
    if (nt_uses_skinny(g, M, K, false, dtype)) return 0;
    return nt_cfg(M, N, K, dtype).bm;
}

extern "C" int capmi_igemm_nt_bnred(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                                    const void* addend, int ld_addend, const void* ysaved, int ld_saved, int dact,
                                    int nred, const void* rx0, const float* mean0, const float* invstd0, float* ws0,
                                    const void* rx1, const float* mean1, const float* invstd1, float* ws1, int dtype, void* stream) {
    CAPMI_CHECK(nred >= 1 && nred <= 2, "capmi_igemm_nt_bnred: nred=%d (1 or 2)", nred);
    CAPMI_CHECK(rx0 && mean0 && invstd0 && ws0 && (nred < 2 || (rx1 && mean1 && invstd1 && ws1)), "capmi_igemm_nt_bnred: null reduction target");
    CAPMI_CHECK(ldy == N, "capmi_igemm_nt_bnred: output must be dense (ldy == N)");
    BnRedTarget red[2] = {{rx0, mean0, invstd0, ws0}, {rx1, mean1, invstd1, ws1}};
    return igemm_nt_impl(x, w, y, g, N, ldw, ldy, nullptr, addend, ld_addend, ysaved, ld_saved, nullptr, 0, dact, 0, nred, red, dtype, stream);
}

/* 1 when capmi_igemm_nt_stat has a kernel with the sums epilogue (EPI 8) for this convolution: the bf16 LDS-DMA tiles and the
 * halo-staged 3x3 kernel (dense shapes: N a multiple of 8, >= 32 columns).  0: launch capmi_igemm_nt with a parts workspace +
 * capmi_bn_finalize + capmi_bn_apply. */
extern "C" int capmi_igemm_nt_stat_supported(const capmi_conv_geom* g, int N, int dtype) {
    if (!g || dtype != CAPMI_BF16 || N % 8 != 0 || N < 32 || g->os > 1) return 0;
    const int M = g->B * g->Ho * g->Wo, K = g->kh * g->kw * g->Cin;
    if (nt_uses_skinny(g, M, K, true, dtype)) return 0;
    const NtCfg c = nt_cfg(M, N, K, dtype);
    IGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K; a.stats = reinterpret_cast<float*>(1);
    if (nt_halo3_ok(a, g, 0)) return 1;
    return (c.wmw == 5 || c.wmw == 6 || c.bn == 128) ? 1 : 0;
}
/* A training convolution with its batch statistics (IC/model/MobileNetV2.py:99-117: conv2d -> batch_norm, train mode) as
 * capmi_bn_stat_apply consumes them.  Default (bf16): per-column sum v and sum v^2 of the f32 accumulators ADDED (f32 atomics)
 * to stat_rows[4][2N], which the caller zeroes once per step -- the sums are of (v - shift[n]) and its square: any shift is exact in
 * exact arithmetic, one near the column's mean (the previous step's batch mean) keeps the one-pass variance free of
 * cancellation in f32.  Deterministic mode (capmi.h), or a shape without the sums
 * epilogue when `parts` is given: the exact (mean, M2) parts of capmi_igemm_nt into `parts` instead -- capmi_bn_stat_apply
 * makes the same choice from the same switch. */
extern "C" int capmi_igemm_nt_stat(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                                   float* parts, float* stat_rows, const float* shift, int dtype, void* stream) {
    CAPMI_CHECK(parts && stat_rows && shift, "capmi_igemm_nt_stat: null statistics buffer");
    CAPMI_CHECK(capmi_igemm_nt_stat_supported(g, N, dtype), "capmi_igemm_nt_stat: no kernel with the sums epilogue for this shape (capmi_igemm_nt_stat_supported)");
    IGemmArgs a;
    if (nt_prepare(a, x, w, y, g, N, ldw, ldy, nullptr, nullptr, 0, nullptr, 0, parts, 0, 0, 0, 0, nullptr, dtype)) return 1;
    if (!capmi_deterministic()) {
        a.stats = stat_rows;
        a.stat_sums = 1;
        a.stat_shift = shift;
    }
    return nt_dispatch(a, g, N, a.stats, 0, dtype, (hipStream_t)stream);
}

/* The kernel family that carries the batch-norm backward sums in its epilogue (EPI 7): the LDS-DMA tiles (any addressing mode,
 * k-groups included) and the halo-staged 3x3 kernel; not the register-staged kernel (ragged / narrow shapes), not the skinny
 * kernel, not grouped launches (strided data gradients).  Returns the row-block height of the per-tile parts (deterministic
 * mode: parts workspace of ceil(M / height) * 2N floats), 0 when the shape has no such kernel. */
extern "C" int capmi_igemm_nt_bnsum_part_rows(const capmi_conv_geom* g, int N, int dtype) {
    if (!g || dtype != CAPMI_BF16 || N % 8 != 0 || N < 32 || g->os > 1) return 0;
    const int M = g->B * g->Ho * g->Wo, K = g->kh * g->kw * g->Cin;
    if (nt_uses_skinny(g, M, K, false, dtype) || (g->Hi == 1 && g->Wi == 1)) return 0;
    const NtCfg c = nt_cfg(M, N, K, dtype);
    IGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K; a.red_mode = 1;
    if (nt_halo3_ok(a, g, 1)) return c.bm;
    const bool glds = c.wmw == 5 || c.wmw == 6 || c.bn == 128;
    return glds ? c.bm : 0;
}
/* capmi_igemm_nt (a convolution's data gradient with its ReLU mask as bits: dact carries CAPMI_DACT_BITMASK) that ALSO takes the
 * batch-norm backward sums of the layer whose output gradient it completes -- `raw` is that layer's conv output [M][N] (same
 * rows as y), mean / invstd its saved statistics: sum_m dz and sum_m dz * (raw - mean) * invstd per channel, dz = the stored
 * (masked, bf16) gradient.  Default: added to acc_rows[4][2N] (f32 atomics; the rows capmi_bn_bwd_apply_spread consumes,
 * zeroed by the caller once per step) -- capmi_bn_bwd_reduce_spread of that layer is not launched at all.  Deterministic mode
 * (capmi.h): per-tile parts into parts_ws + a fixed-order sum into red ([d offset | d scale]), what capmi_bn_bwd_reduce leaves.
 * IC/model/MobileNetV2.py:112-119 backward (batch_norm_grad's reductions) inside conv2d_grad of the consumer. */
extern "C" int capmi_igemm_nt_bnsum(const void* x, const void* w, void* y, const capmi_conv_geom* g, int N, int ldw, int ldy,
                                    const void* addend, int ld_addend, const void* maskbits, int ld_saved, int dact,
                                    const void* raw, const float* mean, const float* invstd, float* acc_rows, float* parts_ws,
                                    float* red, int dtype, void* stream) {
    CAPMI_CHECK(raw && mean && invstd && acc_rows, "capmi_igemm_nt_bnsum: null batch-norm operand");
    CAPMI_CHECK((dact & CAPMI_DACT_BITMASK) && maskbits, "capmi_igemm_nt_bnsum: the ReLU mask must come as bits (CAPMI_DACT_BITMASK)");
    CAPMI_CHECK(ldy == N, "capmi_igemm_nt_bnsum: output must be dense (ldy == N)");
    const int part_rows = capmi_igemm_nt_bnsum_part_rows(g, N, dtype);
    CAPMI_CHECK(part_rows > 0, "capmi_igemm_nt_bnsum: no kernel with the sums epilogue for this shape (capmi_igemm_nt_bnsum_part_rows)");
    const bool det = capmi_deterministic() != 0;
    CAPMI_CHECK(!det || (parts_ws && red), "capmi_igemm_nt_bnsum: deterministic mode needs the parts workspace and red");
    IGemmArgs a;
    if (nt_prepare(a, x, w, y, g, N, ldw, ldy, nullptr, addend, ld_addend, maskbits, ld_saved, nullptr, 0, dact, 0, 0, nullptr, dtype)) return 1;
    a.nred = 1;
    a.red_mode = det ? 2 : 1;
    a.rx[0] = raw; a.rmean[0] = mean; a.rinv[0] = invstd; a.rws[0] = det ? parts_ws : acc_rows;
    if (nt_dispatch(a, g, N, nullptr, 1, dtype, (hipStream_t)stream)) return 1;
    if (det && !t_probe) return capmi_bn_bwd_reduce_final(parts_ws, cdiv(a.M, part_rows), N, red, stream);
    return 0;
}

/* Independent NT products (disjoint outputs) issued together; see capmi.h. */
extern "C" int capmi_igemm_nt_group(const capmi_igemm_nt_call* calls, int count, int dtype, void* stream) {
    CAPMI_CHECK(calls && count >= 1, "capmi_igemm_nt_group: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int first = 0; first < count; first += NT_GROUP_MAX) {
        const int n = count - first < NT_GROUP_MAX ? count - first : NT_GROUP_MAX;
        NtGroup grp;
        // one launch when every call lands on the SAME LDS-DMA tile shape: 64 x 128 (4 x 1 waves) or 64 x 64
        bool fuse128 = dtype == CAPMI_BF16 && n > 1, fuse64 = fuse128, conv1 = true, lin = true;
        long long blocks128 = 0, blocks64 = 0;
        for (int i = 0; i < n; ++i) {
            const capmi_igemm_nt_call& c = calls[first + i];
            if (nt_prepare(grp.a[i], c.x, c.w, c.y, &c.g, c.N, c.ldw, c.ldy, c.bias, c.addend, c.ld_addend, c.ysaved, c.ld_saved, nullptr,
                           c.act, c.dact, 0, 0, nullptr, dtype)) return 1;
            const IGemmArgs& a = grp.a[i];
            const bool skinny = nt_uses_skinny(&c.g, a.M, a.K, false, dtype);
            const NtCfg cfg = nt_cfg(a.M, a.N, a.K, dtype);
            fuse128 = fuse128 && !skinny && cfg.bn == 128 && cfg.wmw == 4 && nt_dense(a);
            fuse64 = fuse64 && !skinny && cfg.wmw == 5 && !nt_halo3_ok(a, &c.g, 0) && nt_dense(a);
            conv1 = conv1 && c.g.up == 1 && c.g.Cin >= 32;
            lin = lin && c.g.kh == 1 && c.g.kw == 1 && c.g.up == 1 && c.g.pad == 0 && (c.g.Ho - 1) * c.g.sd < c.g.Hi && (c.g.Wo - 1) * c.g.sd < c.g.Wi;
            blocks128 += ((long long)cdiv(a.M, 64) * cdiv(a.N, 128) + 7) / 8 * 8;      // ranges start at multiples of 8 (XCD order)
            blocks64 += ((long long)cdiv(a.M, 64) * cdiv(a.N, 64) + 7) / 8 * 8;
        }
        grp.count = n;
        const long long blocks = fuse128 ? blocks128 : blocks64;
        bool mixed = false;      // a group either reads mask bits in every call or in none
        { bool any = false, all = true; for (int i = 0; i < n; ++i) { any = any || nt_bits_class(grp.a[i]); all = all && nt_bits_class(grp.a[i]); } mixed = any && !all; }
        if ((fuse128 || fuse64) && !mixed && blocks < (1ll << 31)) {
            long long b = 0;
            for (int i = 0; i < n; ++i) {
                grp.first[i] = (int)b;
                b += ((long long)cdiv(grp.a[i].M, 64) * cdiv(grp.a[i].N, fuse128 ? 128 : 64) + 7) / 8 * 8;
            }
            grp.first[n] = (int)b;
            bool cc = true;
            for (int i = 0; i < n; ++i) cc = cc && nt_conv_class(grp.a[i]) && !grp.a[i].stats;      // (groups carry no statistics: the data-gradient form)
            bool allbits = true, anybits = false;
            for (int i = 0; i < n; ++i) { allbits = allbits && nt_bits_class(grp.a[i]); anybits = anybits || nt_bits_class(grp.a[i]); }
            bool fc = !cc;
            for (int i = 0; i < n; ++i) fc = fc && nt_fc_class(grp.a[i]);
            if (fuse128 && allbits) {
                if (conv1) CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 128, 3, 2, 6>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
                else CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 128, 3, 0, 6>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
            } else if (fuse128 && cc) {
                if (conv1) CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 128, 3, 2, 4>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
                else CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 128, 3, 0, 4>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
            } else if (fuse128) {
                if (conv1) CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 128, 3, 2>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
                else CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 128, 3, 0>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
            } else if (allbits) {           // (the same tile as without the bits: the kernel choice must not depend on the mask's form)
                if (lin) CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 64, 3, 1, 6>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
                else CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 64, 3, 0, 6>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
            } else if (fc && lin) {         // the decode step's grouped projections (bias, tanh)
                CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 64, 3, 1, 2>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
            } else if (cc) {
                if (lin) CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 64, 3, 1, 4>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
                else CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 64, 3, 0, 4>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
            } else {
                if (lin) CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 64, 3, 1>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
                else CAPMI_KLAUNCH((igemm_nt_glds_group_kernel<64, 64, 3, 0>), dim3((unsigned)blocks), dim3(256), 0, st, grp);
            }
            CAPMI_LAUNCH_CHECK("capmi_igemm_nt_group");
        } else {
            for (int i = 0; i < n; ++i) {
                const capmi_igemm_nt_call& c = calls[first + i];
                if (nt_dispatch(grp.a[i], &c.g, c.N, nullptr, 0, dtype, st)) return 1;
            }
        }
    }
    return 0;
}

// ------------------------------------------------------------------ TN kernel (weight gradient)
struct WGradArgs {
    const void* x;
    const void* dy;
    float* dw;
    int M, N, K;
    int ldy, lddw;
    int m_per_split, splits;
    int linear;       // 1x1 / stride 1 / no padding: A(m,k) = x[m*ldx + k]
    float* slab;      // [splits][Kp][Np] f32 partial tiles (transposed), or NULL: add straight into dw
    int Np, Kp;
    capmi_conv_geom g;
    FastDiv fd_hw, fd_w;
};

// 8 reduction rows (8g..8g+7) x one column of a [rows][LD] LDS tile -> MFMA fragment.
__device__ __forceinline__ void load_frag_tr(Frag<bf16>& f, const bf16* tile, int LD, int g, int col16, int i) {
    // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the address of row q, columns
    // 4p..4p+3 of a 4x16 block; lane i receives column i, rows 0..3 in elements 0..3.
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const bf16* p0 = tile + (8 * g + (i >> 2)) * LD + col16 + (i & 3) * 4;
    const bf16* p1 = p0 + 4 * LD;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, both);
}
__device__ __forceinline__ void load_frag_tr(Frag<float>& f, const float* tile, int LD, int g, int col16, int i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = tile[(8 * g + j) * LD + col16 + i];
}

template <int BNO, int BKO, int TN_, int TK_>
__device__ __forceinline__ void tn_epilogue(const WGradArgs& a, f32x4 (&acc)[TN_][TK_], int n0, int k0, int split, int wn, int wk, int fr, int fg) {
    if (a.slab) {
        // partial tile -> slab[split][k][n] (transposed): the accumulator quad of a lane is 4
        // consecutive n of one k, i.e. one 16-byte store; no atomics (wgrad_reduce_kernel sums the splits)
        float* sl = a.slab + (size_t)split * a.Kp * a.Np;
#pragma unroll
        for (int i = 0; i < TN_; ++i)
#pragma unroll
            for (int j = 0; j < TK_; ++j) {
                const int k = k0 + wk * (TK_ * 16) + j * 16 + fr;
                const int n = n0 + wn * (TN_ * 16) + i * 16 + fg * 4;
                *reinterpret_cast<f32x4*>(&sl[(size_t)k * a.Np + n]) = acc[i][j];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TN_; ++i)
#pragma unroll
        for (int j = 0; j < TK_; ++j) {
            const int k = k0 + wk * (TK_ * 16) + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * (TN_ * 16) + i * 16 + fg * 4 + r;
                if (n < a.N && k < a.K) {
                    float* p = &a.dw[(int64_t)n * a.lddw + k];
                    if (a.splits == 1) *p += acc[i][j][r];          // single split: sole contributor
                    else atomicAdd(p, acc[i][j][r]);                 // small outputs (64x64 tiles): few bytes, direct atomics
                }
            }
        }
}

template <typename T, int BNO, int BKO>
__global__ __launch_bounds__(256) void igemm_tn_kernel(WGradArgs a) {
    constexpr int RM = 64;                      // reduction rows per step: one LDS stage + one stage in registers
    constexpr int VEC = Vec<T>::N;
    constexpr int LDY = BNO + VEC, LDX = BKO + VEC;
    constexpr int CPY = BNO / VEC, CPX = BKO / VEC;        // chunks per row
    constexpr int YCH = RM * CPY / 256, XCH = RM * CPX / 256;
    static_assert(YCH >= 1 && XCH >= 1, "tile too small for 256 threads");
    constexpr int TN_ = BNO / 32, TK_ = BKO / 32;
    __shared__ __attribute__((aligned(16))) T Ys[RM * LDY];
    __shared__ __attribute__((aligned(16))) T Xs[RM * LDX];

    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ DY = (const T*)a.dy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    // split-major work order, dealt to the XCDs in contiguous runs: the tiles of one split (same pixel
    // rows of dY and X) meet in one L2 instead of being fetched by all eight
    const int tiles_k = (a.K + BKO - 1) / BKO;
    const int tiles = tiles_k * ((a.N + BNO - 1) / BNO);
    const int id = xcd_swizzle(blockIdx.x, gridDim.x);
    const int split = id / tiles, tl = id - split * tiles;
    const int n0 = (tl / tiles_k) * BNO;
    const int k0 = (tl % tiles_k) * BKO;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    if (m_begin >= m_end) return;

    const int yc = tid % CPY, yr0 = tid / CPY;      // dY chunk column / first row
    const int xc = tid % CPX, xr0 = tid / CPX;
    constexpr int YRS = 256 / CPY, XRS = 256 / CPX;
    const KPos kp = k_pos(k0 + xc * VEC, a.g);      // this thread's k: fixed for the whole kernel
    const int ncol = n0 + yc * VEC;
    const bool yok = ncol < a.N, xok = kp.k < a.K;

    Vec<T> ry[YCH], rx[XCH];
    auto load_tile = [&](int mt) {
#pragma unroll
        for (int i = 0; i < YCH; ++i) {
            int m = mt + yr0 + i * YRS;
            ry[i] = (m < m_end && yok) ? vload<T>(DY + (int64_t)m * a.ldy + ncol) : vzero<T>();
        }
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            int m = mt + xr0 + i * XRS;
            int64_t off = -1;
            if (m < m_end && xok) {
                if (a.linear) off = (int64_t)m * a.g.ldx + kp.k;
                else off = a_offset(row_pos(m, a.M, a.g, a.fd_hw, a.fd_w), kp, a.K, a.g);
            }
            rx[i] = off >= 0 ? vload<T>(X + off) : vzero<T>();
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < YCH; ++i) vstore<T>(&Ys[(yr0 + i * YRS) * LDY + yc * VEC], ry[i]);
#pragma unroll
        for (int i = 0; i < XCH; ++i) vstore<T>(&Xs[(xr0 + i * XRS) * LDX + xc * VEC], rx[i]);
    };

    f32x4 acc[TN_][TK_];
#pragma unroll
    for (int i = 0; i < TN_; ++i)
#pragma unroll
        for (int j = 0; j < TK_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    load_tile(m_begin);
    for (int mt = m_begin; mt < m_end; mt += RM) {
        if (mt != m_begin) __syncthreads();       // every wave is done reading the previous tile
        store_tile();
        __syncthreads();
        if (mt + RM < m_end) load_tile(mt + RM);  // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < RM / 32; ++ks) {
            Frag<T> af[TN_], bf[TK_];
#pragma unroll
            for (int i = 0; i < TN_; ++i) load_frag_tr(af[i], Ys + ks * 32 * LDY, LDY, fg, wn * (BNO / 2) + i * 16, fr);
#pragma unroll
            for (int j = 0; j < TK_; ++j) load_frag_tr(bf[j], Xs + ks * 32 * LDX, LDX, fg, wk * (BKO / 2) + j * 16, fr);
#pragma unroll
            for (int i = 0; i < TN_; ++i)
#pragma unroll
                for (int j = 0; j < TK_; ++j) mma16(acc[i][j], af[i], bf[j]);
        }
    }
    tn_epilogue<BNO, BKO, TN_, TK_>(a, acc, n0, k0, split, wn, wk, fr, fg);
}

// LDS-DMA form of the weight-gradient kernel (bf16, 128 x 128 output tile, 64 reduction rows per stage,
// 2-stage ring, no staging registers): the kernel runs at ~1.5x its DMA staging time (tools/tn_ablate.hip), not
// bound by MFMA or LDS.  A stage holds dY [64][128] and X(im2col) [64][128] as unpadded 256-byte rows; a DMA wave
// instruction fills 4 rows.  ds_read_b64_tr_b16 touches 16 rows x 32 B per instruction, so the 16-byte
// chunk c of row r sits at position c ^ f(r), f(r) = (r & 3) << 1 | ((r >> 3) & 1) << 3 (applied on the
// SOURCE address of the DMA).  The instruction is served in two 32-lane halves, each 8 rows x 32 B (rows 0-3 and 8-11
// of its 16, or the same + 4): f(r) >> 1 numbers those eight rows 0..7, so their 32-byte segments fill the 256-byte
// bank row exactly once.  (The first form, (r & 3) | ((r >> 3) & 3) << 2, put two rows on every segment: a 2-way
// conflict on every read, SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles.)
// The reads are inline assembly on purpose: behind the builtin the compiler cannot tell the transposing read from the
// LDS-DMA writes in flight and puts s_waitcnt vmcnt(0) in front of the first one -- the next stage's DMA then never
// overlaps this stage's MFMAs (measured: kernel time = DMA time + compute time).  The caller waits on lgkmcnt itself
// (tr_reads_done) before the first MFMA.
__device__ __forceinline__ void load_frag_tr_swz(Frag<bf16>& f, const char* tile, int g, int col16, int i) {
    typedef __attribute__((address_space(3))) char lds_char;
    const int fsw = ((i >> 2) << 1) | ((g & 1) << 3);
    const int bytecol = col16 * 2 + (i & 3) * 8;
    const char* p0 = tile + (8 * g + (i >> 2)) * 256 + ((((bytecol >> 4) ^ fsw) << 4) | (bytecol & 15));
    const uint32_t a0 = (uint32_t)(uintptr_t)(lds_char*)p0;
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a0) : "memory");
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, both);
}
__device__ __forceinline__ void tr_reads_done() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ABL (tools/tn_ablate.hip only; the library instantiates 0): 1 = no MFMAs, 2 = no LDS reads either, 4 = no DMA.
// KS = 1: 8 waves as 2 (n) x 4 (k), 64 x 32 each, every wave walks both k-steps of a stage (12 fragment reads per 16 MFMAs
//         and wave: the transposing LDS reads, 96 KB per stage and CU, are what the loop waits for -- tools/tn_ablate).
// KS = 2: 2 (k-step) x 2 (n) x 2 (k), 64 x 64 each, wave group g multiplies only k-step g of every stage (8 reads per
//         16 MFMAs: a third of the LDS traffic); the two groups' accumulators meet in LDS before the epilogue.
// PIPE: fragments double-buffered in registers -- iteration t issues the DMA of stage t + NST (into the slot whose
//       fragments were just consumed into registers), reads the fragments of stage t + 1 and multiplies stage t: the LDS
//       read latency sits under the MFMAs and a DMA has NST - 1 iterations to land.
template <int ABL, int NST = 2, int KS = 1, bool PIPE = false, int RM = 64>
__global__ __launch_bounds__(512, (KS == 2 || PIPE) ? 1 : 2) void igemm_tn_glds_kernel(WGradArgs a) {
    typedef bf16 T;
    constexpr int BNO = 128, BKO = 128;                       // RM = 64 reduction rows per stage: two MFMA k-steps per barrier
    static_assert(RM == 32 || RM == 64, "a stage is one or two 32-row MFMA k-steps");
    constexpr int OPB = RM * 256, STB = 2 * OPB;            // bytes per operand tile / per stage
    constexpr int TN_ = 4, TK_ = KS == 2 ? 4 : 2;
    static_assert(KS == 1 || NST * STB >= BNO * BKO * 4, "the accumulator exchange needs one f32 tile inside the ring");
    __shared__ __attribute__((aligned(1024))) char smem[NST * STB];

    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ DY = (const T*)a.dy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = KS == 2 ? wave >> 2 : 0;                                  // k-step group
    const int wn = KS == 2 ? (wave >> 1) & 1 : wave >> 2, wk = KS == 2 ? wave & 1 : wave & 3;
    const int tiles_k = (a.K + BKO - 1) / BKO;
    const int tiles = tiles_k * ((a.N + BNO - 1) / BNO);
    const int id = xcd_swizzle(blockIdx.x, gridDim.x);
    const int split = id / tiles, tl = id - split * tiles;
    const int n0 = (tl / tiles_k) * BNO;
    const int k0 = (tl % tiles_k) * BKO;
    const int m_begin = split * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    if (m_begin >= m_end) return;

    // this thread's DMA slot: row 4*wave + (lane >> 4), LDS chunk position lane & 15
    // = global chunk (lane & 15) ^ f(row)
    const int chunk = (lane & 15) ^ (((lane >> 4) << 1) | (((wave >> 1) & 1) << 3));
    const int ncol = n0 + chunk * 8;
    const KPos kp = k_pos(k0 + chunk * 8, a.g);             // fixed for the whole kernel
    const bool yok = ncol < a.N, xok = kp.k < a.K;
    const int r0 = 4 * wave + (lane >> 4);
    const T* zero = reinterpret_cast<const T*>(capmi_zero_page);

    // branch-free addressing (selects only): the loop is bound by instruction issue, not by memory
    const int tap_h = kp.r - a.g.pad, tap_w = kp.q - a.g.pad;       // input row = ho*sd + tap_h
    const int sd = a.g.sd, Hi = a.g.Hi, Wi = a.g.Wi, ldx = a.g.ldx, ldy = a.ldy;
    // Source addresses as integers, invalid lanes folded onto the zero page by a mask (delta & -ok): pure arithmetic,
    // the compiler cannot turn it back into exec-masked branches around the address computation.
    const uint64_t zaddr = (uint64_t)zero, yaddr0 = (uint64_t)(DY + ncol), xaddr0 = (uint64_t)X;
    auto issue_stage = [&](int st, int mt) {
#pragma unroll
      for (int h = 0; h < RM / 32; ++h) {       // rows r0 and r0 + 32: the same chunk swizzle f(row)
        char* base = smem + st * STB + h * 8192 + wave * 1024;
        const int m = mt + r0 + 32 * h;
        const int64_t ok = m < m_end ? -1 : 0;
        const uint64_t ya = yaddr0 + (uint64_t)((int64_t)m * ldy) * 2;
        const uint64_t ysrc = zaddr + ((ya - zaddr) & (uint64_t)(yok ? ok : 0));
        uint64_t xa;
        int64_t okx;
        if (a.linear) {             // 1x1 / stride 1: A(m, k) = x[m][k]
            xa = xaddr0 + (uint64_t)((int64_t)m * ldx + kp.k) * 2;
            okx = xok ? ok : 0;
        } else {
            const int b = fdiv(m, a.fd_hw), rem = m - b * a.fd_hw.d;
            const int ho = fdiv(rem, a.fd_w), wo = rem - ho * a.fd_w.d;
            const int hn = ho * sd + tap_h, wn = wo * sd + tap_w;
            okx = (xok && (unsigned)hn < (unsigned)Hi && (unsigned)wn < (unsigned)Wi) ? ok : 0;
            xa = xaddr0 + (uint64_t)((int64_t)((b * Hi + hn) * Wi + wn) * ldx + kp.c) * 2;
        }
        const uint64_t xsrc = zaddr + ((xa - zaddr) & (uint64_t)okx);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ysrc,
                                         (__attribute__((address_space(3))) void*)base, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xsrc,
                                         (__attribute__((address_space(3))) void*)(base + OPB), 16, 0, 0);
      }
    };

    f32x4 acc[TN_][TK_];
#pragma unroll
    for (int i = 0; i < TN_; ++i)
#pragma unroll
        for (int j = 0; j < TK_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int nsteps = (m_end - m_begin + RM - 1) / RM;
    auto rows_of = [&](int step) { return step < nsteps ? m_begin + step * RM : m_end; };   // >= m_end: nothing to load
    if constexpr (PIPE) {
        constexpr int KPW = RM / 32 / KS, NDMA = 2 * (RM / 32);
        Frag<T> afA[KPW][TN_], bfA[KPW][TK_], afB[KPW][TN_], bfB[KPW][TK_];
        auto read_stage = [&](Frag<T> (&af)[KPW][TN_], Frag<T> (&bf)[KPW][TK_], int slot) {
#pragma unroll
            for (int ks = 0; ks < KPW; ++ks) {
                const char* sk = smem + slot * STB + (KS == 2 ? wg : ks) * 8192;
#pragma unroll
                for (int i = 0; i < TN_; ++i) load_frag_tr_swz(af[ks][i], sk, fg, wn * 64 + i * 16, fr);
#pragma unroll
                for (int j = 0; j < TK_; ++j) load_frag_tr_swz(bf[ks][j], sk + OPB, fg, wk * (TK_ * 16) + j * 16, fr);
            }
        };
        auto mma_stage = [&](Frag<T> (&af)[KPW][TN_], Frag<T> (&bf)[KPW][TK_]) {
#pragma unroll
            for (int ks = 0; ks < KPW; ++ks)
#pragma unroll
                for (int i = 0; i < TN_; ++i)
#pragma unroll
                    for (int j = 0; j < TK_; ++j) mma16(acc[i][j], af[ks][i], bf[ks][j]);
        };
#pragma unroll
        for (int p = 0; p < NST; ++p) issue_stage(p, rows_of(p));
        wait_vmcnt<(NST - 1) * NDMA>();                       // stage 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_stage(afA, bfA, 0);
        int slot = 0;                                         // ring slot of stage t
        // one iteration: stage t + 1 has landed everywhere and every wave holds stage t in registers (its slot is free)
        auto step = [&](int t, Frag<T> (&afc)[KPW][TN_], Frag<T> (&bfc)[KPW][TK_], Frag<T> (&afn)[KPW][TN_], Frag<T> (&bfn)[KPW][TK_]) {
            wait_vmcnt<(NST - 2) * NDMA>();                   // this thread's part of stage t + 1
            tr_reads_done();                                  // the fragments of stage t are in registers
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_stage(slot, rows_of(t + NST));
            slot = slot + 1 == NST ? 0 : slot + 1;
            read_stage(afn, bfn, slot);
            __builtin_amdgcn_sched_barrier(0);                // reads of t + 1 in flight before the MFMAs of t
            mma_stage(afc, bfc);
        };
        for (int t = 0; t < nsteps; t += 2) {
            step(t, afA, bfA, afB, bfB);
            if (t + 1 < nsteps) step(t + 1, afB, bfB, afA, bfA);
        }
        tr_reads_done();                                      // the trailing (unused) fragment reads
    } else {
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (!(ABL & 4)) issue_stage(p, rows_of(p));
    int slot = 0;
    for (int s = 0; s < nsteps; ++s) {
        wait_vmcnt<(NST - 2) * 2 * (RM / 32)>();              // this thread's part of stage s has landed
        __builtin_amdgcn_s_barrier();                         // ... everyone's; all waves are done with stage s-1
        asm volatile("" ::: "memory");
        if (!(ABL & 4)) issue_stage(slot == 0 ? NST - 1 : slot - 1, rows_of(s + NST - 1));
        const char* st = smem + slot * STB;
        slot = slot + 1 == NST ? 0 : slot + 1;
        if (ABL & 2) continue;
        constexpr int KPW = RM / 32 / KS;          // k-steps of a stage per wave
        Frag<T> af[KPW][TN_], bf[KPW][TK_];
#pragma unroll
        for (int ks = 0; ks < KPW; ++ks) {
            const char* sk = st + (KS == 2 ? wg : ks) * 8192;
#pragma unroll
            for (int i = 0; i < TN_; ++i) load_frag_tr_swz(af[ks][i], sk, fg, wn * 64 + i * 16, fr);
#pragma unroll
            for (int j = 0; j < TK_; ++j) load_frag_tr_swz(bf[ks][j], sk + OPB, fg, wk * (TK_ * 16) + j * 16, fr);
        }
        __builtin_amdgcn_sched_barrier(0);        // every transposing read of the stage in flight before the first MFMA
        tr_reads_done();
        if (ABL & 1) {                            // keep the reads alive without the matrix pipe
#pragma unroll
            for (int ks = 0; ks < KPW; ++ks) {
#pragma unroll
                for (int i = 0; i < TN_; ++i) acc[i][0][0] += (float)af[ks][i].v[0];
#pragma unroll
                for (int j = 0; j < TK_; ++j) acc[0][j][1] += (float)bf[ks][j].v[7];
            }
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < KPW; ++ks)
#pragma unroll
            for (int i = 0; i < TN_; ++i)
#pragma unroll
                for (int j = 0; j < TK_; ++j) mma16(acc[i][j], af[ks][i], bf[ks][j]);
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (KS == 2) {
        // group 1 hands its accumulators to group 0 through the (now idle) ring: one 16-byte slot per lane and tile
        __syncthreads();
        f32x4* xch = reinterpret_cast<f32x4*>(smem) + (wave & 3) * (TN_ * TK_ * 64) + lane;
        if (wg == 1) {
#pragma unroll
            for (int i = 0; i < TN_; ++i)
#pragma unroll
                for (int j = 0; j < TK_; ++j) xch[(i * TK_ + j) * 64] = acc[i][j];
        }
        __syncthreads();
        if (wg == 1) return;
#pragma unroll
        for (int i = 0; i < TN_; ++i)
#pragma unroll
            for (int j = 0; j < TK_; ++j) acc[i][j] += xch[(i * TK_ + j) * 64];
    }
    tn_epilogue<BNO, BKO, TN_, TK_>(a, acc, n0, k0, split, wn, wk, fr, fg);
}

// dw[n][k] += sum_s slab[s][k][n]: 32x32 tiles, coalesced reads along n, LDS transpose, coalesced writes along k
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int Np, int Kp, float* dw, int N, int K, int lddw) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) {
        const float* sl = slab + (size_t)s * Kp * Np;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += sl[(size_t)(k0 + ty + 8 * i) * Np + n0 + tx];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = acc[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        if (n < N && k < K) dw[(int64_t)n * lddw + k] += tile[tx][ty + 8 * i];
    }
}

// Splits over the reduction (pixel) axis: at most ONE workgroup per CU for the 128x128 LDS-DMA kernel (two
// for the 64x64 one).  Weight gradients run on the side lane under the BN-backward / data-gradient chain,
// which has twice their work: fewer, longer workgroups leave compute units and L2 to the main lane
// (+1 % per step over filling every resident slot) and write fewer slab partials.  Never one workgroup more
// than a full round (that costs a whole second round); every split >= 256 rows deep; slabs capped at
// 6 M floats (24 MB: ~4 us to store, ~5 us to read back).
static int tn_splits(int M, int N, int K, int bno, int bko, int* per_out) {
    const int tiles = cdiv(N, bno) * cdiv(K, bko);
    const long long out_elems = (long long)cdiv(N, bno) * bno * cdiv(K, bko) * bko;
    long long by_ws = (6ll << 20) / (out_elems > 0 ? out_elems : 1);
    if (by_ws < 1) by_ws = 1;
    static const int slots_env = getenv("CAPMI_TN_SLOTS") ? atoi(getenv("CAPMI_TN_SLOTS")) : 0;      // experiment knob
    int slots = slots_env > 0 ? slots_env * (bno >= 128 ? 1 : 2) : 256 * (bno >= 128 ? 1 : 2);
    // The one weight gradient with more than half a million reduction rows is the stem's, and the stem's is the LAST launch of
    // the backward pass: nothing runs beside it (the main lane is waiting for it), so it fills the chip instead of leaving
    // room -- 97.7 -> 71.3 us alone (tools/wgrad_probe.py), on the step's critical tail.
    static const int stem_fill = getenv("CAPMI_STEM_FILL") ? atoi(getenv("CAPMI_STEM_FILL")) : 1;      // experiment knob
    if (!slots_env && stem_fill && M >= (1 << 19)) slots *= 2;
    int want = slots / tiles;
    int max_splits = cdiv(M, 256);
    if (max_splits > by_ws) max_splits = (int)by_ws;
    int splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    int per = cdiv(M, splits);
    per = (per + 63) / 64 * 64;
    *per_out = per;
    return cdiv(M, per);
}
static void tn_tile(int N, int K, int dtype, int* bno, int* bko) {
    // CAPMI_TN_PAD64=1 (experiment): the LDS-DMA kernel also for 64-wide operands (the 56 x 56 layers of a ResNet, the stem) -- its 128 x 128
    // tile padded from the zero page (MFMA and DMA slots on zeros are free next to the register-staged kernel's 77 us in the model)
    static const int pad64 = getenv("CAPMI_TN_PAD64") ? atoi(getenv("CAPMI_TN_PAD64")) : 0;
    const bool big = dtype == CAPMI_BF16 && ((N >= 128 && K >= 128) || (pad64 && N >= 64 && K >= 64 && N % 8 == 0 && K % 8 == 0));
    *bno = *bko = big ? 128 : 64;
}

extern "C" long long capmi_igemm_tn_ws_bytes(int M, int N, int K, int dtype) {
    int bno, bko, per;
    tn_tile(N, K, dtype, &bno, &bko);
    const int splits = tn_splits(M, N, K, bno, bko, &per);
    if (splits <= 1 || ((splits > 24 || bno < 128) && !capmi_deterministic())) return 0;
    return (long long)splits * cdiv(N, bno) * bno * cdiv(K, bko) * bko * 4;
}

template <typename T, int BNO, int BKO>
static int launch_tn(WGradArgs& a, float* ws, long long ws_bytes, hipStream_t st) {
    int tiles = cdiv(a.N, BNO) * cdiv(a.K, BKO);
    if (tiles <= 0 || a.M <= 0) return 0;
    int per;
    const int splits = tn_splits(a.M, a.N, a.K, BNO, BKO, &per);
    a.m_per_split = per;
    a.splits = splits;
    CAPMI_CHECK((long long)tiles * splits < (1ll << 31), "capmi_igemm_tn_wgrad: grid too large");
    a.Np = cdiv(a.N, BNO) * BNO;
    a.Kp = cdiv(a.K, BKO) * BKO;
    a.slab = nullptr;
    static const int force_atomic = getenv("CAPMI_TN_ATOMIC") ? atoi(getenv("CAPMI_TN_ATOMIC")) : 0;  // experiment knob
    // many splits / tiny outputs: the slab reduce would be latency-bound -> f32 atomics, unless the deterministic mode asks
    // for a fixed summation order (capmi.h): then every split product goes through the slabs
    const bool use_slab = splits > 1 && ((splits <= 24 && BNO >= 128 && !force_atomic) || capmi_deterministic());
    if (use_slab) {
        CAPMI_CHECK(ws && ws_bytes >= (long long)splits * a.Np * a.Kp * 4, "capmi_igemm_tn_wgrad: workspace too small (%lld bytes needed)",
                    (long long)splits * a.Np * a.Kp * 4);
        a.slab = ws;
    }
    if constexpr (sizeof(T) == 2 && BNO == 128) {
        // The LDS footprint of this kernel decides how many workgroups of the MAIN lane fit next to it on a CU (160 KB): beside the
        // 64 KB ring of two 64-row stages the 36 KB forward / data-gradient tiles run 2 per CU instead of 4 and the 96 KB k-group
        // kernels not at all; beside 48 KB (three 32-row stages) 3 per CU resp. 1.  Alone the 48 KB form is a few per cent
        // slower, in the step it is worth 0.10 ms (8.84 -> 8.74, A/B on one box); 32 KB (two 32-row stages) starves the kernel
        // itself (+0.2 ms).  CAPMI_TN_SMALL = 0 / 1 / 2 selects 64 / 32 / 48 KB (lesson 47).
        static const int small = getenv("CAPMI_TN_SMALL") ? atoi(getenv("CAPMI_TN_SMALL")) : 2;
        if (small == 1) CAPMI_KLAUNCH((igemm_tn_glds_kernel<0, 2, 1, false, 32>), dim3(tiles * splits), dim3(512), 0, st, a);
        else if (small == 2) CAPMI_KLAUNCH((igemm_tn_glds_kernel<0, 3, 1, false, 32>), dim3(tiles * splits), dim3(512), 0, st, a);
        else CAPMI_KLAUNCH(igemm_tn_glds_kernel<0>, dim3(tiles * splits), dim3(512), 0, st, a);
    }
    else CAPMI_KLAUNCH((igemm_tn_kernel<T, BNO, BKO>), dim3(tiles * splits), dim3(256), 0, st, a);
    if (use_slab)
        CAPMI_KLAUNCH(wgrad_reduce_kernel, dim3(a.Np / 32, a.Kp / 32), dim3(256), 0, st, ws, splits, a.Np, a.Kp, a.dw, a.N, a.K, a.lddw);
    CAPMI_LAUNCH_CHECK("capmi_igemm_tn_wgrad");
    return 0;
}

extern "C" int capmi_igemm_tn_wgrad(const void* x, const void* dy, float* dw, const capmi_conv_geom* g,
                                    int N, int ldy, int lddw, float* ws, long long ws_bytes, int dtype, void* stream) {
    CAPMI_CHECK(x && dy && dw && g, "capmi_igemm_tn_wgrad: null pointer");
    const int vec = dtype == CAPMI_F32 ? 4 : 8;
    CAPMI_CHECK(g->Cin % vec == 0 && g->ldx % vec == 0 && ldy % vec == 0,
                "capmi_igemm_tn_wgrad: Cin=%d ldx=%d ldy=%d must be multiples of %d", g->Cin, g->ldx, ldy, vec);
    CAPMI_CHECK(g->up >= 1 && (g->up & (g->up - 1)) == 0, "capmi_igemm_tn_wgrad: up must be a power of two");
    WGradArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.M = g->B * g->Ho * g->Wo; a.N = N; a.K = g->kh * g->kw * g->Cin;
    a.ldy = ldy; a.lddw = lddw; a.g = *g; a.m_per_split = a.M;
    CAPMI_CHECK((long long)(a.M + 256) * g->Ho * g->Wo < (1ll << 40), "capmi_igemm_tn_wgrad: M * Ho*Wo outside the fast-division range");
    CAPMI_CHECK((long long)(g->B + 1) * g->Hi * g->Wi < (1ll << 31), "capmi_igemm_tn_wgrad: more than 2^31 input pixels");
    a.fd_hw = fast_div(g->Ho * g->Wo); a.fd_w = fast_div(g->Wo);
    a.linear = (g->kh == 1 && g->kw == 1 && g->sd == 1 && g->up == 1 && g->pad == 0 && g->Hi == g->Ho && g->Wi == g->Wo) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    int bno, bko;
    tn_tile(N, a.K, dtype, &bno, &bko);
    if (dtype == CAPMI_BF16) {
        if (bno == 128) return launch_tn<bf16, 128, 128>(a, ws, ws_bytes, st);
        return launch_tn<bf16, 64, 64>(a, ws, ws_bytes, st);
    } else if (dtype == CAPMI_F32) {
        return launch_tn<float, 64, 64>(a, ws, ws_bytes, st);
    }
    capmi_set_error("capmi_igemm_tn_wgrad: bad dtype %d", dtype);
    return 1;
}
