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
#include "common.h"

struct IGemmArgs {
    const void* x;
    const void* w;
    void* y;
    const float* bias;
    const void* addend;
    const void* ysaved;
    float* stats;
    int M, N, K;
    int ldw, ldy, ld_addend, ld_saved;
    capmi_conv_geom g;
    int act, dact, out_f32;
};

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
__device__ __forceinline__ RowPos row_pos(int m, int M, const capmi_conv_geom& g) {
    RowPos r;
    r.ok = m < M;
    int hw = g.Ho * g.Wo;
    int b = m / hw;
    int rem = m - b * hw;
    int ho = rem / g.Wo;
    int wo = rem - ho * g.Wo;
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

// ------------------------------------------------------------------ NT kernel
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_nt_kernel(IGemmArgs a) {
    constexpr int BK = 32;
    constexpr int VEC = Vec<T>::N;
    constexpr int CPR = BK / VEC;          // 16-byte chunks per tile row
    constexpr int LD = BK + VEC;           // padded LDS row, elements (16-byte aligned rows)
    constexpr int RSTEP = 256 / CPR;
    constexpr int ACH = BM / RSTEP, BCH = BN / RSTEP;
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int EPI_LD = BN + 4;                                     // f32 words per epilogue row
    constexpr int STAGE_BYTES = 2 * (BM + BN) * LD * (int)sizeof(T);
    constexpr int EPI_BYTES = (BM / 2) * EPI_LD * 4;
    constexpr int SMEM_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    T* As = reinterpret_cast<T*>(smem);                // [2][BM*LD]
    T* Bs = As + 2 * BM * LD;                          // [2][BN*LD]
    float* epi = reinterpret_cast<float*>(smem);       // [BM/2][EPI_LD], reuses the staging space

    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ W = (const T*)a.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (a.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;
    const int kc = tid % CPR, r0 = tid / CPR;

    RowPos rp[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) rp[i] = row_pos(m0 + r0 + i * RSTEP, a.M, a.g);
    KPos kp = k_pos(kc * VEC, a.g);

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
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) vstore<T>(&As[buf * BM * LD + (r0 + i * RSTEP) * LD + kc * VEC], ra[i]);
#pragma unroll
        for (int i = 0; i < BCH; ++i) vstore<T>(&Bs[buf * BN * LD + (r0 + i * RSTEP) * LD + kc * VEC], rb[i]);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (a.K + BK - 1) / BK;
    load_tile();
    store_tile(0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile();
        Frag<T> af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i].load(&As[buf * BM * LD + (wm * (BM / 2) + i * 16 + fr) * LD + fg * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j].load(&Bs[buf * BN * LD + (wn * (BN / 2) + j * 16 + fr) * LD + fg * 8]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma16(acc[i][j], af[i], bf[j]);
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue, phase 1 (registers).  C/D layout: col = lane&15, row = (lane>>4)*4 + reg.
    // bias, then the fused batch-norm statistics of this wave's BM/2-row block.
    const int wrow0 = m0 + wm * (BM / 2);                      // first row of this wave's sub-tile
    const int wcnt = min(BM / 2, a.M - wrow0);                 // valid rows in it (<= 0: none)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 16 + fr;
        const bool cok = col < a.N;
        const float bias = (a.bias && cok) ? a.bias[col] : 0.f;
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][j][r] += bias;
                if (wrow0 + i * 16 + fg * 4 + r < a.M) s1 += acc[i][j][r];
            }
        if (a.stats && wcnt > 0) {
            // exact (mean, M2 = sum (v-mean)^2) per column from the f32 accumulators; plain stores,
            // one producer per (part, column): deterministic, cancellation-free (see bn_ops.hip)
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            const float mean = s1 / (float)wcnt;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = acc[i][j][r] - mean;
                    if (wrow0 + i * 16 + fg * 4 + r < a.M) m2 += d * d;
                }
            m2 += __shfl_xor(m2, 16, 64); m2 += __shfl_xor(m2, 32, 64);
            if (fg == 0 && cok) {
                float* w = a.stats + ((int64_t)(wrow0 / (BM / 2)) * a.N + col) * 2;
                w[0] = mean;
                w[1] = m2;
            }
        }
    }
    // ---- phase 2: one BM/2-row half at a time through LDS, then 8 consecutive columns per lane:
    // addend / activation / activation-derivative applied on 16-byte vectors, 16-byte stores.
    const T* addend = (const T*)a.addend;
    const T* ysaved = (const T*)a.ysaved;
    const bool vec_ok = (a.ldy % 8 == 0) && (!addend || a.ld_addend % 8 == 0) && (!a.dact || a.ld_saved % 8 == 0);
    constexpr int CH = BN / 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) epi[(i * 16 + fg * 4 + r) * EPI_LD + wn * (BN / 2) + j * 16 + fr] = acc[i][j][r];
        }
        __syncthreads();
        for (int idx = tid; idx < (BM / 2) * CH; idx += 256) {
            const int rl = idx / CH, cch = idx - rl * CH;
            const int row = m0 + h * (BM / 2) + rl, col = n0 + cch * 8;
            if (row >= a.M || col >= a.N) continue;
            float v[8];
            load8<float>(&epi[rl * EPI_LD + cch * 8], v);
            const bool full = vec_ok && col + 8 <= a.N;
            if (full) {
                if (addend) {
                    float t[8];
                    load8<T>(addend + (int64_t)row * a.ld_addend + col, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], a.act);
                if (a.dact) {
                    float t[8];
                    load8<T>(ysaved + (int64_t)row * a.ld_saved + col, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= act_grad_from_out(t[e], a.dact);
                }
                if (a.out_f32) store8<float>((float*)a.y + (int64_t)row * a.ldy + col, v);
                else store8<T>((T*)a.y + (int64_t)row * a.ldy + col, v);
            } else {
                for (int e = 0; e < 8 && col + e < a.N; ++e) {
                    float f = v[e];
                    if (addend) f += to_f32(addend[(int64_t)row * a.ld_addend + col + e]);
                    f = apply_act(f, a.act);
                    if (a.dact) f *= act_grad_from_out(to_f32(ysaved[(int64_t)row * a.ld_saved + col + e]), a.dact);
                    if (a.out_f32) ((float*)a.y)[(int64_t)row * a.ldy + col + e] = f;
                    else ((T*)a.y)[(int64_t)row * a.ldy + col + e] = from_f32<T>(f);
                }
            }
        }
    }
}

// tile selection shared by the launcher and the statistics-workspace query
static void nt_tile(int M, int N, int dtype, int* bm, int* bn) {
    const bool wide = N > 64;
    const bool tall = (int64_t)cdiv(M, 128) * cdiv(N, wide ? 128 : 64) >= 256;
    if (dtype == CAPMI_BF16) {
        *bm = tall ? 128 : 64;
        *bn = wide ? 128 : 64;
    } else {                      // f32 tiles are twice the bytes: stay under 64 KiB of static LDS
        *bm = wide ? 64 : (tall ? 128 : 64);
        *bn = wide ? 128 : 64;
    }
}

extern "C" int capmi_igemm_nt_stats_part_rows(int M, int N, int dtype) {
    int bm, bn;
    nt_tile(M, N, dtype, &bm, &bn);
    return bm / 2;
}

template <typename T, int BM, int BN>
static int launch_nt(const IGemmArgs& a, hipStream_t st) {
    int64_t tiles = (int64_t)cdiv(a.M, BM) * cdiv(a.N, BN);
    if (tiles <= 0) return 0;
    CAPMI_CHECK(tiles < (1ll << 31), "capmi_igemm_nt: grid too large");
    hipLaunchKernelGGL((igemm_nt_kernel<T, BM, BN>), dim3((unsigned)tiles), dim3(256), 0, st, a);
    CAPMI_LAUNCH_CHECK("capmi_igemm_nt");
    return 0;
}

extern "C" int capmi_igemm_nt(const void* x, const void* w, void* y, const capmi_conv_geom* g,
                              int N, int ldw, int ldy, const float* bias, const void* addend,
                              int ld_addend, const void* ysaved, int ld_saved, float* stats,
                              int act, int dact, int out_f32, int dtype, void* stream) {
    CAPMI_CHECK(x && w && y && g, "capmi_igemm_nt: null pointer");
    const int vec = dtype == CAPMI_F32 ? 4 : 8;
    CAPMI_CHECK(g->Cin % vec == 0 && g->ldx % vec == 0 && ldw % vec == 0,
                "capmi_igemm_nt: Cin=%d ldx=%d ldw=%d must be multiples of %d", g->Cin, g->ldx, ldw, vec);
    CAPMI_CHECK(g->up >= 1 && (g->up & (g->up - 1)) == 0 && g->sd >= 1 && g->kh >= 1 && g->kw >= 1,
                "capmi_igemm_nt: bad geometry (up must be a power of two)");
    CAPMI_CHECK(!dact || ysaved, "capmi_igemm_nt: dact needs ysaved");
    CAPMI_CHECK(!(stats && addend), "capmi_igemm_nt: fused statistics and addend are mutually exclusive");
    IGemmArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.addend = addend; a.ysaved = ysaved; a.stats = stats;
    a.M = g->B * g->Ho * g->Wo; a.N = N; a.K = g->kh * g->kw * g->Cin;
    a.ldw = ldw; a.ldy = ldy; a.ld_addend = ld_addend; a.ld_saved = ld_saved;
    a.g = *g; a.act = act; a.dact = dact; a.out_f32 = out_f32;
    CAPMI_CHECK(ldw >= a.K, "capmi_igemm_nt: ldw=%d < K=%d", ldw, a.K);
    hipStream_t st = (hipStream_t)stream;
    int bm, bn;
    nt_tile(a.M, N, dtype, &bm, &bn);
    if (dtype == CAPMI_BF16) {
        if (bm == 128 && bn == 128) return launch_nt<bf16, 128, 128>(a, st);
        if (bm == 64 && bn == 128) return launch_nt<bf16, 64, 128>(a, st);
        if (bm == 128 && bn == 64) return launch_nt<bf16, 128, 64>(a, st);
        return launch_nt<bf16, 64, 64>(a, st);
    } else if (dtype == CAPMI_F32) {
        if (bm == 64 && bn == 128) return launch_nt<float, 64, 128>(a, st);
        if (bm == 128 && bn == 64) return launch_nt<float, 128, 64>(a, st);
        return launch_nt<float, 64, 64>(a, st);
    }
    capmi_set_error("capmi_igemm_nt: bad dtype %d", dtype);
    return 1;
}

// ------------------------------------------------------------------ TN kernel (weight gradient)
struct WGradArgs {
    const void* x;
    const void* dy;
    float* dw;
    int M, N, K;
    int ldy, lddw;
    int m_per_split;
    int linear;       // 1x1 / stride 1 / no padding: A(m,k) = x[m*ldx + k]
    capmi_conv_geom g;
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

template <typename T, int BNO, int BKO>
__global__ __launch_bounds__(256) void igemm_tn_kernel(WGradArgs a) {
    constexpr int RM = 32;                      // reduction rows per step
    constexpr int VEC = Vec<T>::N;
    constexpr int LDY = BNO + VEC, LDX = BKO + VEC;
    constexpr int CPY = BNO / VEC, CPX = BKO / VEC;        // chunks per row
    constexpr int YCH = RM * CPY / 256, XCH = RM * CPX / 256;
    static_assert(YCH >= 1 && XCH >= 1, "tile too small for 256 threads");
    constexpr int TN_ = BNO / 32, TK_ = BKO / 32;
    __shared__ __attribute__((aligned(16))) T Ys[2][RM * LDY];
    __shared__ __attribute__((aligned(16))) T Xs[2][RM * LDX];

    const T* __restrict__ X = (const T*)a.x;
    const T* __restrict__ DY = (const T*)a.dy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int tiles_k = (a.K + BKO - 1) / BKO;
    const int n0 = (blockIdx.x / tiles_k) * BNO;
    const int k0 = (blockIdx.x % tiles_k) * BKO;
    const int m_begin = blockIdx.y * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    if (m_begin >= m_end) return;

    const int yc = tid % CPY, yr0 = tid / CPY;      // dY chunk column / first row
    const int xc = tid % CPX, xr0 = tid / CPX;
    constexpr int YRS = 256 / CPY, XRS = 256 / CPX;
    const KPos kp = k_pos(k0 + xc * VEC, a.g);      // this thread's k: fixed for the whole kernel
    const int ncol = n0 + yc * VEC;

    Vec<T> ry[YCH], rx[XCH];
    auto load_tile = [&](int mt) {
#pragma unroll
        for (int i = 0; i < YCH; ++i) {
            int m = mt + yr0 + i * YRS;
            ry[i] = (m < m_end && ncol < a.N) ? vload<T>(DY + (int64_t)m * a.ldy + ncol) : vzero<T>();
        }
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            int m = mt + xr0 + i * XRS;
            int64_t off = -1;
            if (m < m_end) {
                if (a.linear) off = kp.k < a.K ? (int64_t)m * a.g.ldx + kp.k : -1;
                else off = a_offset(row_pos(m, a.M, a.g), kp, a.K, a.g);
            }
            rx[i] = off >= 0 ? vload<T>(X + off) : vzero<T>();
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < YCH; ++i) vstore<T>(&Ys[buf][(yr0 + i * YRS) * LDY + yc * VEC], ry[i]);
#pragma unroll
        for (int i = 0; i < XCH; ++i) vstore<T>(&Xs[buf][(xr0 + i * XRS) * LDX + xc * VEC], rx[i]);
    };

    f32x4 acc[TN_][TK_];
#pragma unroll
    for (int i = 0; i < TN_; ++i)
#pragma unroll
        for (int j = 0; j < TK_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    load_tile(m_begin);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (int mt = m_begin; mt < m_end; mt += RM) {
        const bool more = mt + RM < m_end;
        if (more) load_tile(mt + RM);
        Frag<T> af[TN_], bf[TK_];
#pragma unroll
        for (int i = 0; i < TN_; ++i) load_frag_tr(af[i], Ys[buf], LDY, fg, wn * (BNO / 2) + i * 16, fr);
#pragma unroll
        for (int j = 0; j < TK_; ++j) load_frag_tr(bf[j], Xs[buf], LDX, fg, wk * (BKO / 2) + j * 16, fr);
#pragma unroll
        for (int i = 0; i < TN_; ++i)
#pragma unroll
            for (int j = 0; j < TK_; ++j) mma16(acc[i][j], af[i], bf[j]);
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const bool single = gridDim.y == 1;           // sole contributor to its tile: plain read-modify-write
#pragma unroll
    for (int i = 0; i < TN_; ++i)
#pragma unroll
        for (int j = 0; j < TK_; ++j) {
            const int k = k0 + wk * (BKO / 2) + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * (BNO / 2) + i * 16 + fg * 4 + r;
                if (n < a.N && k < a.K) {
                    float* p = &a.dw[(int64_t)n * a.lddw + k];
                    if (single) *p += acc[i][j][r];
                    else atomicAdd(p, acc[i][j][r]);
                }
            }
        }
}

template <typename T, int BNO, int BKO>
static int launch_tn(WGradArgs& a, hipStream_t st) {
    int tiles = cdiv(a.N, BNO) * cdiv(a.K, BKO);
    if (tiles <= 0 || a.M <= 0) return 0;
    // Splits over the reduction (pixel) axis: enough workgroups to fill 256 CUs, but each split
    // at least 1024 rows deep -- every split adds its whole tile with f32 atomics, which the chip
    // retires at only ~1.3 TB/s, so a shallow split costs more in atomics than it gains in occupancy.
    int want = cdiv(512, tiles);
    int max_splits = cdiv(a.M, 1024);
    int splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    int per = cdiv(a.M, splits);
    per = (per + 31) / 32 * 32;
    splits = cdiv(a.M, per);
    a.m_per_split = per;
    hipLaunchKernelGGL((igemm_tn_kernel<T, BNO, BKO>), dim3(tiles, splits), dim3(256), 0, st, a);
    CAPMI_LAUNCH_CHECK("capmi_igemm_tn_wgrad");
    return 0;
}

extern "C" int capmi_igemm_tn_wgrad(const void* x, const void* dy, float* dw, const capmi_conv_geom* g,
                                    int N, int ldy, int lddw, int dtype, void* stream) {
    CAPMI_CHECK(x && dy && dw && g, "capmi_igemm_tn_wgrad: null pointer");
    const int vec = dtype == CAPMI_F32 ? 4 : 8;
    CAPMI_CHECK(g->Cin % vec == 0 && g->ldx % vec == 0 && ldy % vec == 0,
                "capmi_igemm_tn_wgrad: Cin=%d ldx=%d ldy=%d must be multiples of %d", g->Cin, g->ldx, ldy, vec);
    CAPMI_CHECK(g->up >= 1 && (g->up & (g->up - 1)) == 0, "capmi_igemm_tn_wgrad: up must be a power of two");
    WGradArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.M = g->B * g->Ho * g->Wo; a.N = N; a.K = g->kh * g->kw * g->Cin;
    a.ldy = ldy; a.lddw = lddw; a.g = *g; a.m_per_split = a.M;
    a.linear = (g->kh == 1 && g->kw == 1 && g->sd == 1 && g->up == 1 && g->pad == 0 && g->Hi == g->Ho && g->Wi == g->Wo) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool big = N >= 128 && a.K >= 128;
    if (dtype == CAPMI_BF16) {
        if (big) return launch_tn<bf16, 128, 128>(a, st);
        return launch_tn<bf16, 64, 64>(a, st);
    } else if (dtype == CAPMI_F32) {
        return launch_tn<float, 64, 64>(a, st);
    }
    capmi_set_error("capmi_igemm_tn_wgrad: bad dtype %d", dtype);
    return 1;
}
