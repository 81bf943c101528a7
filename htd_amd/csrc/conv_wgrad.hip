// Weight gradient of the NHWC convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
//
//   gw[co][kh][kw][ci] = sum over pixels k = (b, ho, wo) of gy[k][co] * x[b][ho*s-p+kh*d][wo*s-p+kw*d][ci]
//
// GEMM view: M = Co, N = kh*kw*Ci, reduction K = B*Ho*Wo (10^5..10^6).  Both operands are "K-major" in memory
// (a pixel row holds all channels contiguously), so slices are staged to LDS as [k][m] / [k][n] with 16-byte
// coalesced loads; MFMA block j of a wave takes the interleaved rows TM*lane + j, so a lane's operands are adjacent
// in the [k][m] image and arrive with one conflict-free ds_read_b64.  The reduction is split over `splits`
// independent workgroups per output tile (split-K); partial tiles go to a workspace and a second kernel sums them in
// a fixed order, so the result is bitwise reproducible (no float atomics).  The tiles of the first N column also
// add up the gy elements they stage: the bias gradient is a by-product of the same launch.
#include <algorithm>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// value select (a ternary between two float4 lvalues would select between ADDRESSES and push both to scratch)
__device__ __forceinline__ float4 keep4(bool ok, float4 v)
{
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

struct WgradParams {
    const float *x, *gy;
    float *out;            // gw if splits == 1 else workspace [splits][Co][Ntot]
    float *bias_out;       // NULL, or column sums of gy: gbias if splits == 1 else workspace [splits][Co]
    int B, H, W, Ci, Co, kh, kw, stride, pad, dil, Ho, Wo;
    int64_t K;             // B*Ho*Wo
    int Ntot;              // kh*kw*Ci
    int mt, nt, splits;
    int64_t slices_per_split;
    // H2 arithmetic (conv_x3.hip, h2_scale): device scalars holding max |x| and max |gy| of the two tensors; both non-NULL: two fp16
    // pieces per operand scaled by powers of two from these maxima, three products per block, partial sums scaled back
    const float *amax_x, *amax_g;
};

constexpr int BKW = 32;    // pixels per K slice

// how a staged pixel row finds its x address
enum : int {
    PIX_POINTWISE = 0,     // 1x1, stride 1, no padding: x row == pixel index (also every Linear layer)
    PIX_WIDE = 1,          // Wo >= BKW: (b, ho, wo) advanced with at most one carry per slice, no divisions
    PIX_GENERAL = 2        // narrow maps: decode the pixel index every slice
};

// WGM x WGN waves, each wave TM x TN MFMA blocks of 32x32.  COVEC: Co % 4 == 0 (16-byte gy loads).
// The staging code is branch-free (selects only): a branch in front of a global load stops the scheduler from
// issuing the slice's eight loads back to back ahead of the MFMA phase.
template <int WGM, int WGN, int TM, int TN, int PIX, bool COVEC>
__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(WgradParams p)
{
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int SA = BM + 4, SB = BN + 4;                 // LDS row strides (floats), +16 B pad
    constexpr int VA = BM / 4, VB = BN / 4;                 // float4 per row
    constexpr int PA = (BKW * VA + 255) / 256, PB = (BKW * VB + 255) / 256;
    // ONE LDS slice buffer + register prefetch: the next slice's global loads are in flight during the whole
    // MFMA phase of the current one (64 MFMAs per wave), then written between two barriers.  Half the LDS of a
    // double buffer => twice the resident workgroups, which is what hides the barrier bubbles.
    __shared__ __attribute__((aligned(16))) float lds[1][BKW * (SA + SB)];

    // XCD-aware order: all output tiles of one K split run on the same XCD (block ids congruent mod 8 share an
    // XCD), so the gy / x pixel slices of that split are fetched into ONE L2 and reused by every tile.
    int tile, split;
    {
        const int tiles = p.mt * p.nt, L = blockIdx.x;
        if (p.splits % 8 == 0) {
            const int xcd = L % 8, idx = L / 8;
            tile = idx % tiles;
            split = (idx / tiles) * 8 + xcd;
        } else {
            tile = L % tiles;
            split = L / tiles;
        }
    }
    const int tile_m = tile % p.mt, tile_n = tile / p.mt;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // fixed per-thread column chunks
    int a_col[PA], a_row[PA];
    bool a_cok[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int v = tid + i * 256;
        a_row[i] = v / VA;
        a_col[i] = (v % VA) * 4;
        a_cok[i] = (v < BKW * VA) && (m0 + a_col[i] < p.Co);
    }
    int b_col[PB], b_row[PB], b_dy[PB], b_dx[PB], b_ci[PB];
    bool b_cok[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int v = tid + i * 256;
        b_row[i] = v / VB;
        b_col[i] = (v % VB) * 4;
        const int n = n0 + b_col[i];
        b_cok[i] = (v < BKW * VB) && (n < p.Ntot);
        const int tap = b_cok[i] ? n / p.Ci : 0;
        b_ci[i] = b_cok[i] ? n - tap * p.Ci : 0;
        b_dy[i] = (tap / p.kw) * p.dil - p.pad;
        b_dx[i] = (tap % p.kw) * p.dil - p.pad;
    }

    const int64_t total_slices = (p.K + BKW - 1) / BKW;
    const int64_t s_begin = (int64_t)split * p.slices_per_split;
    const int64_t s_end = min(total_slices, s_begin + p.slices_per_split);

    // pixel coordinates of every staged B row, advanced by BKW pixels per slice without 64-bit divisions
    unsigned a_k[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) a_k[i] = (unsigned)(s_begin * BKW) + a_row[i];
    unsigned b_k[PB];
    int b_b[PB], b_ho[PB], b_wo[PB];
    auto decode = [&](int i) {
        const unsigned kk = b_k[i] < (unsigned)p.K ? b_k[i] : 0u;
        b_wo[i] = (int)(kk % (unsigned)p.Wo);
        const unsigned t = kk / (unsigned)p.Wo;
        b_ho[i] = (int)(t % (unsigned)p.Ho);
        b_b[i] = (int)(t / (unsigned)p.Ho);
    };
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        b_k[i] = (unsigned)(s_begin * BKW) + b_row[i];
        if constexpr (PIX != PIX_POINTWISE) decode(i);
    }

    float4 ra[PA], rb[PB];
    unsigned ra_ok = 0u, rb_ok = 0u;
    auto load_slice = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const bool ok = a_cok[i] && a_k[i] < (unsigned)p.K;
            const unsigned off = ok ? a_k[i] * (unsigned)p.Co + m0 + a_col[i] : 0u;
            if constexpr (COVEC) {
                ra[i] = *reinterpret_cast<const float4 *>(p.gy + off);
            } else {              // Co not a multiple of 4 (RPN / class heads): scalar loads, clamped inside the row
                const int rem = ok ? p.Co - (m0 + a_col[i]) : 1;
                const float *g = p.gy + off;
                ra[i] = make_float4(g[0], g[rem > 1 ? 1 : 0], g[rem > 2 ? 2 : 0], g[rem > 3 ? 3 : 0]);
                ra[i].y = rem > 1 ? ra[i].y : 0.f;
                ra[i].z = rem > 2 ? ra[i].z : 0.f;
                ra[i].w = rem > 3 ? ra[i].w : 0.f;
            }
            ra_ok = ok ? (ra_ok | (1u << i)) : (ra_ok & ~(1u << i));           // zeroed at store time
            a_k[i] += BKW;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            bool ok = b_cok[i] && b_k[i] < (unsigned)p.K;
            unsigned off;
            if constexpr (PIX == PIX_POINTWISE) {
                off = b_k[i] * (unsigned)p.Ci + b_ci[i];
            } else {
                const int hi = b_ho[i] * p.stride + b_dy[i], wi = b_wo[i] * p.stride + b_dx[i];
                ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                off = (((unsigned)b_b[i] * p.H + hi) * p.W + wi) * (unsigned)p.Ci + b_ci[i];
            }
            rb[i] = *reinterpret_cast<const float4 *>(p.x + (ok ? off : 0u));
            rb_ok = ok ? (rb_ok | (1u << i)) : (rb_ok & ~(1u << i));
            b_k[i] += BKW;
            if constexpr (PIX == PIX_WIDE) {       // incremental carry: at most one wrap per step, selects only
                const int wo = b_wo[i] + BKW;
                const bool c1 = wo >= p.Wo;
                b_wo[i] = c1 ? wo - p.Wo : wo;
                const int ho = b_ho[i] + (c1 ? 1 : 0);
                const bool c2 = ho == p.Ho;
                b_ho[i] = c2 ? 0 : ho;
                b_b[i] += c2 ? 1 : 0;
            } else if constexpr (PIX == PIX_GENERAL) {
                decode(i);
            }
        }
    };
    // bias gradient = column sums of gy: the tiles of the first N column already stage every gy element of their
    // (M tile, K split) once, so they add it up on the way into LDS (saves a separate pass over gy per layer)
    const bool do_bias = p.bias_out != nullptr && tile_n == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store_slice = [&](int buf) {
        float *la = lds[buf], *lb = lds[buf] + BKW * SA;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if ((BKW * VA) % 256 == 0 || tid + i * 256 < BKW * VA) {
                const float4 v = keep4((ra_ok >> i) & 1u, ra[i]);
                *reinterpret_cast<float4 *>(la + a_row[i] * SA + a_col[i]) = v;
                if (do_bias) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
            }
#pragma unroll
        for (int i = 0; i < PB; ++i)
            if ((BKW * VB) % 256 == 0 || tid + i * 256 < BKW * VB) *reinterpret_cast<float4 *>(lb + b_row[i] * SB + b_col[i]) = keep4((rb_ok >> i) & 1u, rb[i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fidx = lane & 31, fhalf = lane >> 5;
    if (s_begin < s_end) {
        load_slice();
        store_slice(0);
    }
    __syncthreads();
    for (int64_t s = s_begin; s < s_end; ++s) {
        constexpr int cur = 0;
        if (s + 1 < s_end) load_slice();
        // Fragment reads: lane (fidx, fhalf) needs, for k = 2t + fhalf, TM rows of A and TN columns of B.  MFMA block
        // j of a wave takes the INTERLEAVED rows TM*fidx + j (not 32*j + fidx), so a lane's TM values are adjacent
        // in the [k][m] LDS image and arrive with one ds_read_b64 (TM = 2) instead of TM ds_read_b32.
        const float *la = lds[cur] + fhalf * SA + wm * TM * 32 + fidx * TM;
        const float *lb = lds[cur] + BKW * SA + fhalf * SB + wn * TN * 32 + fidx * TN;
        // fragments of k-pair t+1 are read while the MFMAs of k-pair t execute (register double buffer)
        float fa[2][TM], fb[2][TN];
        auto read_frag = [&](int t, float (&a)[TM], float (&b)[TN]) {
            if constexpr (TM == 2) {
                const float2 v = *reinterpret_cast<const float2 *>(la + 2 * t * SA);
                a[0] = v.x; a[1] = v.y;
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = la[2 * t * SA + i];
            }
            if constexpr (TN == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(lb + 2 * t * SB);
                b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
            } else if constexpr (TN == 2) {
                const float2 v = *reinterpret_cast<const float2 *>(lb + 2 * t * SB);
                b[0] = v.x; b[1] = v.y;
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = lb[2 * t * SB + j];
            }
        };
        read_frag(0, fa[0], fb[0]);
#pragma unroll
        for (int t = 0; t < BKW / 2; ++t) {
            if (t + 1 < BKW / 2) read_frag(t + 1, fa[(t + 1) & 1], fb[(t + 1) & 1]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][i], fb[t & 1][j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < s_end) {
            __syncthreads();               // every wave has read this slice
            store_slice(0);
            __syncthreads();
        }
    }

    if (do_bias) {          // threads sharing a column quad (tid % VA) fold their row partials in a fixed order
        static_assert(256 % VA == 0, "column quads repeat every VA threads");
        __syncthreads();
        float4 *red = reinterpret_cast<float4 *>(&lds[0][0]);
        red[tid] = bsum;
        __syncthreads();
        if (tid < VA) {
            float4 t = red[tid];
            for (int r = 1; r < 256 / VA; ++r) {
                const float4 v = red[r * VA + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            float *dst = p.bias_out + (int64_t)split * p.Co;
            const int m = m0 + tid * 4;
            if (m < p.Co) dst[m] = t.x;
            if (m + 1 < p.Co) dst[m + 1] = t.y;
            if (m + 2 < p.Co) dst[m + 2] = t.z;
            if (m + 3 < p.Co) dst[m + 3] = t.w;
        }
    }
    float *out = p.out + (int64_t)split * p.Co * p.Ntot;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + fidx * TN + j;                 // interleaved block columns
        if (n >= p.Ntot) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + ((r & 3) + 8 * (r >> 2) + 4 * fhalf) * TM + i;   // interleaved rows
                if (m < p.Co) out[(int64_t)m * p.Ntot + n] = acc[i][j][r];
            }
    }
}

// ---- fp32 through three-way bf16 splits on the bf16 matrix pipe (see conv_fwd.hip, "X3") ----------------------------
// Same tiling, split-K and bias-gradient scheme as conv_wgrad_kernel<2,2,2,2,PIX,true>; what differs is the LDS image
// and the MFMA loop.  Both operands are K-major in memory ([pixel][channel]).  Every staged float4 (four channels of one
// pixel) is split into three bf16 pieces and written to three [k][m] plane images with 320-byte rows; the MFMA operands
// (8 consecutive k of one channel) come out of the hardware transposing read ds_read_b64_tr_b16, two reads per
// 32 x 16 fragment and plane, and six v_mfma_f32_32x32x16_bf16 per block and 16 k carry the fp32 product.
using bf16x8w = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2w = __attribute__((ext_vector_type(2))) __bf16;
using f32x2w = __attribute__((ext_vector_type(2))) float;
using s16x4w = __attribute__((ext_vector_type(4))) short;

constexpr int X3_ROW = 160;                    // half-words per LDS row: 128 + 32 pad (rows 16 banks apart)
constexpr int X3_PLANE = BKW * X3_ROW;         // half-words per plane image

__device__ __forceinline__ void split3x2w(float a, float b, unsigned &h, unsigned &m, unsigned &l)
{
    union { bf16x2w v; unsigned u; } c;
    c.v = __builtin_convertvector(f32x2w{a, b}, bf16x2w);
    h = c.u;
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    c.v = __builtin_convertvector(f32x2w{ra, rb}, bf16x2w);
    m = c.u;
    c.v = __builtin_convertvector(f32x2w{ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u)}, bf16x2w);
    l = c.u;
}

// H2 (see conv_x3.hip): scale from a tensor's largest magnitude, and the two-piece fp16 split of two scaled floats
using f16x8w = __attribute__((ext_vector_type(8))) _Float16;
using f16x2w = __attribute__((ext_vector_type(2))) _Float16;
struct H2ScaleW { float s, inv; };
__device__ __forceinline__ H2ScaleW h2w_scale(const float *amax)
{
    const unsigned E = (__float_as_uint(*amax) >> 23) & 0xffu;
    int e = E == 0u ? 126 : (E == 255u ? 0 : 141 - (int)E);
    e = e > 126 ? 126 : e;
    return H2ScaleW{__uint_as_float((unsigned)(127 + e) << 23), __uint_as_float((unsigned)(127 - e) << 23)};
}
__device__ __forceinline__ void split2w(float a, float b, unsigned &h, unsigned &l)
{
    union { f16x2w v; unsigned u; } c;
    c.v = __builtin_convertvector(f32x2w{a, b}, f16x2w);
    h = c.u;
    const f32x2w back = __builtin_convertvector(c.v, f32x2w);
    c.v = __builtin_convertvector(f32x2w{a - back[0], b - back[1]}, f16x2w);
    l = c.u;
}
template <bool H2>
__device__ __forceinline__ f32x16 mfma_x3w(bf16x8w a, bf16x8w b, f32x16 c)
{
    if constexpr (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, a), __builtin_bit_cast(f16x8w, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8w tr_frag(const unsigned short *img, int k0, int c0, int lane)
{
    // lane l: h = l>>5 takes k0 + 8h .. +7; its 16-lane group covers columns c0 + 16*((l>>4)&1) .. +15
    const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
    const unsigned short *a = img + (k0 + 8 * h + q) * X3_ROW + c0 + 16 * g + 4 * pp;
    const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w *)a);
    const s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w *)(a + 4 * X3_ROW));
    union { struct { s16x4w a, b; } s; bf16x8w v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

template <int PIX>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(WgradParams p)
{
    constexpr int WGN = 2, TM = 2, TN = 2, BM = 128, BN = 128;
    constexpr int VA = BM / 4, VB = BN / 4, PA = BKW * VA / 256, PB = BKW * VB / 256;      // 4 float4 per thread and operand
    __shared__ __attribute__((aligned(16))) unsigned short lds[6 * X3_PLANE];
    unsigned short *la = lds, *lb = lds + 3 * X3_PLANE;

    int tile, split;
    {
        const int tiles = p.mt * p.nt, L = blockIdx.x;
        if (p.splits % 8 == 0) {
            const int xcd = L % 8, idx = L / 8;
            tile = idx % tiles;
            split = (idx / tiles) * 8 + xcd;
        } else {
            tile = L % tiles;
            split = L / tiles;
        }
    }
    const int tile_m = tile % p.mt, tile_n = tile / p.mt;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    int a_row[PA], b_row[PB];
    const int a_col = (tid % VA) * 4, b_col = (tid % VB) * 4;
    const bool a_cok = m0 + a_col < p.Co;
    const int nb = n0 + b_col;
    const bool b_cok = nb < p.Ntot;
    const int tap = b_cok ? nb / p.Ci : 0;
    const int b_ci = b_cok ? nb - tap * p.Ci : 0;
    const int b_dy = (tap / p.kw) * p.dil - p.pad, b_dx = (tap % p.kw) * p.dil - p.pad;
#pragma unroll
    for (int i = 0; i < PA; ++i) a_row[i] = (tid + i * 256) / VA;
#pragma unroll
    for (int i = 0; i < PB; ++i) b_row[i] = (tid + i * 256) / VB;

    const int64_t total_slices = (p.K + BKW - 1) / BKW;
    const int64_t s_begin = (int64_t)split * p.slices_per_split;
    const int64_t s_end = min(total_slices, s_begin + p.slices_per_split);

    unsigned a_k[PA], b_k[PB];
    int b_b[PB], b_ho[PB], b_wo[PB];
    auto decode = [&](int i) {
        const unsigned kk = b_k[i] < (unsigned)p.K ? b_k[i] : 0u;
        b_wo[i] = (int)(kk % (unsigned)p.Wo);
        const unsigned t = kk / (unsigned)p.Wo;
        b_ho[i] = (int)(t % (unsigned)p.Ho);
        b_b[i] = (int)(t / (unsigned)p.Ho);
    };
#pragma unroll
    for (int i = 0; i < PA; ++i) a_k[i] = (unsigned)(s_begin * BKW) + a_row[i];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        b_k[i] = (unsigned)(s_begin * BKW) + b_row[i];
        if constexpr (PIX != PIX_POINTWISE) decode(i);
    }

    float4 ra[PA], rb[PB];
    unsigned ra_ok = 0u, rb_ok = 0u;
    auto load_slice = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const bool ok = a_cok && a_k[i] < (unsigned)p.K;
            ra[i] = *reinterpret_cast<const float4 *>(p.gy + (ok ? a_k[i] * (unsigned)p.Co + m0 + a_col : 0u));
            ra_ok = ok ? (ra_ok | (1u << i)) : (ra_ok & ~(1u << i));
            a_k[i] += BKW;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            bool ok = b_cok && b_k[i] < (unsigned)p.K;
            unsigned off;
            if constexpr (PIX == PIX_POINTWISE) {
                off = b_k[i] * (unsigned)p.Ci + b_ci;
            } else {
                const int hi = b_ho[i] * p.stride + b_dy, wi = b_wo[i] * p.stride + b_dx;
                ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                off = (((unsigned)b_b[i] * p.H + hi) * p.W + wi) * (unsigned)p.Ci + b_ci;
            }
            rb[i] = *reinterpret_cast<const float4 *>(p.x + (ok ? off : 0u));
            rb_ok = ok ? (rb_ok | (1u << i)) : (rb_ok & ~(1u << i));
            b_k[i] += BKW;
            if constexpr (PIX == PIX_WIDE) {
                const int wo = b_wo[i] + BKW;
                const bool c1 = wo >= p.Wo;
                b_wo[i] = c1 ? wo - p.Wo : wo;
                const int ho = b_ho[i] + (c1 ? 1 : 0);
                const bool c2 = ho == p.Ho;
                b_ho[i] = c2 ? 0 : ho;
                b_b[i] += c2 ? 1 : 0;
            } else if constexpr (PIX == PIX_GENERAL) {
                decode(i);
            }
        }
    };
    const bool do_bias = p.bias_out != nullptr && tile_n == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto put = [&](unsigned short *img, int row, int col, float4 v) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split3x2w(v.x, v.y, h0, m0_, l0);
        split3x2w(v.z, v.w, h1, m1, l1);
        unsigned short *d = img + row * X3_ROW + col;
        *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(d + X3_PLANE) = make_uint2(m0_, m1);
        *reinterpret_cast<uint2 *>(d + 2 * X3_PLANE) = make_uint2(l0, l1);
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const float4 v = keep4((ra_ok >> i) & 1u, ra[i]);
            put(la, a_row[i], a_col, v);
            if (do_bias) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) put(lb, b_row[i], b_col, keep4((rb_ok >> i) & 1u, rb[i]));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (s_begin < s_end) {
        load_slice();
        store_slice();
    }
    __syncthreads();
    for (int64_t s = s_begin; s < s_end; ++s) {
        if (s + 1 < s_end) load_slice();
#pragma unroll
        for (int kk = 0; kk < BKW / 16; ++kk) {
            bf16x8w fa[TM][3], fb[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) fa[i][q] = tr_frag(la + q * X3_PLANE, kk * 16, wm * 64 + i * 32, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) fb[j][q] = tr_frag(lb + q * X3_PLANE, kk * 16, wn * 64 + j * 32, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {      // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                }
        }
        if (s + 1 < s_end) {
            __syncthreads();
            store_slice();
        }
        __syncthreads();
    }

    if (do_bias) {          // threads sharing a column quad (tid % VA) fold their row partials in a fixed order
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = bsum;
        __syncthreads();
        if (tid < VA) {
            float4 t = red[tid];
            for (int r = 1; r < 256 / VA; ++r) {
                const float4 v = red[r * VA + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            float *dst = p.bias_out + (int64_t)split * p.Co;
            const int m = m0 + tid * 4;
            if (m < p.Co) dst[m] = t.x;
            if (m + 1 < p.Co) dst[m + 1] = t.y;
            if (m + 2 < p.Co) dst[m + 2] = t.z;
            if (m + 3 < p.Co) dst[m + 3] = t.w;
        }
    }
    float *out = p.out + (int64_t)split * p.Co * p.Ntot;
    const int fcol = lane & 31, fhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 64 + j * 32 + fcol;
            if (n >= p.Ntot) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (m < p.Co) out[(int64_t)m * p.Ntot + n] = acc[i][j][r];
            }
        }
}

// ---- the same product with the split of slice s+1 in the shadow of the MFMAs of slice s (round 3) -----------------------------
// conv_wgrad_x3_kernel alternates phases: 48 MFMAs per wave on the staged slice, barrier, ~250 vector instructions that split
// the next slice into its planes, barrier -- the matrix pipe idles through every split phase (105-125 TF/s on the 1x1 layers,
// where no tap shares a staged run).  Here a slice is 16 pixels, LDS holds TWO of them (the same 61 KB), and the loop body is
// one basic block with one barrier: split + store of slice s+1 into the other buffer, the global loads of slice s+3 into the
// registers just freed (slice s+2 is in flight in the second register set), 24 MFMAs per wave on slice s.  Vector and matrix
// instructions of the same wave now interleave, and the loads have two slices of time to land.  Slices past the end of the
// split's K range stage zeros (no branches in the body).  Same tiles, splits, partial layout and summation order as
// conv_wgrad_x3_kernel: the results are bit-identical.
constexpr int XD_KS = 16;                      // pixels per K slice
constexpr int XD_PLANE = XD_KS * X3_ROW;       // half-words per plane image

// TM x TN 32x32 blocks per wave (2 x 2 waves): 128x128 tiles, or 64x128 / 128x64 for layers with 64 output or 64 reduction-side
// channels (layer1, the stem), where half of a 128-wide tile would multiply zeros.  p.mt / p.nt count tiles of THIS shape; the
// split count and the K ranges are those of the 128x128 form (same summation order, same bits).
template <int PIX, int TM, int TN, bool H2 = false>
__global__ __launch_bounds__(256, (H2 && PIX == PIX_POINTWISE ? 3 : 2)) void conv_wgrad_x3d_kernel(WgradParams p)
{
    constexpr int NPL = H2 ? 2 : 3;             // planes per operand: H2 keeps two, 40 KB of LDS, three workgroups per CU
    constexpr int WGN = 2, BM = 2 * TM * 32, BN = 2 * TN * 32;
    constexpr int VA = BM / 4, VB = BN / 4, PA = XD_KS * VA / 256, PB = XD_KS * VB / 256;      // float4 per thread, operand and slice
    static_assert(PA >= 1 && PB >= 1 && (H2 || (6 * TM * TN) % (2 * (PA + PB)) == 0), "whole MFMAs per split chunk");
    float sg = 1.f, sx = 1.f, sg_inv = 1.f, sx_inv = 1.f;          // H2: power-of-two scales of the two operands
    if constexpr (H2) {
        const H2ScaleW a = h2w_scale(p.amax_g), b = h2w_scale(p.amax_x);
        sg = a.s; sg_inv = a.inv; sx = b.s; sx_inv = b.inv;
    }
    constexpr int XD_BUFX = 2 * NPL * XD_PLANE;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * XD_BUFX];

    // Workgroup -> (tile, split).  Consecutive workgroups go to consecutive XCDs, each with its own L2:
    //   splits % 8 == 0: XCD x owns the splits = x (mod 8) -- all tiles of a K range on one XCD, both operands' rows of that
    //                    range fetched once;
    //   otherwise (few splits, many tiles: the FC layers): the mt tiles of one (split, N tile) group -- they read the same x
    //                    columns -- on ONE XCD, back to back; groups dealt round-robin over the XCDs.  With the plain order
    //                    tile_m = workgroup % 8 put the eight readers of an x tile on eight different XCDs (FC1: 815 MB
    //                    fetched for 222 MB of operands, L2 hit rate 48 %).  Workgroups past the last group exit.
    int tile_m, tile_n, split;
    {
        const int tiles = p.mt * p.nt, L = blockIdx.x;
        const int xcd = L % 8, idx = L / 8;
        if (p.splits % 8 == 0) {
            const int tile = idx % tiles;
            split = (idx / tiles) * 8 + xcd;
            tile_m = tile % p.mt;
            tile_n = tile / p.mt;
        } else {
            const int g = xcd + 8 * (idx / p.mt);
            if (g >= p.nt * p.splits) return;
            tile_m = idx % p.mt;
            split = g / p.nt;
            tile_n = g % p.nt;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    int a_row[PA], b_row[PB];
    const int a_col = (tid % VA) * 4, b_col = (tid % VB) * 4;
    const bool a_cok = m0 + a_col < p.Co;
    const int nb = n0 + b_col;
    const bool b_cok = nb < p.Ntot;
    const int tap = b_cok ? nb / p.Ci : 0;
    const int b_ci = b_cok ? nb - tap * p.Ci : 0;
    const int b_dy = (tap / p.kw) * p.dil - p.pad, b_dx = (tap % p.kw) * p.dil - p.pad;
#pragma unroll
    for (int i = 0; i < PA; ++i) a_row[i] = (tid + i * 256) / VA;
#pragma unroll
    for (int i = 0; i < PB; ++i) b_row[i] = (tid + i * 256) / VB;

    // the split's K range in pixels: [k_begin, k_end), the slice boundaries of conv_wgrad_x3_kernel (BKW pixels each)
    const int64_t k_begin64 = (int64_t)split * p.slices_per_split * BKW;
    const unsigned k_begin = (unsigned)k_begin64;
    const unsigned k_end = (unsigned)min(p.K, k_begin64 + p.slices_per_split * BKW);
    const int n_slices = k_begin64 < p.K ? (int)((k_end - k_begin + XD_KS - 1) / XD_KS) : 0;

    // Operands come through buffer descriptors: a load whose byte offset lies past the descriptor's size returns zeros, so rows
    // past the split's K range (the gy descriptor -- and for 1x1 layers the x descriptor -- ends at pixel k_end), columns past
    // Co / Ntot and padding pixels (offset OOB) need no select on the data, no validity bits and no 64-bit address arithmetic:
    // one vector add per load in the 1x1 form (4.5 vector instructions per MFMA instead of 6.8, PMC r03).
    constexpr unsigned OOB = 0x80000000u;          // the host keeps both operands below 2^31 bytes
    const __amdgpu_buffer_rsrc_t gy_desc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.gy), 0, (int)((unsigned)k_end * (unsigned)p.Co * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_desc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.x), 0,
        (int)(PIX == PIX_POINTWISE ? (unsigned)k_end * (unsigned)p.Ci * 4u : (unsigned)p.B * p.H * p.W * p.Ci * 4u), 0x00020000);
    unsigned a_off[PA], b_k[PB], b_off[PB];
    const unsigned a_step = a_cok ? XD_KS * (unsigned)p.Co * 4u : 0u, b_step = b_cok ? XD_KS * (unsigned)p.Ci * 4u : 0u;
    int b_b[PB], b_ho[PB], b_wo[PB];
    auto decode = [&](int i) {
        const unsigned kk = b_k[i] < (unsigned)p.K ? b_k[i] : 0u;
        b_wo[i] = (int)(kk % (unsigned)p.Wo);
        const unsigned t = kk / (unsigned)p.Wo;
        b_ho[i] = (int)(t % (unsigned)p.Ho);
        b_b[i] = (int)(t / (unsigned)p.Ho);
    };
#pragma unroll
    for (int i = 0; i < PA; ++i) a_off[i] = a_cok ? ((k_begin + a_row[i]) * (unsigned)p.Co + m0 + a_col) * 4u : OOB;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        b_k[i] = k_begin + b_row[i];
        b_off[i] = b_cok ? (b_k[i] * (unsigned)p.Ci + b_ci) * 4u : OOB;      // the 1x1 form; the others work from b_k
        if constexpr (PIX != PIX_POINTWISE) decode(i);
    }

    float4 ra[2][PA], rb[2][PB];
    using u32x4w = __attribute__((ext_vector_type(4))) unsigned;
    auto load16 = [&](__amdgpu_buffer_rsrc_t d, unsigned off) {
        union { u32x4w u; float4 f; } c;
        c.u = __builtin_amdgcn_raw_buffer_load_b128(d, (int)off, 0, 0);
        return c.f;
    };
    auto load_slice = [&](float4 (&qa)[PA], float4 (&qb)[PB]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            qa[i] = load16(gy_desc, a_off[i]);
            a_off[i] += a_step;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if constexpr (PIX == PIX_POINTWISE) {
                qb[i] = load16(x_desc, b_off[i]);
                b_off[i] += b_step;
            } else {
                const int hi = b_ho[i] * p.stride + b_dy, wi = b_wo[i] * p.stride + b_dx;
                const bool ok = b_cok && b_k[i] < k_end && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const unsigned off = ((((unsigned)b_b[i] * p.H + hi) * p.W + wi) * (unsigned)p.Ci + b_ci) * 4u;
                qb[i] = load16(x_desc, off | (ok ? 0u : OOB));
                b_k[i] += XD_KS;
                if constexpr (PIX == PIX_WIDE) {            // Wo >= BKW > XD_KS: at most one carry
                    const int wo = b_wo[i] + XD_KS;
                    const bool c1 = wo >= p.Wo;
                    b_wo[i] = c1 ? wo - p.Wo : wo;
                    const int ho = b_ho[i] + (c1 ? 1 : 0);
                    const bool c2 = ho == p.Ho;
                    b_ho[i] = c2 ? 0 : ho;
                    b_b[i] += c2 ? 1 : 0;
                } else {
                    decode(i);
                }
            }
        }
    };
    const bool do_bias = p.bias_out != nullptr && tile_n == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto put = [&](unsigned short *img, int row, int col, float4 v, float sc) {
        unsigned short *d = img + row * X3_ROW + col;
        if constexpr (H2) {
            unsigned h0, l0, h1, l1;
            split2w(v.x * sc, v.y * sc, h0, l0);
            split2w(v.z * sc, v.w * sc, h1, l1);
            *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(d + XD_PLANE) = make_uint2(l0, l1);
        } else {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3x2w(v.x, v.y, h0, m0_, l0);
            split3x2w(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(d + XD_PLANE) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2 *>(d + 2 * XD_PLANE) = make_uint2(l0, l1);
        }
    };
    auto store_slice = [&](unsigned short *buf, const float4 (&qa)[PA], const float4 (&qb)[PB]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const float4 v = qa[i];
            put(buf, a_row[i], a_col, v, sg);
            bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;      // every tile (a branch would cut the loop body in two)
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) put(buf + NPL * XD_PLANE, b_row[i], b_col, qb[i], sx);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one half of the loop body: the MFMAs of the slice in `rd` with the split of the staged registers (the next slice but
    // one of that buffer's parity) into `wr` between them -- chunks of one split3x2w (two channels, ~12 vector
    // instructions) and MPC MFMAs (eight chunks of three for the 128x128 tile), pinned in this order by sched_barrier: left to itself hipcc moves all MFMAs behind
    // all of the vector work, and the matrix pipe idles through the split as it did in conv_wgrad_x3_kernel
    auto half = [&](const unsigned short *rd, unsigned short *wr, float4 (&qa)[PA], float4 (&qb)[PB]) {
        bf16x8w fa[TM][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NPL; ++q) fa[i][q] = tr_frag(rd + q * XD_PLANE, 0, wm * TM * 32 + i * 32, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) fb[j][q] = tr_frag(rd + (NPL + q) * XD_PLANE, 0, wn * TN * 32 + j * 32, lane);
        __builtin_amdgcn_sched_barrier(0);
        // smallest terms first, as conv_wgrad_x3_kernel; H2: a1 b0, a0 b1, a0 b0
        constexpr int QA[6] = {H2 ? 1 : 2, 0, H2 ? 0 : 1, 1, 0, 0}, QB[6] = {0, H2 ? 1 : 2, H2 ? 0 : 1, 0, 1, 0};
        unsigned h[2], m[2], l[2];
        constexpr int NC = 2 * (PA + PB), NM = (H2 ? 3 : 6) * TM * TN;       // split chunks; MFMAs of the slice
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int u = c >> 1;                    // staged float4: gy rows first, then x rows
            const bool is_a = u < PA;
            const float4 v = is_a ? qa[u] : qb[u - PA];
            const float sc = is_a ? sg : sx;
            if ((c & 1) == 0) {
                if constexpr (H2) split2w(v.x * sc, v.y * sc, h[0], l[0]);
                else split3x2w(v.x, v.y, h[0], m[0], l[0]);
            } else {
                if constexpr (H2) split2w(v.z * sc, v.w * sc, h[1], l[1]);
                else split3x2w(v.z, v.w, h[1], m[1], l[1]);
                unsigned short *d = (is_a ? wr + a_row[u] * X3_ROW + a_col : wr + NPL * XD_PLANE + b_row[u - PA] * X3_ROW + b_col);
                *reinterpret_cast<uint2 *>(d) = make_uint2(h[0], h[1]);
                if constexpr (H2) {
                    *reinterpret_cast<uint2 *>(d + XD_PLANE) = make_uint2(l[0], l[1]);
                } else {
                    *reinterpret_cast<uint2 *>(d + XD_PLANE) = make_uint2(m[0], m[1]);
                    *reinterpret_cast<uint2 *>(d + 2 * XD_PLANE) = make_uint2(l[0], l[1]);
                }
                if (is_a) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }     // every tile: no branch in the body
            }
#pragma unroll
            for (int t = c * NM / NC; t < (c + 1) * NM / NC; ++t) {          // MFMA t: product q of block (i, j)
                const int q = t / (TM * TN), i = (t % (TM * TN)) / TN, j = t % TN;
                acc[i][j] = mfma_x3w<H2>(fa[i][QA[q]], fb[j][QB[q]], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        load_slice(qa, qb);
        __syncthreads();
    };

    // set 0: slices 0, 2, 4 ...; set 1: slices 1, 3, 5 ...  (loads past k_end return zeros)
    unsigned short *buf0 = lds, *buf1 = lds + XD_BUFX;
    load_slice(ra[0], rb[0]);
    load_slice(ra[1], rb[1]);
    store_slice(buf0, ra[0], rb[0]);
    load_slice(ra[0], rb[0]);
    __syncthreads();
    for (int s = 0; s < n_slices; s += 2) {
        half(buf0, buf1, ra[1], rb[1]);           // MFMAs of slice s, slice s+1 staged, slice s+3 requested
        half(buf1, buf0, ra[0], rb[0]);           // slice s+1 (zeros when n_slices is odd), s+2 staged, s+4 requested
    }

    if (do_bias) {          // threads sharing a column quad (tid % VA) fold their row partials in a fixed order
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = bsum;
        __syncthreads();
        if (tid < VA) {
            float4 t = red[tid];
            for (int r = 1; r < 256 / VA; ++r) {
                const float4 v = red[r * VA + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            float *dst = p.bias_out + (int64_t)split * p.Co;
            const int m = m0 + tid * 4;
            if (m < p.Co) dst[m] = t.x;
            if (m + 1 < p.Co) dst[m + 1] = t.y;
            if (m + 2 < p.Co) dst[m + 2] = t.z;
            if (m + 3 < p.Co) dst[m + 3] = t.w;
        }
    }
    float *out = p.out + (int64_t)split * p.Co * p.Ntot;
    const int fcol = lane & 31, fhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 32 + fcol;
            if (n >= p.Ntot) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (m < p.Co) out[(int64_t)m * p.Ntot + n] = H2 ? acc[i][j][r] * sg_inv * sx_inv : acc[i][j][r];
            }
        }
}

// ---- 3x3 layers: the three taps of a filter ROW in one workgroup (round 3) ------------------------------------------------------
// conv_wgrad_x3_kernel gives every (128 co x 128 n) tile its own copy of the staging work: a gy element is loaded and split
// once per N tile (18 times for a 3x3 256 -> 256 layer), an x element once per M tile and tap.  Here a workgroup owns
// 128 co x 64 ci x the THREE taps kx of one filter row ky: per K slice the gy rows are staged and split once for three taps, and
// the x operand is ONE run of 32 + 2 pixels read at row shifts 0 / 1 / 2 by the transposing LDS reads -- 6.1 staged float4
// per thread and 72 MFMAs per wave and slice instead of 8 and 48 (the split and its address arithmetic were 7.6 vector
// instructions per MFMA, PMC r03).  Borders need no masks: the reduction runs over a VIRTUAL pixel index with one padding
// column per image row (W + 1 columns, the last one zero in both operands), so the neighbours of the first / last pixel of a
// row are zeros by construction; rows above / below the image are zeroed when the x run is staged (ky is fixed per workgroup).
constexpr int XH_KS = 32;                       // virtual pixels per K slice
constexpr int XH_RUN = XH_KS + 2;               // x rows staged per slice
constexpr int XH_ROWB = 96;                     // half-words per LDS row of the x run: 64 + 32 pad (rows 48 banks apart)
constexpr int XH_PLANE_A = XH_KS * X3_ROW, XH_PLANE_B = XH_RUN * XH_ROWB;

template <int ROW>
__device__ __forceinline__ bf16x8w tr_frag_row(const unsigned short *img, int k0, int c0, int lane)
{
    const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
    const unsigned short *a = img + (k0 + 8 * h + q) * ROW + c0 + 16 * g + 4 * pp;
    const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w *)a);
    const s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w *)(a + 4 * ROW));
    union { struct { s16x4w a, b; } s; bf16x8w v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

// WIDE: W + 1 >= 32 (a slice advances a row's coordinates by at most one carry), else the pixel is decoded per slice
template <bool WIDE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3h_kernel(WgradParams p)
{
    constexpr int BM = 128, BNC = 64;
    __shared__ __attribute__((aligned(16))) unsigned short lds[3 * XH_PLANE_A + 3 * XH_PLANE_B];
    unsigned short *la = lds, *lb = lds + 3 * XH_PLANE_A;
    const int ntc = (p.Ci + BNC - 1) / BNC;
    const int tiles = p.mt * ntc * 3;
    int tile, split;
    {
        const int L = blockIdx.x;
        if (p.splits % 8 == 0) {                 // the splits of a tile spread over the XCD groups, its tiles share one
            const int xcd = L % 8, idx = L / 8;
            tile = idx % tiles;
            split = (idx / tiles) * 8 + xcd;
        } else {
            tile = L % tiles;
            split = L / tiles;
        }
    }
    const int tile_m = tile % p.mt, rest = tile / p.mt;
    const int tile_nc = rest % ntc, ky = rest / ntc;
    const int m0 = tile_m * BM, ci0 = tile_nc * BNC, dy = ky - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int Wv = p.W + 1;                                   // virtual row length
    const int64_t V = (int64_t)p.B * p.H * Wv;
    const int64_t total_slices = (V + XH_KS - 1) / XH_KS;
    const int64_t s_begin = (int64_t)split * p.slices_per_split;
    const int64_t s_end = min(total_slices, s_begin + p.slices_per_split);

    // staging: gy 32 rows x 32 float4 (4 per thread), x run 34 rows x 16 float4 (3 passes, the last one rows 32, 33 only)
    const int a_col = (tid & 31) * 4, a_row0 = tid >> 5;      // rows a_row0 + 8 i
    const int b_col = (tid & 15) * 4, b_row0 = tid >> 4;      // rows b_row0 + 16 i
    const bool a_cok = m0 + a_col < p.Co, b_cok = ci0 + b_col < p.Ci;
    // coordinates of every staged row in the virtual index: image-row pixel x in [0, Wv), row y, image b
    int ax[4], ay[4], ab[4], bx[3], by[3], bb[3];
    auto decode = [&](int64_t v, int &x, int &y, int &b) {
        const int64_t vv = v < 0 ? 0 : v;
        x = (int)(vv % Wv);
        const int64_t t = vv / Wv;
        y = (int)(t % p.H);
        b = (int)(t / p.H);
        if (v < 0) { x = Wv - 1; y = -1; }                     // v = -1, the entry before the first pixel: padding column of "row -1"
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) decode(s_begin * XH_KS + a_row0 + 8 * i, ax[i], ay[i], ab[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) decode(s_begin * XH_KS - 1 + b_row0 + 16 * i, bx[i], by[i], bb[i]);
    auto advance = [&](int &x, int &y, int &b) {              // + XH_KS virtual pixels
        if constexpr (WIDE) {
            x += XH_KS;
            const bool c1 = x >= Wv;
            x = c1 ? x - Wv : x;
            y += c1 ? 1 : 0;
            const bool c2 = y >= p.H;
            y = c2 ? y - p.H : y;
            b += c2 ? 1 : 0;
        } else {
            const int64_t v = ((int64_t)b * p.H + y) * Wv + x + XH_KS;
            x = (int)(v % Wv);
            const int64_t t = v / Wv;
            y = (int)(t % p.H);
            b = (int)(t / p.H);
        }
    };

    float4 ra[4], rb[3];
    unsigned ok_a = 0u, ok_b = 0u;
    auto load_slice = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = a_cok && ab[i] < p.B && ax[i] < p.W;
            const unsigned off = (((unsigned)ab[i] * p.H + ay[i]) * (unsigned)p.W + ax[i]) * (unsigned)p.Co + m0 + a_col;
            ra[i] = *reinterpret_cast<const float4 *>(p.gy + (ok ? off : 0u));
            ok_a = ok ? (ok_a | (1u << i)) : (ok_a & ~(1u << i));
            advance(ax[i], ay[i], ab[i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int yy = by[i] + dy;
            const bool ok = b_cok && (i < 2 || b_row0 < 2) && bb[i] < p.B && bx[i] < p.W && (unsigned)yy < (unsigned)p.H;
            const unsigned off = (((unsigned)bb[i] * p.H + yy) * (unsigned)p.W + bx[i]) * (unsigned)p.Ci + ci0 + b_col;
            rb[i] = *reinterpret_cast<const float4 *>(p.x + (ok ? off : 0u));
            ok_b = ok ? (ok_b | (1u << i)) : (ok_b & ~(1u << i));
            advance(bx[i], by[i], bb[i]);
        }
    };
    const bool do_bias = p.bias_out != nullptr && tile_nc == 0 && ky == 1;       // the centre row sees every gy row once
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto put = [&](unsigned short *img, int plane, int rowpitch, int row, int col, float4 v) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split3x2w(v.x, v.y, h0, m0_, l0);
        split3x2w(v.z, v.w, h1, m1, l1);
        unsigned short *d = img + row * rowpitch + col;
        *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(d + plane) = make_uint2(m0_, m1);
        *reinterpret_cast<uint2 *>(d + 2 * plane) = make_uint2(l0, l1);
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = keep4((ok_a >> i) & 1u, ra[i]);
            put(la, XH_PLANE_A, X3_ROW, a_row0 + 8 * i, a_col, v);
            if (do_bias) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < 2 || b_row0 < 2) put(lb, XH_PLANE_B, XH_ROWB, b_row0 + 16 * i, b_col, keep4((ok_b >> i) & 1u, rb[i]));
    };

    f32x16 acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;

    if (s_begin < s_end) {
        load_slice();
        store_slice();
    }
    __syncthreads();
    for (int64_t s = s_begin; s < s_end; ++s) {
        if (s + 1 < s_end) load_slice();
#pragma unroll
        for (int kk = 0; kk < XH_KS / 16; ++kk) {
            bf16x8w fa[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) fa[i][q] = tr_frag_row<X3_ROW>(la + q * XH_PLANE_A, kk * 16, wm * 64 + i * 32, lane);
#pragma unroll
            for (int t = 0; t < 3; ++t) {          // tap kx = t reads the run one row further: gy row k meets x row k + t
                bf16x8w fb[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) fb[q] = tr_frag_row<XH_ROWB>(lb + q * XH_PLANE_B, kk * 16 + t, wn * 32, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i) {      // smallest terms first
                    acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[0], acc[t][i], 0, 0, 0);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[2], acc[t][i], 0, 0, 0);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[1], acc[t][i], 0, 0, 0);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[0], acc[t][i], 0, 0, 0);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[1], acc[t][i], 0, 0, 0);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[0], acc[t][i], 0, 0, 0);
                }
            }
        }
        if (s + 1 < s_end) {
            __syncthreads();
            store_slice();
        }
        __syncthreads();
    }

    if (do_bias) {          // threads sharing a column quad (tid & 31) fold their 8 row partials in a fixed order
        float4 *red = reinterpret_cast<float4 *>(lds);
        red[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            float4 t = red[tid];
            for (int r = 1; r < 8; ++r) {
                const float4 v = red[r * 32 + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            float *dst = p.bias_out + (int64_t)split * p.Co;
            const int m = m0 + tid * 4;
            if (m < p.Co) dst[m] = t.x;
            if (m + 1 < p.Co) dst[m + 1] = t.y;
            if (m + 2 < p.Co) dst[m + 2] = t.z;
            if (m + 3 < p.Co) dst[m + 3] = t.w;
        }
    }
    float *out = p.out + (int64_t)split * p.Co * p.Ntot;
    const int fcol = lane & 31, fhalf = lane >> 5;
    const int ci = ci0 + wn * 32 + fcol;
    if (ci < p.Ci) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int n = (ky * 3 + t) * p.Ci + ci;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    if (m < p.Co) out[(int64_t)m * p.Ntot + n] = acc[t][i][r];
                }
        }
    }
}

// ---- the tap-fused 3x3 kernel with the split in the shadow of the MFMAs (round 3) -------------------------------------------------
// conv_wgrad_x3h_kernel in the loop form of conv_wgrad_x3d_kernel: slices of 16 virtual pixels, two of them in LDS (51 KB), one
// barrier per slice, the split of slice s+1 cut into chunks between the 36 MFMAs of slice s, operands through buffer
// descriptors.  A gy row's byte offset advances by a constant per slice, less one pixel for every image row it crosses (the
// virtual index has W + 1 columns, memory has W): no coordinates beyond the column are tracked for gy; the x run tracks the image
// row as well (rows above / below the image are zeros).  The x run of a slice is 16 + 2 rows: 16 x 16 float4 for the 256 threads
// and a ragged pass of 2 x 16 for the first 32.  Any map on which a slice spans at most H image rows; same tiles, splits and summation order as
// conv_wgrad_x3h_kernel: bit-identical results.
constexpr int XHD_KS = 16;
constexpr int XHD_RUN = XHD_KS + 2;
constexpr int XHD_PLANE_A = XHD_KS * X3_ROW, XHD_PLANE_B = XHD_RUN * XH_ROWB;
constexpr int XHD_BUF = 3 * XHD_PLANE_A + 3 * XHD_PLANE_B;
template <int TM, bool H2 = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3hd_kernel(WgradParams p)
{
    float sg = 1.f, sx = 1.f, sg_inv = 1.f, sx_inv = 1.f;          // H2: power-of-two scales of the two operands
    if constexpr (H2) {
        const H2ScaleW a = h2w_scale(p.amax_g), b = h2w_scale(p.amax_x);
        sg = a.s; sg_inv = a.inv; sx = b.s; sx_inv = b.inv;
    }
    // TM 32-row blocks of output channels per wave: 128 co per workgroup, or 64 for the 64-channel layers (layer1), where half of
    // a 128-row tile would multiply zeros; p.mt counts tiles of this height, the splits are those of the 128-row form
    constexpr int BM = 2 * TM * 32, BNC = 64;
    constexpr int VA = BM / 4, RA = 256 / VA, PA = XHD_KS / RA;          // gy staging: VA threads per row, PA rows per thread
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * XHD_BUF];
    const int ntc = (p.Ci + BNC - 1) / BNC;
    const int tiles = p.mt * ntc * 3;
    int tile, split;
    {
        const int L = blockIdx.x;
        if (p.splits % 8 == 0) {
            const int xcd = L % 8, idx = L / 8;
            tile = idx % tiles;
            split = (idx / tiles) * 8 + xcd;
        } else {
            tile = L % tiles;
            split = L / tiles;
        }
    }
    const int tile_m = tile % p.mt, rest = tile / p.mt;
    const int tile_nc = rest % ntc, ky = rest / ntc;
    const int m0 = tile_m * BM, ci0 = tile_nc * BNC, dy = ky - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int Wv = p.W + 1;                                   // virtual row length
    const int64_t V = (int64_t)p.B * p.H * Wv;
    // the split's range of virtual pixels, on the slice boundaries of conv_wgrad_x3h_kernel (XH_KS pixels each)
    const int64_t v_begin = (int64_t)split * p.slices_per_split * XH_KS;
    const int64_t v_end = min((V + XH_KS - 1) / XH_KS * XH_KS, v_begin + p.slices_per_split * XH_KS);
    const int n_slices = v_begin < v_end ? (int)((v_end - v_begin) / XHD_KS) : 0;

    constexpr unsigned OOB = 0x80000000u;          // the host keeps both operands below 2^31 bytes
    // the gy descriptor ends at the split's last pixel (v_end less one padding column per image row before it): the slice staged
    // after the last one of the range is zeros, it must not reach the bias sums
    const int64_t px_end = min((int64_t)p.B * p.H * p.W, v_end - v_end / Wv);
    const __amdgpu_buffer_rsrc_t gy_desc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.gy), 0, (int)((unsigned)px_end * (unsigned)p.Co * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_desc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.x), 0, (int)((unsigned)p.B * p.H * p.W * p.Ci * 4u), 0x00020000);

    const int a_col = (tid % VA) * 4, a_row0 = tid / VA;      // gy rows a_row0 (+ RA)
    const int b_col = (tid & 15) * 4, b_row0 = tid >> 4;      // x run row b_row0; threads 0..31 also row 16 + b_row0
    const bool a_cok = m0 + a_col < p.Co, b_cok = ci0 + b_col < p.Ci;
    const bool ragged = tid < 32;
    // a staged row: column x in [0, Wv) of its image row, byte offset of the pixel it stands for (the padding column, x = W,
    // stands for the first pixel of the next row and is masked); the x run also carries its image row y
    auto decode = [&](int64_t v, int &x, int &y, int64_t &real) {
        const int64_t vv = v < 0 ? 0 : v;
        x = (int)(vv % Wv);
        const int64_t t = vv / Wv;                  // image rows before this one (b * H + y)
        y = (int)(t % p.H);
        real = vv - t;
        if (v < 0) { x = Wv - 1; y = -1; real = 0; }          // v = -1: the padding column of "row -1"
    };
    int ax[PA], bx[2], by[2];
    unsigned a_off[PA], b_off[2];
    const unsigned a_px = a_cok ? (unsigned)p.Co * 4u : 0u, b_px = b_cok ? (unsigned)p.Ci * 4u : 0u;       // bytes per pixel
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int y;
        int64_t real;
        decode(v_begin + a_row0 + RA * i, ax[i], y, real);
        a_off[i] = a_cok ? (unsigned)(real * p.Co + m0 + a_col) * 4u : OOB;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int64_t real;
        decode(v_begin - 1 + b_row0 + 16 * i, bx[i], by[i], real);
        b_off[i] = b_cok && (i == 0 || ragged) ? (unsigned)((real + (int64_t)dy * p.W) * p.Ci + ci0 + b_col) * 4u : OOB;
    }
    const unsigned b_px1 = ragged ? b_px : 0u;
    const int adv_rows = XHD_KS / Wv, adv_x = XHD_KS - adv_rows * Wv;      // narrow maps: a slice spans several image rows

    using u32x4w = __attribute__((ext_vector_type(4))) unsigned;
    auto load16 = [&](__amdgpu_buffer_rsrc_t d, unsigned off) {
        union { u32x4w u; float4 f; } c;
        c.u = __builtin_amdgcn_raw_buffer_load_b128(d, (int)off, 0, 0);
        return c.f;
    };
    float4 ra[2][PA], rb[2][2];
    auto load_slice = [&](float4 (&qa)[PA], float4 (&qb)[2]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            qa[i] = load16(gy_desc, ax[i] < p.W ? a_off[i] : OOB);
            ax[i] += adv_x;                                   // + 16 virtual pixels = adv_rows image rows + adv_x columns (+ a carry)
            const bool c1 = ax[i] >= Wv;
            ax[i] -= c1 ? Wv : 0;
            a_off[i] += XHD_KS * a_px - (unsigned)(adv_rows + (c1 ? 1 : 0)) * a_px;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned px = i == 0 ? b_px : b_px1;
            qb[i] = load16(x_desc, bx[i] < p.W && (unsigned)(by[i] + dy) < (unsigned)p.H ? b_off[i] : OOB);
            bx[i] += adv_x;
            const bool c1 = bx[i] >= Wv;
            bx[i] -= c1 ? Wv : 0;
            const int rows = adv_rows + (c1 ? 1 : 0);         // <= H (the host's condition for this kernel)
            by[i] += rows;
            by[i] -= by[i] >= p.H ? p.H : 0;
            b_off[i] += XHD_KS * px - (unsigned)rows * px;
        }
    };
    const bool do_bias = p.bias_out != nullptr && tile_nc == 0 && ky == 1;       // the centre row sees every gy row once
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto put = [&](unsigned short *d, int plane, float4 v, float sc) {
        if constexpr (H2) {
            unsigned h0, l0, h1, l1;
            split2w(v.x * sc, v.y * sc, h0, l0);
            split2w(v.z * sc, v.w * sc, h1, l1);
            *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(d + plane) = make_uint2(l0, l1);
        } else {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3x2w(v.x, v.y, h0, m0_, l0);
            split3x2w(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(d + plane) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2 *>(d + 2 * plane) = make_uint2(l0, l1);
        }
    };
    const int a_lds0 = a_row0 * X3_ROW + a_col, a_lds1 = (a_row0 + RA) * X3_ROW + a_col;
    const int b_lds0 = 3 * XHD_PLANE_A + b_row0 * XH_ROWB + b_col, b_lds1 = 3 * XHD_PLANE_A + (16 + b_row0) * XH_ROWB + b_col;

    f32x16 acc[3][TM];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;

    // one half of the loop body (see conv_wgrad_x3d_kernel): 36 MFMAs on the slice in `rd`, the split of the staged registers
    // into `wr` in six chunks of one split3x2w between them, the ragged rows of the x run last
    auto half = [&](const unsigned short *rd, unsigned short *wr, float4 (&qa)[PA], float4 (&qb)[2]) {
        constexpr int NPL = H2 ? 2 : 3;             // planes per operand
        bf16x8w fa[TM][3], fb[3][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NPL; ++q) fa[i][q] = tr_frag_row<X3_ROW>(rd + q * XHD_PLANE_A, 0, wm * TM * 32 + i * 32, lane);
#pragma unroll
        for (int t = 0; t < 3; ++t)            // tap kx = t reads the run one row further: gy row k meets x row k + t
#pragma unroll
            for (int q = 0; q < NPL; ++q) fb[t][q] = tr_frag_row<XH_ROWB>(rd + 3 * XHD_PLANE_A + q * XHD_PLANE_B, t, wn * 32, lane);
        __builtin_amdgcn_sched_barrier(0);
        // smallest terms first, as conv_wgrad_x3h_kernel; H2: a1 b0, a0 b1, a0 b0
        constexpr int QA[6] = {H2 ? 1 : 2, 0, H2 ? 0 : 1, 1, 0, 0}, QB[6] = {0, H2 ? 1 : 2, H2 ? 0 : 1, 0, 1, 0};
        unsigned h[2], m[2], l[2];
        constexpr int NC = 2 * (PA + 1), NM = (H2 ? 9 : 18) * TM;       // split chunks; MFMAs of the slice (products x 3 taps x TM blocks)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int u = c >> 1;                    // staged float4: the gy row(s), then the x row
            const float4 v = u < PA ? qa[u < PA ? u : 0] : qb[0];
            const float sc = u < PA ? sg : sx;
            if ((c & 1) == 0) {
                if constexpr (H2) split2w(v.x * sc, v.y * sc, h[0], l[0]);
                else split3x2w(v.x, v.y, h[0], m[0], l[0]);
            } else {
                if constexpr (H2) split2w(v.z * sc, v.w * sc, h[1], l[1]);
                else split3x2w(v.z, v.w, h[1], m[1], l[1]);
                unsigned short *d = wr + (u >= PA ? b_lds0 : u == 0 ? a_lds0 : a_lds1);
                const int plane = u < PA ? XHD_PLANE_A : XHD_PLANE_B;
                *reinterpret_cast<uint2 *>(d) = make_uint2(h[0], h[1]);
                if constexpr (H2) {
                    *reinterpret_cast<uint2 *>(d + plane) = make_uint2(l[0], l[1]);
                } else {
                    *reinterpret_cast<uint2 *>(d + plane) = make_uint2(m[0], m[1]);
                    *reinterpret_cast<uint2 *>(d + 2 * plane) = make_uint2(l[0], l[1]);
                }
                if (u < PA) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }    // every tile: no branch in the body
            }
#pragma unroll
            for (int e = c * NM / NC; e < (c + 1) * NM / NC; ++e) {          // MFMA e: product q of block (tap t, row block i)
                const int q = e / (3 * TM), t = (e % (3 * TM)) / TM, i = e % TM;
                acc[t][i] = mfma_x3w<H2>(fa[i][QA[q]], fb[t][QB[q]], acc[t][i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (__builtin_amdgcn_readfirstlane(wave) == 0) {      // a scalar branch: the other three waves skip the instructions
            if (ragged) put(wr + b_lds1, XHD_PLANE_B, qb[1], sx);
        }
        load_slice(qa, qb);
        __syncthreads();
    };
    auto store_slice = [&](unsigned short *wr, const float4 (&qa)[PA], const float4 (&qb)[2]) {
        put(wr + a_lds0, XHD_PLANE_A, qa[0], sg);
        if constexpr (PA > 1) put(wr + a_lds1, XHD_PLANE_A, qa[PA - 1], sg);
#pragma unroll
        for (int i = 0; i < PA; ++i) { bsum.x += qa[i].x; bsum.y += qa[i].y; bsum.z += qa[i].z; bsum.w += qa[i].w; }     // row by row
        put(wr + b_lds0, XHD_PLANE_B, qb[0], sx);
        if (ragged) put(wr + b_lds1, XHD_PLANE_B, qb[1], sx);
    };

    unsigned short *buf0 = lds, *buf1 = lds + XHD_BUF;
    load_slice(ra[0], rb[0]);
    load_slice(ra[1], rb[1]);
    store_slice(buf0, ra[0], rb[0]);
    load_slice(ra[0], rb[0]);
    __syncthreads();
    for (int s = 0; s < n_slices; s += 2) {          // n_slices is even: the split's range is whole XH_KS slices
        half(buf0, buf1, ra[1], rb[1]);
        half(buf1, buf0, ra[0], rb[0]);
    }

    if (do_bias) {          // threads sharing a column quad (tid % VA) fold their RA row partials in a fixed order
        float4 *red = reinterpret_cast<float4 *>(lds);
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < VA) {
            float4 t = red[tid];
            for (int r = 1; r < RA; ++r) {
                const float4 v = red[r * VA + tid];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            float *dst = p.bias_out + (int64_t)split * p.Co;
            const int m = m0 + tid * 4;
            if (m < p.Co) dst[m] = t.x;
            if (m + 1 < p.Co) dst[m + 1] = t.y;
            if (m + 2 < p.Co) dst[m + 2] = t.z;
            if (m + 3 < p.Co) dst[m + 3] = t.w;
        }
    }
    float *out = p.out + (int64_t)split * p.Co * p.Ntot;
    const int fcol = lane & 31, fhalf = lane >> 5;
    const int ci = ci0 + wn * 32 + fcol;
    if (ci < p.Ci) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int n = (ky * 3 + t) * p.Ci + ci;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    if (m < p.Co) out[(int64_t)m * p.Ntot + n] = H2 ? acc[t][i][r] * sg_inv * sx_inv : acc[t][i][r];
                }
        }
    }
}

// acc != 0: the sums are ADDED to what out / out2 hold (htd_conv2d_bwd_weight_acc: a weight shared by several layers -- the RPN
// convolutions over five pyramid levels -- collects its gradient in place, level after level on one stream: a fixed order)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, float *__restrict__ out,
                                                            int64_t n, int splits, const float *__restrict__ ws2 = nullptr,
                                                            float *__restrict__ out2 = nullptr, int n2 = 0, int acc = 0)
{
    // a second, small set of partials (bias gradient [splits][n2]) rides along in ceil(n2/256) extra workgroups
    // appended to the grid: one column per thread, four loads in flight
    const int extra = out2 ? (n2 + 255) / 256 : 0;
    if ((int)blockIdx.x >= (int)gridDim.x - extra) {
        const int i = ((int)blockIdx.x - ((int)gridDim.x - extra)) * 256 + threadIdx.x;
        if (i < n2) {
            float s = acc ? out2[i] : 0.f;
            int k = 0;
            for (; k + 3 < splits; k += 4) {
                const float a = ws2[(int64_t)k * n2 + i], b = ws2[(int64_t)(k + 1) * n2 + i];
                const float c = ws2[(int64_t)(k + 2) * n2 + i], d = ws2[(int64_t)(k + 3) * n2 + i];
                s += a; s += b; s += c; s += d;
            }
            for (; k < splits; ++k) s += ws2[(int64_t)k * n2 + i];
            out2[i] = s;
        }
        return;
    }
    const unsigned main_blocks = gridDim.x - extra;
    // out[i] = sum_k ws[k][i] in fixed order; float4 lanes, four partial slabs in flight per thread
    const int64_t n4 = (n & 3) == 0 ? (n >> 2) : 0;      // slabs are 16-byte aligned only when n % 4 == 0
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)main_blocks * blockDim.x) {
        float4 s = acc ? *reinterpret_cast<const float4 *>(out + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 3 < splits; k += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(ws + (int64_t)(k + u) * n + i * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < splits; ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(ws + (int64_t)k * n + i * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + i * 4) = s;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            float s = acc ? out[i] : 0.f;
            for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * n + i];
            out[i] = s;
        }
}

// out[c] = sum_k ws[k][c] for FEW columns and MANY partial rows (second stage of the bias gradient: n = C <= 2048,
// rows = up to 512 block partials).  splitk_reduce_kernel would walk the rows serially in one or two workgroups;
// here a workgroup takes 16 columns and spreads the rows over 16 thread groups, folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float *__restrict__ ws, float *__restrict__ out, int n,
                                                          int rows)
{
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + col;
    float s = 0.f;
    if (c < n)
        for (int k = grp; k < rows; k += 16) s += ws[(int64_t)k * n + c];
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && c < n) {
        float t = red[0][col];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += red[g][col];
        out[c] = t;
    }
}

// column sums of a [rows][C] matrix (bias gradient), optionally fused with the ReLU-mask of the incoming
// gradient:  gm = g * (y > 0) written back, gbias[c] = sum_rows gm.  Deterministic two-stage reduction.
// amax_out (may be NULL): max |gm| of what the launch stores (max |g| without a mask) is left in this device scalar -- zero or an
// earlier maximum on entry -- for an H2 consumer of the gradient (conv_x3.hip).  Magnitudes order like their bits; NaN above all.
using htd::mag_bits;
using htd::wave_mag_out;

__global__ __launch_bounds__(256) void colsum_mask_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                          float *__restrict__ gm, float *__restrict__ partial,
                                                          int64_t rows, int C, int64_t rows_per_block, float *__restrict__ amax_out)
{
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    unsigned mx = 0u;
    if (c < C)
        for (int64_t r = r0; r < r1; ++r) {
            float v = g[r * C + c];
            if (y) {
                v = y[r * C + c] > 0.f ? v : 0.f;
                gm[r * C + c] = v;
            }
            mx = max(mx, mag_bits(v));
            s += v;
        }
    if (amax_out) wave_mag_out(mx, amax_out);
    if (partial && c < C) partial[(int64_t)blockIdx.x * C + c] = s;
}

// float4 version for C % 4 == 0: a block's 256 threads cover R = 256 / (C/4) rows per iteration with 16-byte
// accesses (full-rate streaming), then fold their R partial rows through LDS.
__global__ __launch_bounds__(256) void colsum_mask_vec_kernel(const float *__restrict__ g, const float *__restrict__ y,
                                                              float *__restrict__ gm, float *__restrict__ partial,
                                                              int64_t rows, int C, int64_t rows_per_block,
                                                              float *__restrict__ amax_out)
{
    __shared__ float4 red[256];
    unsigned mx = 0u;
    const int C4 = C >> 2;
    const int cbase = blockIdx.y * 256;                         // float4 column chunk of this block
    const int cw = min(256, C4 - cbase);                        // float4 columns handled here
    const int R = 256 / cw;
    const int c4 = threadIdx.x % cw, rsub = threadIdx.x / cw;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rsub < R) {
        // four independent row streams per thread keep 8 x 16-byte loads in flight
        int64_t r = r0 + rsub;
        for (; r + 3 * (int64_t)R < r1; r += 4 * (int64_t)R) {
            float4 v[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t o = (r + u * (int64_t)R) * C + (int64_t)(cbase + c4) * 4;
                v[u] = *reinterpret_cast<const float4 *>(g + o);
                if (y) yv[u] = *reinterpret_cast<const float4 *>(y + o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (y) {
                    const int64_t o = (r + u * (int64_t)R) * C + (int64_t)(cbase + c4) * 4;
                    v[u].x = yv[u].x > 0.f ? v[u].x : 0.f; v[u].y = yv[u].y > 0.f ? v[u].y : 0.f;
                    v[u].z = yv[u].z > 0.f ? v[u].z : 0.f; v[u].w = yv[u].w > 0.f ? v[u].w : 0.f;
                    *reinterpret_cast<float4 *>(gm + o) = v[u];
                }
                mx = htd::mag_bits4(mx, v[u]);
                s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
            }
        }
        for (; r < r1; r += R) {
            const int64_t o = r * C + (int64_t)(cbase + c4) * 4;
            float4 v = *reinterpret_cast<const float4 *>(g + o);
            if (y) {
                const float4 yv = *reinterpret_cast<const float4 *>(y + o);
                v.x = yv.x > 0.f ? v.x : 0.f; v.y = yv.y > 0.f ? v.y : 0.f;
                v.z = yv.z > 0.f ? v.z : 0.f; v.w = yv.w > 0.f ? v.w : 0.f;
                *reinterpret_cast<float4 *>(gm + o) = v;
            }
            mx = htd::mag_bits4(mx, v);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (amax_out) wave_mag_out(mx, amax_out);
    if (!partial) return;                                       // mask-only call: no column sums wanted
    red[threadIdx.x] = s;
    __syncthreads();
    if (rsub == 0) {
        for (int k = 1; k < R; ++k) {
            const float4 t = red[k * cw + c4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        *reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.x * C + (int64_t)(cbase + c4) * 4) = s;
    }
}

struct Cfg { int splits; int mt, nt; int bm, bn; };

template <int WGM, int WGN, int TM, int TN>
void launch_wgrad(int pix, bool covec, dim3 grid, hipStream_t s, const WgradParams &p)
{
#define HTD_WGRAD_CASE(PIX, CV)                                                                                   \
    if (pix == PIX && covec == CV) {                                                                               \
        hipLaunchKernelGGL((conv_wgrad_kernel<WGM, WGN, TM, TN, PIX, CV>), grid, dim3(256), 0, s, p);             \
        return;                                                                                                    \
    }
    HTD_WGRAD_CASE(PIX_POINTWISE, true) HTD_WGRAD_CASE(PIX_WIDE, true) HTD_WGRAD_CASE(PIX_GENERAL, true)
    HTD_WGRAD_CASE(PIX_POINTWISE, false) HTD_WGRAD_CASE(PIX_WIDE, false) HTD_WGRAD_CASE(PIX_GENERAL, false)
#undef HTD_WGRAD_CASE
}

Cfg choose(int Co, int Ntot, int64_t K)
{
    Cfg c;
    c.bm = Co <= 32 ? 32 : 128;
    c.bn = 128;
    c.mt = (int)htd::ceil_div(Co, c.bm);
    c.nt = (int)htd::ceil_div(Ntot, c.bn);
    const int64_t slices = htd::ceil_div(K, BKW);
    // ~6 workgroups per CU.  Round 2 measured 768..6144 on the fp32-input kernels: flat from 2304; with the H2 kernels the reduce
    // pass weighs more and 768..1536 is 1.4 % less weight-gradient time per step than 2304 (tools/sweep_wgrad_units.sh)
    static const int units = getenv("HTD_WGRAD_UNITS") ? atoi(getenv("HTD_WGRAD_UNITS")) : 1536;
    int64_t want = htd::ceil_div(units, (int64_t)c.mt * c.nt);
    const int64_t tiles = (int64_t)c.mt * c.nt;
    int64_t cap = std::max<int64_t>(1, slices / 20);                 // at least 20 slices (640 pixels) per split ...
    if (tiles * cap < 256)                                           // ... unless that leaves CUs idle (short reductions)
        cap = std::max(cap, std::min<int64_t>(htd::ceil_div(256, tiles), std::max<int64_t>(1, slices / 5)));
    want = std::min<int64_t>(want, cap);
    c.splits = (int)std::max<int64_t>(1, std::min<int64_t>(want, 192));
    if (c.splits >= 6) c.splits = (c.splits + 7) / 8 * 8;            // multiples of 8: one split per XCD group
    static const int forced = getenv("HTD_WGRAD_SPLITS") ? atoi(getenv("HTD_WGRAD_SPLITS")) : 0;       // tuning runs only
    if (forced > 0) c.splits = (int)std::min<int64_t>(forced, slices);
    return c;
}

// plan of the tap-fused 3x3 kernel: tiles = (Co / 128) x (Ci / 64) x 3 filter rows, K = virtual pixels B * H * (W + 1)
static const bool g_wgrad_x3h_off = getenv("HTD_WGRAD_X3H") != nullptr && atoi(getenv("HTD_WGRAD_X3H")) == 0;
bool x3h_takes(int Ci, int Co, int kh, int kw, int stride, int pad, int dil)
{
    return !g_wgrad_x3h_off && kh == 3 && kw == 3 && stride == 1 && pad == 1 && dil == 1 && Co >= 64 && (Co & 3) == 0 &&
           Ci >= 32 && (Ci & 3) == 0;
}
int x3h_splits(int B, int H, int W, int Ci, int Co)
{
    const int64_t tiles = htd::ceil_div(Co, 128) * htd::ceil_div(Ci, 64) * 3;
    const int64_t slices = htd::ceil_div((int64_t)B * H * (W + 1), XH_KS);
    // 768 workgroups (1.5 x the 512 resident ones), at least 8 slices each, at most 128 splits: measured over the splits of every
    // 3x3 shape of the step with conv_wgrad_x3hd_kernel (profiles/r03_wgrad_x3h_splits_sweep.txt) -- beyond that the partial
    // tiles (a workgroup writes 98 KB, the reduce kernel reads them back) cost more than the fuller chip gains
    static const int target = getenv("HTD_WGRAD_X3H_UNITS") ? atoi(getenv("HTD_WGRAD_X3H_UNITS")) : 768;
    int64_t want = htd::ceil_div(target, tiles);
    int64_t cap = std::max<int64_t>(1, slices / 8);
    if (tiles * cap < 256) cap = std::max(cap, std::min<int64_t>(htd::ceil_div(256, tiles), std::max<int64_t>(1, slices / 4)));
    want = std::min<int64_t>(want, cap);
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>(want, 128));
    if (splits >= 6) splits = (splits + 7) / 8 * 8;
    static const int forced = getenv("HTD_WGRAD_X3H_SPLITS") ? atoi(getenv("HTD_WGRAD_X3H_SPLITS")) : 0;      // tuning runs only
    if (forced > 0) splits = (int)std::min<int64_t>(forced, std::max<int64_t>(1, slices / 2));
    return splits;
}

}  // namespace

int g_wgrad_math = -1;       // -1: HTD_CONV_MATH / default; set together with the forward mode by htd_conv2d_set_math

extern "C" int64_t htd_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Co, int kh, int kw, int stride,
                                                    int pad, int dil)
{
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    const Cfg c = choose(Co, kh * kw * Ci, (int64_t)B * Ho * Wo);
    int splits = c.splits;
    if (x3h_takes(Ci, Co, kh, kw, stride, pad, dil)) splits = std::max(splits, x3h_splits(B, H, W, Ci, Co));
    return (int64_t)splits * Co * (kh * kw * Ci + 1) * 4 + 256;        // weight partials + bias partials
}

// gbias (may be NULL): also returns the bias gradient, column sums of gy, accumulated by the same kernel.
// accumulate: gw (and gbias) += the gradient; the kernels then always write partials and the reduce pass adds them in.
static int bwd_weight_impl(const float *x, const float *gy, float *gw, float *gbias, int B, int H, int W, int Ci, int Co, int kh,
                           int kw, int stride, int pad, int dil, void *workspace, void *stream, int accumulate,
                           const float *amax_x = nullptr, const float *amax_g = nullptr)
{
    HTD_REQUIRE(B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0 && pad >= 0,
                "conv2d_bwd_weight: bad sizes");
    HTD_REQUIRE(Ci % 4 == 0, "conv2d_bwd_weight: Ci=%d must be a multiple of 4", Ci);
    HTD_REQUIRE(x && gy && gw && workspace, "conv2d_bwd_weight: null pointer");
    WgradParams p{};
    p.x = x; p.gy = gy;
    p.amax_x = amax_x; p.amax_g = amax_g;
    const bool h2 = amax_x != nullptr && amax_g != nullptr;
    p.B = B; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.dil = dil;
    p.Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    p.Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    HTD_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d_bwd_weight: empty output");
    p.K = (int64_t)B * p.Ho * p.Wo;
    HTD_REQUIRE((int64_t)B * H * W * Ci < (1ll << 31) && p.K * Co < (1ll << 31),
                "conv2d_bwd_weight: operand larger than 2^31 elements (32-bit element offsets)");
    p.Ntot = kh * kw * Ci;
    static const int math_env0 = getenv("HTD_CONV_MATH") ? atoi(getenv("HTD_CONV_MATH")) : 1;
    if ((g_wgrad_math < 0 ? math_env0 : g_wgrad_math) == 1 && x3h_takes(Ci, Co, kh, kw, stride, pad, dil)) {
        // 3x3, stride 1: the three taps of a filter row per workgroup (conv_wgrad_x3h_kernel)
        p.mt = (int)htd::ceil_div(Co, 128);
        p.nt = (int)htd::ceil_div(Ci, 64);
        p.splits = x3h_splits(B, H, W, Ci, Co);
        const int64_t slices = htd::ceil_div((int64_t)B * H * (W + 1), XH_KS);
        p.slices_per_split = htd::ceil_div(slices, p.splits);
        p.splits = (int)htd::ceil_div(slices, p.slices_per_split);
        const bool direct = p.splits == 1 && !accumulate;
        p.out = direct ? gw : (float *)workspace;
        float *bias_part = (float *)workspace + (int64_t)p.splits * Co * p.Ntot;
        p.bias_out = !gbias ? nullptr : (direct ? gbias : bias_part);
        hipStream_t s = (hipStream_t)stream;
        const dim3 grid((unsigned)(p.mt * p.nt * 3 * p.splits));
        static const bool x3hd_on = !(getenv("HTD_WGRAD_X3D") && atoi(getenv("HTD_WGRAD_X3D")) == 0);
        // conv_wgrad_x3hd_kernel addresses its operands with 32-bit BYTE offsets into buffer descriptors
        // (a slice of 16 virtual pixels must not span more than H image rows: the row index wraps once per slice at most)
        if (x3hd_on && XHD_KS / (W + 1) + 1 <= H && (int64_t)B * H * W * Ci * 4 < (1ll << 31) &&
            (int64_t)B * H * W * Co * 4 < (1ll << 31))
        {
            static const bool tile64 = !(getenv("HTD_WGRAD_TILE64") && atoi(getenv("HTD_WGRAD_TILE64")) == 0);
            if (tile64 && Co <= 64) {                 // 64 output channels: 64-row tiles (same tile count, no rows of zeros)
                p.mt = (int)htd::ceil_div(Co, 64);
                const dim3 g64((unsigned)(p.mt * p.nt * 3 * p.splits));
                if (h2) hipLaunchKernelGGL((conv_wgrad_x3hd_kernel<1, true>), g64, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((conv_wgrad_x3hd_kernel<1, false>), g64, dim3(256), 0, s, p);
            } else if (h2) {
                hipLaunchKernelGGL((conv_wgrad_x3hd_kernel<2, true>), grid, dim3(256), 0, s, p);
            } else {
                hipLaunchKernelGGL((conv_wgrad_x3hd_kernel<2, false>), grid, dim3(256), 0, s, p);
            }
        }
        else {
            HTD_REQUIRE(!h2, "conv2d_bwd_weight_h2: this 3x3 layer is not taken by the H2 kernels (htd_conv2d_bwd_weight_h2_supported)");
            if (W + 1 >= XH_KS) hipLaunchKernelGGL(conv_wgrad_x3h_kernel<true>, grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL(conv_wgrad_x3h_kernel<false>, grid, dim3(256), 0, s, p);
        }
        if (!direct) {
            const int64_t n = (int64_t)Co * p.Ntot;
            const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(n, 256), 2048);
            const unsigned extra = gbias ? (unsigned)htd::ceil_div(Co, 256) : 0u;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks + extra), dim3(256), 0, s, (const float *)workspace, gw, n,
                               p.splits, (const float *)(gbias ? bias_part : nullptr), gbias, Co, accumulate);
        }
        return htd::check_launch("conv2d_bwd_weight");
    }
    const Cfg c = choose(Co, p.Ntot, p.K);
    p.mt = c.mt; p.nt = c.nt; p.splits = c.splits;
    p.slices_per_split = htd::ceil_div(htd::ceil_div(p.K, BKW), c.splits);
    const bool direct = c.splits == 1 && !accumulate;
    p.out = direct ? gw : (float *)workspace;
    float *bias_partial = (float *)workspace + (int64_t)c.splits * Co * p.Ntot;
    p.bias_out = !gbias ? nullptr : (direct ? gbias : bias_partial);
    hipStream_t s = (hipStream_t)stream;
    p.slices_per_split = htd::ceil_div(htd::ceil_div(p.K, BKW), c.splits);
    dim3 grid((unsigned)(c.mt * c.nt * c.splits));
    const int pix = (kh == 1 && kw == 1 && stride == 1 && pad == 0) ? PIX_POINTWISE : (p.Wo >= BKW ? PIX_WIDE : PIX_GENERAL);
    const bool covec = (Co & 3) == 0;
    static const int math_env = getenv("HTD_CONV_MATH") ? atoi(getenv("HTD_CONV_MATH")) : 1;
    const bool x3 = (g_wgrad_math < 0 ? math_env : g_wgrad_math) == 1 && c.bm == 128 && covec && (Ci & 3) == 0;
    if (x3) {
        static const bool x3d_on = !(getenv("HTD_WGRAD_X3D") && atoi(getenv("HTD_WGRAD_X3D")) == 0);
        // conv_wgrad_x3d_kernel addresses its operands with 32-bit BYTE offsets into buffer descriptors
        const bool x3d = x3d_on && (int64_t)B * H * W * Ci * 4 < (1ll << 31) && p.K * Co * 4 < (1ll << 31);
        if (x3d) {
            // 64-wide tiles where a 128-wide one would be half zeros (the split count stays that of the 128x128 form)
            static const bool tile64 = !(getenv("HTD_WGRAD_TILE64") && atoi(getenv("HTD_WGRAD_TILE64")) == 0);
            const bool m64 = tile64 && Co <= 64 && pix != PIX_GENERAL, n64 = tile64 && !m64 && p.Ntot <= 64 && pix == PIX_POINTWISE;
            p.mt = (int)htd::ceil_div(Co, m64 ? 64 : 128);
            p.nt = (int)htd::ceil_div(p.Ntot, n64 ? 64 : 128);
            // splits % 8 != 0: (split, N tile) groups dealt over the 8 XCDs, mt workgroups each (see the kernel)
            const dim3 gd = c.splits % 8 == 0 ? dim3((unsigned)(p.mt * p.nt * c.splits))
                                              : dim3((unsigned)(8 * htd::ceil_div(p.nt * c.splits, 8) * p.mt));
            if (h2) {
                if (m64 && pix == PIX_POINTWISE) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_POINTWISE, 1, 2, true>), gd, dim3(256), 0, s, p);
                else if (m64) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_WIDE, 1, 2, true>), gd, dim3(256), 0, s, p);
                else if (n64) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_POINTWISE, 2, 1, true>), gd, dim3(256), 0, s, p);
                else if (pix == PIX_POINTWISE) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_POINTWISE, 2, 2, true>), gd, dim3(256), 0, s, p);
                else if (pix == PIX_WIDE) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_WIDE, 2, 2, true>), gd, dim3(256), 0, s, p);
                else hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_GENERAL, 2, 2, true>), gd, dim3(256), 0, s, p);
            }
            else if (m64 && pix == PIX_POINTWISE) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_POINTWISE, 1, 2>), gd, dim3(256), 0, s, p);
            else if (m64) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_WIDE, 1, 2>), gd, dim3(256), 0, s, p);
            else if (n64) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_POINTWISE, 2, 1>), gd, dim3(256), 0, s, p);
            else if (pix == PIX_POINTWISE) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_POINTWISE, 2, 2>), gd, dim3(256), 0, s, p);
            else if (pix == PIX_WIDE) hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_WIDE, 2, 2>), gd, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((conv_wgrad_x3d_kernel<PIX_GENERAL, 2, 2>), gd, dim3(256), 0, s, p);
        } else if (h2) {
            HTD_REQUIRE(false, "conv2d_bwd_weight_h2: operands too large for the H2 kernels (htd_conv2d_bwd_weight_h2_supported)");
        } else if (pix == PIX_POINTWISE) hipLaunchKernelGGL(conv_wgrad_x3_kernel<PIX_POINTWISE>, grid, dim3(256), 0, s, p);
        else if (pix == PIX_WIDE) hipLaunchKernelGGL(conv_wgrad_x3_kernel<PIX_WIDE>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_wgrad_x3_kernel<PIX_GENERAL>, grid, dim3(256), 0, s, p);
    } else if (h2) {
        HTD_REQUIRE(false, "conv2d_bwd_weight_h2: layer not taken by the H2 kernels (htd_conv2d_bwd_weight_h2_supported)");
    } else if (c.bm == 32)
        launch_wgrad<1, 4, 1, 1>(pix, covec, grid, s, p);
    else
        launch_wgrad<2, 2, 2, 2>(pix, covec, grid, s, p);
    if (!direct) {
        const int64_t n = (int64_t)Co * p.Ntot;
        const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(n, 256), 2048);
        const unsigned extra = gbias ? (unsigned)htd::ceil_div(Co, 256) : 0u;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks + extra), dim3(256), 0, s, (const float *)workspace, gw, n,
                           c.splits, (const float *)(gbias ? bias_partial : nullptr), gbias, Co, accumulate);
    }
    return htd::check_launch("conv2d_bwd_weight");
}

extern "C" int htd_conv2d_bwd_weight(const float *x, const float *gy, float *gw, float *gbias, int B, int H, int W,
                                     int Ci, int Co, int kh, int kw, int stride, int pad, int dil, void *workspace,
                                     void *stream)
{
    return bwd_weight_impl(x, gy, gw, gbias, B, H, W, Ci, Co, kh, kw, stride, pad, dil, workspace, stream, 0);
}

// gw += the weight gradient, gbias += the bias gradient: for a parameter that several layers share (RPNHead's convolutions run
// on five pyramid levels, anchor_head.py:123-140; the stage-1 classifier that stage 2 reuses, htd_bbox_head.py:158): the
// gradient collects in place, call after call on one stream, instead of five tensors that autograd then adds.
extern "C" int htd_conv2d_bwd_weight_acc(const float *x, const float *gy, float *gw, float *gbias, int B, int H, int W,
                                         int Ci, int Co, int kh, int kw, int stride, int pad, int dil, void *workspace,
                                         void *stream)
{
    return bwd_weight_impl(x, gy, gw, gbias, B, H, W, Ci, Co, kh, kw, stride, pad, dil, workspace, stream, 1);
}

// The weight gradient on the H2 arithmetic (conv_x3.hip: two fp16 pieces per operand, three products): amax_x / amax_g are device
// scalars holding max |x| / max |gy| of the two tensors (htd_absmax, or what the epilogue that wrote them left behind).
// htd_conv2d_bwd_weight_h2_supported: 1 when the layer runs on the kernels that have the H2 form (the tap-fused 3x3 kernel and
// the 128-wide split-in-the-shadow kernel); accumulate != 0: htd_conv2d_bwd_weight_acc's semantics.
extern "C" int htd_conv2d_bwd_weight_h2_supported(int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil)
{
    static const int math_env = getenv("HTD_CONV_MATH") ? atoi(getenv("HTD_CONV_MATH")) : 1;
    if (htd_conv2d_set_h2(-1) != 1 || (g_wgrad_math < 0 ? math_env : g_wgrad_math) != 1) return 0;
    if (B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || (Ci & 3) != 0 || stride <= 0 || dil <= 0 || pad < 0) return 0;
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1, Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const int64_t K = (int64_t)B * Ho * Wo;
    static const bool d_on = !(getenv("HTD_WGRAD_X3D") && atoi(getenv("HTD_WGRAD_X3D")) == 0);
    if (!d_on || (int64_t)B * H * W * Ci * 4 >= (1ll << 31) || K * Co * 4 >= (1ll << 31)) return 0;
    if (x3h_takes(Ci, Co, kh, kw, stride, pad, dil)) return XHD_KS / (W + 1) + 1 <= H ? 1 : 0;
    const Cfg c = choose(Co, kh * kw * Ci, K);
    return (c.bm == 128 && (Co & 3) == 0) ? 1 : 0;
}

extern "C" int htd_conv2d_bwd_weight_h2(const float *x, const float *gy, const float *amax_x, const float *amax_g, float *gw,
                                        float *gbias, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                                        int dil, int accumulate, void *workspace, void *stream)
{
    HTD_REQUIRE(amax_x && amax_g, "conv2d_bwd_weight_h2: null maximum");
    HTD_REQUIRE(htd_conv2d_bwd_weight_h2_supported(B, H, W, Ci, Co, kh, kw, stride, pad, dil) == 1,
                "conv2d_bwd_weight_h2: layer not taken by the H2 kernels");
    return bwd_weight_impl(x, gy, gw, gbias, B, H, W, Ci, Co, kh, kw, stride, pad, dil, workspace, stream, accumulate ? 1 : 0, amax_x,
                           amax_g);
}

// g [rows][C], y (may be NULL) [rows][C]; gm (out, required iff y) ; gbias [C]; workspace >= 2048*C*4 bytes
static int bias_grad_relu_mask_impl(const float *g, const float *y, float *gm, float *gbias, int64_t rows, int C, void *workspace,
                                    float *amax_out, void *stream)
{
    HTD_REQUIRE(rows >= 0 && C > 0, "bias_grad: bad sizes");
    HTD_REQUIRE(g && (gbias || y) && (!gbias || workspace) && (!y || gm), "bias_grad: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // 8 rows per block at least: the FC stacks' 2048 x 1024 gradients used to run on 32 workgroups (23 us for 25 MB)
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(1024, htd::ceil_div(rows, 8)));
    const int64_t rpb = htd::ceil_div(std::max<int64_t>(rows, 1), nb);
    float *partial = gbias ? (float *)workspace : nullptr;          // gbias == NULL: ReLU mask only
    if ((C & 3) == 0 && (((uintptr_t)g | (uintptr_t)y | (uintptr_t)gm | (uintptr_t)partial) & 15) == 0)
        hipLaunchKernelGGL(colsum_mask_vec_kernel, dim3(nb, (unsigned)htd::ceil_div(C / 4, 256)), dim3(256), 0, s, g, y,
                           gm, partial, rows, C, rpb, amax_out);
    else
        hipLaunchKernelGGL(colsum_mask_kernel, dim3(nb, (unsigned)htd::ceil_div(C, 256)), dim3(256), 0, s, g, y, gm,
                           partial, rows, C, rpb, amax_out);
    if (gbias)
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)htd::ceil_div(C, 16)), dim3(256), 0, s,
                           (const float *)partial, gbias, C, nb);
    return htd::check_launch("bias_grad");
}

extern "C" int htd_bias_grad_relu_mask(const float *g, const float *y, float *gm, float *gbias, int64_t rows, int C,
                                       void *workspace, void *stream)
{
    return bias_grad_relu_mask_impl(g, y, gm, gbias, rows, C, workspace, nullptr, stream);
}

// the same, and the largest magnitude of gm (of g when y is NULL) is left in *amax_out (zero or an earlier maximum on entry): the
// `amax` of the data- and weight-gradient launches that read the masked gradient on the H2 arithmetic
extern "C" int htd_bias_grad_relu_mask_amax(const float *g, const float *y, float *gm, float *gbias, int64_t rows, int C,
                                            void *workspace, float *amax_out, void *stream)
{
    HTD_REQUIRE(amax_out, "bias_grad: null maximum");
    return bias_grad_relu_mask_impl(g, y, gm, gbias, rows, C, workspace, amax_out, stream);
}
