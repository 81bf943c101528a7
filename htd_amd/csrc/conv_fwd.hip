// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   Y[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] + R[m][n] )
//   m = (b, ho, wo) output pixel, n = output channel, k = (kh, kw, ci)
//
// NHWC activations make every A row a contiguous run of Ci floats per filter tap, and KRSC weights make
// every B row contiguous in the same k order, so both operands are staged global -> registers -> LDS as
// 16-byte vectors with no im2col buffer (im2col happens in the address arithmetic; padding = zero fill).
//
// Tiling (64-wide wavefronts): block = 256 threads = 4 waves.  128x128 tile: 2x2 waves, each wave a 64x64
// accumulator tile = 2x2 MFMA 32x32 blocks (64 accumulator VGPRs); 128x64 / 128x32: 4x1 waves; 64x64: 2x2 waves of
// 32x32.  K is walked in BK-float slices through ONE LDS slice buffer plus a register prefetch: the next slice's
// global loads are issued before the MFMAs of the current slice and written to LDS between two barriers (half the LDS
// of a double buffer => three resident workgroups per CU; a double-buffered variant measured the same 119 TF/s).
// LDS rows are padded by 16 B so the ds_read_b128 fragment reads (lane = row, two K-halves per wave) are
// bank-conflict free for BK in {8,16,32}.  The MFMA k index is permuted (lane half h takes floats 4h..4h+3 of an
// 8-float group) -- legal because A and B use the same permutation -- which lets one ds_read_b128 per operand block
// feed four MFMAs.  __launch_bounds__(256, 3) keeps the accumulators in architectural VGPRs (no AGPR moves).
//
// The same kernel serves: forward conv, Linear layers (1x1 on "pixels" = rows), and the data gradient
// (forward conv of gy with spatially flipped, channel-transposed weights; a stride-s data gradient is split into
// s*s dense sub-problems, one per output parity class, through an explicit tap table).
#include <algorithm>
#include <mutex>
#include <unordered_map>

#include <stdlib.h>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// ---- fp32 on the bf16 matrix pipe (X3 kernels) -------------------------------------------------------------------
// An fp32 value has a 24-bit significand; three bf16 numbers (8 significant bits each, the exponent range of fp32)
// carry it:  a = a0 + a1 + a2 (+ <= 2^-27 |a|),  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1), the two
// differences being exact in fp32.  A product is then  a*b = sum_{i,j} a_i*b_j;  every a_i*b_j is exact in fp32 (8 x 8
// bits) and the six terms with i + j <= 2 carry everything down to ~2^-24 |a*b| (the three dropped ones, a1*b2, a2*b1,
// a2*b2, are <= 2^-25 |a*b| together).  v_mfma_f32_32x32x16_bf16 accumulates them in fp32 exactly like the fp32-input MFMA
// accumulates its products, at 16x the per-instruction rate: six of them per 16 k cost 192 cycles against the 512 of
// eight v_mfma_f32_32x32x2_f32 -- fp32 results (same rounding error class, tests/test_gpu_conv.py compares both forms with
// an fp64 reference) at 2.67x the matrix throughput.  The split is done once per element while a K slice is written to
// LDS (three 16-bit planes per operand row); the MFMA loop reads three fragments per operand block.
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// two floats -> three packed bf16 pairs (element 0 in the low half): one v_cvt_pk_bf16_f32 (round to nearest even) per
// piece; the residuals a - a0 and a - a0 - a1 are exact in fp32, the last piece leaves <= 2^-27 |a|.  (An infinity turns
// into a NaN on the way -- Inf - Inf -- which is what the sum of products it feeds would mostly become anyway.)
__device__ __forceinline__ void split3x2(float a, float b, unsigned &h, unsigned &m, unsigned &l)
{
    union { bf16x2 v; unsigned u; } c;
    c.v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    h = c.u;
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    c.v = __builtin_convertvector(f32x2{ra, rb}, bf16x2);
    m = c.u;
    c.v = __builtin_convertvector(f32x2{ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u)}, bf16x2);
    l = c.u;
}

// value select (a ternary between two float4 lvalues would select between ADDRESSES and push both to scratch)
__device__ __forceinline__ float4 keep4(bool ok, float4 v)
{
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// One split-bf16 product step with the accumulator pinned to the ACC half of the register file ("a" operands): the
// matrix core reads and writes C through the AGPR ports and leaves the VGPR ports to the operand split and the address
// arithmetic of the other waves (MI355X: +2 % on the large 3x3 layers; the compiler's own choice at this register budget
// is the VGPR form).
__device__ __forceinline__ void mfma_x3(f32x16 &acc, const bf16x8 &a, const bf16x8 &b)
{
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

struct ConvParams {
    const float *x, *w, *bias, *residual, *mask_src;
    float *y;
    int B, H, W, Ci, Co, kh, kw, stride, pad, dil, Ho, Wo;
    int Hx, Wx;         // physical size of x
    // explicit tap table (strided data gradients are split into stride*stride dense sub-problems, one per
    // output parity class): tap t reads input pixel (i + tap_dy[t], j + tap_dx[t]) with weight tap tap_w[t],
    // and sub-grid pixel (i, j) is written to output pixel (o_h0 + o_step*i, o_w0 + o_step*j) of an oH x oW map
    int ntaps;
    int tap_dy[16], tap_dx[16], tap_w[16];
    int o_h0, o_w0, o_step, oH, oW;
    int relu;
    // res_H > 0: `residual` is a coarser [B][res_H][res_W][Co] map read through nearest-neighbour up-sampling
    // (the FPN top-down path): source row = min(floor(ho * res_sh), res_H - 1), ATen's nearest rule
    int res_H, res_W;
    float res_sh, res_sw;
    int64_t w_bstride;  // >0: batched GEMM, image b uses weights w + b*w_bstride (tiles never straddle images)
    // batched GEMM over zero-padded groups (PGraph): group_count[b] rows / columns / reduction entries of group b are real,
    // the rest is padding that is known to be zero.  lim bits: 1 = rows (M), 2 = columns (N), 4 = reduction (K).  Tiles
    // beyond the count store zeros without computing, the K loop stops at the count -- no host read of the group sizes.
    const int64_t *group_count;
    int lim;
    int64_t M;          // B*Ho*Wo
    int mt, nt;         // tiles along M, N
    int splits;         // split-K: gridDim.y workgroups share one output tile, partials go to `partial`
    int slices_per_split;
    float *partial;     // [splits][M][Co] raw accumulators when splits > 1
    // balanced tail (splits == 1 only): tiles [0, tail_first) are computed whole; the last few tiles -- the ones that
    // would leave some CUs one tile more than the others -- are cut into tail_splits K ranges each, so that the
    // remainder is spread over the whole chip.  Their partial tiles go to partial[(tile - tail_first) * tail_splits +
    // split][BM][BN]; conv_tail_epilogue_kernel sums them and applies the epilogue.
    int tail_first, tail_splits, tail_sps;
    // K order of the slices: 0 = tap-major (all channels of a tap, then the next tap), 1 = channel-major (the kh*kw taps of
    // one BK-channel slice back to back: a tile re-reads its own footprint from L1/L2 instead of streaming the whole
    // input once per tap through an L2 that the concurrent tiles of the XCD overflow)
    int cmajor;
    // amax_out != NULL: the epilogues also leave the largest magnitude they store in this device scalar (zero or an earlier maximum
    // on entry) -- the `amax` of an H2 consumer of y (conv_x3.hip), without a pass over y
    float *amax_out;
};

// BK: floats of K per slice; WGM x WGN: wave grid of the block; TM x TN: 32x32 MFMA blocks per wave
template <int BK, int WGM, int WGN, int TM, int TN, bool X3 = false>
struct Tile {
    static constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    static constexpr int LDS_STRIDE = BK + 4;                 // floats
    // X3: a row holds three bf16 planes of BK elements + 16 B pad: (3*BK + 8) half-words; 52 dwords at BK = 32, which
    // puts the 16 rows of a ds_read_b128 group on 16 different 4-bank sets
    static constexpr int ROW_HALFS = 3 * BK + 8;
    static constexpr int VEC_PER_ROW = BK / 4;                // float4 per row slice
    static constexpr int ROWS_PER_PASS = 256 / VEC_PER_ROW;   // rows covered by the 256 threads at once
    static constexpr int PASSES_A = (BM + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
    static constexpr int PASSES_B = (BN + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
    static constexpr int MAIN_FLOATS = X3 ? (BM + BN) * ROW_HALFS / 2 : (BM + BN) * LDS_STRIDE;   // ONE slice buffer + register prefetch
    static constexpr int EPI_STRIDE = BN + 4;
    static constexpr int EPI_ROWS = WGM * 32;                 // rows staged per epilogue round
    static constexpr int EPI_FLOATS = EPI_ROWS * EPI_STRIDE;
    static constexpr int LDS_FLOATS = MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS;
};

template <int BK, int WGM, int WGN, int TM, int TN, bool TAPS, bool X3 = false>
__global__ __launch_bounds__(256, (TM * TN >= 4 ? 3 : 4)) void conv_igemm_kernel(ConvParams p)
{
    static_assert(!X3 || BK % 16 == 0, "the bf16 MFMA takes 16 k per instruction");
    using T = Tile<BK, WGM, WGN, TM, TN, X3>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];

    // XCD-aware tile order: blocks that share an XCD (ids congruent mod 8) get one contiguous run of tiles with
    // the N tiles of an M tile adjacent: the A rows of an M tile are fetched into that XCD's L2 once for all its
    // N tiles, neighbouring M tiles share their halo rows, and the (small) weight panel stays L2-resident.
    const int nblk = p.mt * p.nt;
    int bid = blockIdx.x, tail_split = -1;
    if (p.tail_splits > 0) {
        // per XCD (workgroups congruent mod 8, dispatched in increasing id): first its run of whole tiles, then its share
        // of the tail units; tail_first is a multiple of 8
        const int xcd = bid % 8, idx = bid / 8, dp = p.tail_first / 8;
        if (idx < dp) {
            bid = xcd * dp + idx;
        } else {
            const int units = (nblk - p.tail_first) * p.tail_splits, q = units / 8, r = units % 8;
            const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (idx - dp);
            bid = p.tail_first + w / p.tail_splits;
            tail_split = w % p.tail_splits;
        }
    } else {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.nt, tile_n = bid % p.nt;
    const int64_t m0 = (int64_t)tile_m * T::BM;
    const int n0 = tile_n * T::BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    int k_limit = 0x7fffffff;                        // real reduction length of the tile's group (padded groups only)
    if (p.group_count) {
        const int rows_per_group = p.Ho * p.Wo;
        const int g = (int)(m0 / rows_per_group);
        const int cnt = (int)p.group_count[g];
        const bool dead = ((p.lim & 1) && (int)(m0 - (int64_t)g * rows_per_group) >= cnt) || ((p.lim & 2) && n0 >= cnt);
        if (dead) {                                  // padding only: the tile is zeros (wave-uniform exit before any barrier)
            for (int e = tid; e < T::BM * (T::BN / 4); e += 256) {
                const int64_t m = m0 + e / (T::BN / 4);
                const int n = n0 + (e % (T::BN / 4)) * 4;
                if (m < p.M && n + 3 < p.Co) *reinterpret_cast<float4 *>(p.y + m * p.Co + n) = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (m < p.M)
                    for (int q = 0; q < 4; ++q)
                        if (n + q < p.Co) p.y[m * p.Co + n + q] = 0.f;
            }
            return;
        }
        if (p.lim & 4) k_limit = cnt;
    }

    // ---- per-thread staging coordinates
    const int vcol = tid % T::VEC_PER_ROW;            // which float4 of the K slice
    const int vrow = tid / T::VEC_PER_ROW;            // first row handled
    int a_hi0[T::PASSES_A], a_wi0[T::PASSES_A];
    unsigned a_img[T::PASSES_A];                       // element offsets fit 32 bits (checked on the host)
    bool a_ok[T::PASSES_A];
#pragma unroll
    for (int i = 0; i < T::PASSES_A; ++i) {
        const int r = vrow + i * T::ROWS_PER_PASS;
        const int64_t m = m0 + r;
        a_ok[i] = r < T::BM && m < p.M;
        const unsigned mm = a_ok[i] ? (unsigned)m : 0u;
        const unsigned wo = mm % (unsigned)p.Wo;
        const unsigned t = mm / (unsigned)p.Wo;
        const unsigned ho = t % (unsigned)p.Ho;
        const unsigned b = t / (unsigned)p.Ho;
        a_hi0[i] = TAPS ? (int)ho : (int)ho * p.stride - p.pad;
        a_wi0[i] = TAPS ? (int)wo : (int)wo * p.stride - p.pad;
        a_img[i] = b * (unsigned)(p.Hx * p.Wx);
    }
    const unsigned wrow_stride = (unsigned)(p.kh * p.kw * p.Ci);
    const float *wbase = p.w + (p.w_bstride > 0 ? (m0 / ((int64_t)p.Ho * p.Wo)) * p.w_bstride : 0);
    bool b_ok[T::PASSES_B];
    unsigned b_off[T::PASSES_B];
#pragma unroll
    for (int i = 0; i < T::PASSES_B; ++i) {
        const int r = vrow + i * T::ROWS_PER_PASS;
        const int n = n0 + r;
        b_ok[i] = r < T::BN && n < p.Co;
        b_off[i] = (unsigned)(b_ok[i] ? n : 0) * wrow_stride + vcol * 4;
    }

    const int slices_per_tap = p.Ci / BK;
    const int total_slices = (TAPS ? p.ntaps : p.kh * p.kw) * slices_per_tap;
    const int s_begin = tail_split >= 0 ? tail_split * p.tail_sps : blockIdx.y * p.slices_per_split;
    int num_slices = min(total_slices, s_begin + (tail_split >= 0 ? p.tail_sps : p.slices_per_split));
    if (k_limit != 0x7fffffff) num_slices = min(num_slices, (k_limit + BK - 1) / BK);       // 1x1 problems: slice s = k / BK

    // state of the NEXT slice to stage (advanced incrementally: no divisions inside the K loop)
    int ld_ci0, ld_ky, ld_kx;
    unsigned ld_woff;                                   // k offset inside a weight row
    const bool cmajor = !TAPS && p.cmajor;
    if (cmajor) {
        const int ntap = p.kh * p.kw, cs = s_begin / ntap, tap = s_begin - cs * ntap;
        ld_ci0 = cs * BK;
        ld_ky = tap / p.kw;
        ld_kx = tap - ld_ky * p.kw;
        ld_woff = (unsigned)(tap * p.Ci + ld_ci0);
    } else {
        const int tap = s_begin / slices_per_tap;
        ld_ci0 = (s_begin - tap * slices_per_tap) * BK;
        ld_ky = TAPS ? tap : tap / p.kw;
        ld_kx = TAPS ? 0 : tap - ld_ky * p.kw;
        ld_woff = (unsigned)s_begin * BK;               // tap-major: slices are contiguous in k
    }

    float4 ra[T::PASSES_A], rb[T::PASSES_B];
    unsigned ra_ok = 0u;                              // bit i: pass i of the staged A slice is in range
    // element offset of the row's window origin (mod 2^32: only used when the tap is in range)
    unsigned a_base[T::PASSES_A];
#pragma unroll
    for (int i = 0; i < T::PASSES_A; ++i)
        a_base[i] = (a_img[i] + (unsigned)(a_hi0[i] * p.Wx + a_wi0[i])) * (unsigned)p.Ci + vcol * 4;
    auto load_slice = [&]() {
        // branch-free: out-of-range taps read element 0 (always mapped) and are replaced by zeros afterwards
        const int dy = TAPS ? p.tap_dy[ld_ky] : ld_ky * p.dil;      // ld_ky indexes the tap table
        const int dx = TAPS ? p.tap_dx[ld_ky] : ld_kx * p.dil;
        const unsigned koff = (unsigned)((dy * p.Wx + dx) * p.Ci + ld_ci0);      // wave-uniform
#pragma unroll
        for (int i = 0; i < T::PASSES_A; ++i) {
            const bool ok = a_ok[i] & ((unsigned)(a_hi0[i] + dy) < (unsigned)p.H) & ((unsigned)(a_wi0[i] + dx) < (unsigned)p.W);
            const unsigned off = (a_base[i] + koff) & (0u - (unsigned)ok);
            ra[i] = *reinterpret_cast<const float4 *>(p.x + off);      // zeroed at store time (keeps the load in flight)
            ra_ok = ok ? (ra_ok | (1u << i)) : (ra_ok & ~(1u << i));
        }
        const unsigned woff = TAPS ? (unsigned)(p.tap_w[ld_ky] * p.Ci + ld_ci0) : ld_woff;
#pragma unroll
        for (int i = 0; i < T::PASSES_B; ++i) rb[i] = *reinterpret_cast<const float4 *>(wbase + b_off[i] + woff);
        if (cmajor) {
            ld_woff += (unsigned)p.Ci;
            if (++ld_kx == p.kw) {
                ld_kx = 0;
                if (++ld_ky == p.kh) {
                    ld_ky = 0;
                    ld_ci0 += BK;
                    ld_woff = (unsigned)ld_ci0;
                }
            }
        } else {
            ld_woff += BK;
            ld_ci0 += BK;
            if (ld_ci0 == p.Ci) {
                ld_ci0 = 0;
                if (TAPS) ++ld_ky;
                else if (++ld_kx == p.kw) { ld_kx = 0; ++ld_ky; }
            }
        }
    };
    auto store_slice = [&](int buf) {
        if constexpr (X3) {
            // float4 -> three planes of 4 bf16 (8 bytes each) at [row][plane][vcol * 4]
            unsigned short *la = reinterpret_cast<unsigned short *>(lds);
            unsigned short *lb = la + T::BM * T::ROW_HALFS;
            auto put = [&](unsigned short *row, float4 v) {
                unsigned h0, m0_, l0, h1, m1, l1;
                split3x2(v.x, v.y, h0, m0_, l0);
                split3x2(v.z, v.w, h1, m1, l1);
                *reinterpret_cast<uint2 *>(row + vcol * 4) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(row + BK + vcol * 4) = make_uint2(m0_, m1);
                *reinterpret_cast<uint2 *>(row + 2 * BK + vcol * 4) = make_uint2(l0, l1);
            };
#pragma unroll
            for (int i = 0; i < T::PASSES_A; ++i) {
                const int r = vrow + i * T::ROWS_PER_PASS;
                if (T::BM % T::ROWS_PER_PASS == 0 || r < T::BM) put(la + r * T::ROW_HALFS, keep4((ra_ok >> i) & 1u, ra[i]));
            }
#pragma unroll
            for (int i = 0; i < T::PASSES_B; ++i) {
                const int r = vrow + i * T::ROWS_PER_PASS;
                if (T::BN % T::ROWS_PER_PASS == 0 || r < T::BN) put(lb + r * T::ROW_HALFS, keep4(b_ok[i], rb[i]));
            }
            return;
        }
        float *la = lds;
        float *lb = lds + T::BM * T::LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < T::PASSES_A; ++i) {
            const int r = vrow + i * T::ROWS_PER_PASS;
            if (T::BM % T::ROWS_PER_PASS == 0 || r < T::BM)
                *reinterpret_cast<float4 *>(la + r * T::LDS_STRIDE + vcol * 4) = keep4((ra_ok >> i) & 1u, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < T::PASSES_B; ++i) {
            const int r = vrow + i * T::ROWS_PER_PASS;
            if (T::BN % T::ROWS_PER_PASS == 0 || r < T::BN)
                *reinterpret_cast<float4 *>(lb + r * T::LDS_STRIDE + vcol * 4) = keep4(b_ok[i], rb[i]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    if (s_begin < num_slices) {
        load_slice();
        store_slice(0);
    }
    __syncthreads();
    for (int s = s_begin; s < num_slices; ++s) {
        if (s + 1 < num_slices) load_slice();
        if constexpr (X3) {
            // lane (row frow, half fhalf) takes k = 16 kk + 8 fhalf .. + 7 of its row from each plane: one ds_read_b128
            const unsigned short *lh = reinterpret_cast<const unsigned short *>(lds);
            const unsigned short *la = lh + (wm * TM * 32 + frow) * T::ROW_HALFS + fhalf * 8;
            const unsigned short *lb = lh + (T::BM + wn * TN * 32 + frow) * T::ROW_HALFS + fhalf * 8;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        fa[i][q] = *reinterpret_cast<const bf16x8 *>(la + i * 32 * T::ROW_HALFS + q * BK + kk * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        fb[j][q] = *reinterpret_cast<const bf16x8 *>(lb + j * 32 * T::ROW_HALFS + q * BK + kk * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {      // smallest terms first
                        mfma_x3(acc[i][j], fa[i][2], fb[j][0]);
                        mfma_x3(acc[i][j], fa[i][0], fb[j][2]);
                        mfma_x3(acc[i][j], fa[i][1], fb[j][1]);
                        mfma_x3(acc[i][j], fa[i][1], fb[j][0]);
                        mfma_x3(acc[i][j], fa[i][0], fb[j][1]);
                        mfma_x3(acc[i][j], fa[i][0], fb[j][0]);
                    }
            }
        }
        const float *la = lds + (wm * TM * 32 + frow) * T::LDS_STRIDE + fhalf * 4;
        const float *lb = lds + (T::BM + wn * TN * 32 + frow) * T::LDS_STRIDE + fhalf * 4;
#pragma unroll
        for (int kk = 0; kk < (X3 ? 0 : BK / 8); ++kk) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4 *>(la + i * 32 * T::LDS_STRIDE + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4 *>(lb + j * 32 * T::LDS_STRIDE + kk * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (s + 1 < num_slices) {
            __syncthreads();               // every wave has read this slice
            store_slice(0);                // the prefetched registers (global loads were in flight during the MFMAs)
        }
        __syncthreads();
    }

    if constexpr (X3) {
        // the MFMAs above are inline asm, so the compiler's hazard recogniser does not see them: an 8-pass MFMA result may be
        // read by a VALU instruction only 11 wait states after issue (one s_nop 15 per accumulator, once per tile)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("s_nop 15" : "+a"(acc[i][j]));
    }

    // ---- epilogue.  D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // Accumulators go through LDS (one 32-row band per wave row per round) so that global stores -- and the
    // residual / mask loads -- are 16 B per lane along full output rows instead of 4 B column fragments.
    const bool vec_ok = (p.Co & 3) == 0;
    unsigned omx = 0u;                                // largest magnitude stored to y (p.amax_out)
    constexpr int V = T::BN / 4;                      // float4 per staged row
    constexpr int RPP = 256 / V;                      // staged rows written per pass
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                lds[row * T::EPI_STRIDE + (wn * TN + j) * 32 + frow] = acc[i][j][r];
            }
        __syncthreads();
        const int c4 = tid % V;
        const int n = n0 + c4 * 4;
        constexpr int NPASS = T::EPI_ROWS / RPP;
        if (vec_ok && tail_split < 0 && p.splits <= 1 && p.res_H == 0) {
            // the common form, four row passes at a time with all their residual / mask loads issued first (conv_x3.hip,
            // epilogue: the general loop below is one dependent memory round trip per pass)
            const bool col_ok = n < p.Co;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias != nullptr && col_ok) bv = *reinterpret_cast<const float4 *>(p.bias + n);
            constexpr int UB = NPASS < 4 ? NPASS : 4;
#pragma unroll
            for (int pass0 = 0; pass0 < NPASS; pass0 += UB) {
                float4 v[UB], rv[UB], mv[UB];
                int64_t o[UB];
                bool ok[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int row = tid / V + (pass0 + u) * RPP;
                    const int64_t m = m0 + ((row >> 5) * TM + i) * 32 + (row & 31);
                    ok[u] = m < p.M && col_ok;
                    int64_t oo;
                    if constexpr (TAPS) {
                        const unsigned mm = ok[u] ? (unsigned)m : 0u, wo = mm % (unsigned)p.Wo, t = mm / (unsigned)p.Wo;
                        const unsigned ho = t % (unsigned)p.Ho, b = t / (unsigned)p.Ho;
                        oo = (((int64_t)b * p.oH + p.o_h0 + p.o_step * (int)ho) * p.oW + p.o_w0 + p.o_step * (int)wo) * p.Co + n;
                    } else
                        oo = m * p.Co + n;
                    o[u] = oo & -(int64_t)ok[u];                      // rows / columns past the end: element 0, read and dropped
                    v[u] = *reinterpret_cast<const float4 *>(lds + row * T::EPI_STRIDE + c4 * 4);
                }
                if (p.residual != nullptr) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) rv[u] = *reinterpret_cast<const float4 *>(p.residual + o[u]);
                }
                if (p.mask_src != nullptr) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) mv[u] = *reinterpret_cast<const float4 *>(p.mask_src + o[u]);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) { v[u].x += bv.x; v[u].y += bv.y; v[u].z += bv.z; v[u].w += bv.w; }
                if (p.residual != nullptr) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) { v[u].x += rv[u].x; v[u].y += rv[u].y; v[u].z += rv[u].z; v[u].w += rv[u].w; }
                }
                if (p.relu) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f); v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
                    }
                }
                if (p.mask_src != nullptr) {
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        v[u].x = mv[u].x > 0.f ? v[u].x : 0.f; v[u].y = mv[u].y > 0.f ? v[u].y : 0.f;
                        v[u].z = mv[u].z > 0.f ? v[u].z : 0.f; v[u].w = mv[u].w > 0.f ? v[u].w : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (ok[u]) {
                        *reinterpret_cast<float4 *>(p.y + o[u]) = v[u];
                        omx = htd::mag_bits4(omx, v[u]);
                    }
            }
            if (i + 1 < TM) __syncthreads();
            continue;
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = tid / V + pass * RPP;                    // staged row: wave-row (row >> 5), line (row & 31)
            const int64_t m = m0 + ((row >> 5) * TM + i) * 32 + (row & 31);
            if (m >= p.M || n >= p.Co) continue;
            float4 v = *reinterpret_cast<const float4 *>(lds + row * T::EPI_STRIDE + c4 * 4);
            int64_t o;
            if constexpr (TAPS) {
                const unsigned mm = (unsigned)m, wo = mm % (unsigned)p.Wo, t = mm / (unsigned)p.Wo;
                const unsigned ho = t % (unsigned)p.Ho, b = t / (unsigned)p.Ho;
                o = (((int64_t)b * p.oH + p.o_h0 + p.o_step * (int)ho) * p.oW + p.o_w0 + p.o_step * (int)wo) * p.Co + n;
            } else
                o = m * p.Co + n;
            if (tail_split >= 0) {       // tail unit: raw partial tile, dense [BM][BN]
                const int lrow = ((row >> 5) * TM + i) * 32 + (row & 31);
                float *dst = p.partial + ((int64_t)(bid - p.tail_first) * p.tail_splits + tail_split) * (T::BM * T::BN) +
                             lrow * T::BN + c4 * 4;
                *reinterpret_cast<float4 *>(dst) = v;
                continue;
            }
            if (p.splits > 1) {          // raw partial sums; bias / residual / activation happen in the reduce pass
                float *dst = p.partial + (int64_t)blockIdx.y * p.M * p.Co + o;
                if (vec_ok) *reinterpret_cast<float4 *>(dst) = v;
                else {
                    dst[0] = v.x;
                    if (n + 1 < p.Co) dst[1] = v.y;
                    if (n + 2 < p.Co) dst[2] = v.z;
                    if (n + 3 < p.Co) dst[3] = v.w;
                }
                continue;
            }
            if (vec_ok) {
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                if (p.residual) {
                    int64_t ro = o;
                    if (p.res_H > 0) {
                        const unsigned mm = (unsigned)m, wo = mm % (unsigned)p.Wo, t = mm / (unsigned)p.Wo;
                        const unsigned ho = t % (unsigned)p.Ho, b = t / (unsigned)p.Ho;
                        const int rh = min((int)floorf(ho * p.res_sh), p.res_H - 1);
                        const int rw = min((int)floorf(wo * p.res_sw), p.res_W - 1);
                        ro = (((int64_t)b * p.res_H + rh) * p.res_W + rw) * p.Co + n;
                    }
                    const float4 rv = *reinterpret_cast<const float4 *>(p.residual + ro);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (p.mask_src) {   // data gradient through the ReLU of the layer that produced the conv input
                    const float4 mv = *reinterpret_cast<const float4 *>(p.mask_src + o);
                    v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
                    v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4 *>(p.y + o) = v;
                omx = htd::mag_bits4(omx, v);
            } else {
                auto put = [&](int e, float t) {
                    if (n + e >= p.Co) return;
                    t += p.bias ? p.bias[n + e] : 0.f;
                    if (p.residual) t += p.residual[o + e];
                    if (p.relu) t = fmaxf(t, 0.f);
                    if (p.mask_src) t = p.mask_src[o + e] > 0.f ? t : 0.f;
                    p.y[o + e] = t;
                    omx = max(omx, htd::mag_bits(t));
                };
                put(0, v.x); put(1, v.y); put(2, v.z); put(3, v.w);
            }
        }
        if (i + 1 < TM) __syncthreads();
    }
    if (p.amax_out != nullptr && tail_split < 0 && p.splits <= 1) htd::wave_mag_out(omx, p.amax_out);
}

__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(ConvParams p)
{
    const int64_t total = p.M * p.Co;
    unsigned omx = 0u;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int k = 0; k < p.splits; ++k) v += p.partial[(int64_t)k * total + o];
        const int n = (int)(o % p.Co);
        if (p.bias) v += p.bias[n];
        if (p.residual) {
            int64_t ro = o;
            if (p.res_H > 0) {
                const int64_t m = o / p.Co;
                const unsigned mm = (unsigned)m, wo = mm % (unsigned)p.Wo, t = mm / (unsigned)p.Wo;
                const unsigned ho = t % (unsigned)p.Ho, b = t / (unsigned)p.Ho;
                const int rh = min((int)floorf(ho * p.res_sh), p.res_H - 1);
                const int rw = min((int)floorf(wo * p.res_sw), p.res_W - 1);
                ro = (((int64_t)b * p.res_H + rh) * p.res_W + rw) * p.Co + n;
            }
            v += p.residual[ro];
        }
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.mask_src) v = p.mask_src[o] > 0.f ? v : 0.f;
        p.y[o] = v;
        omx = max(omx, htd::mag_bits(v));
    }
    if (p.amax_out != nullptr) htd::wave_mag_out(omx, p.amax_out);
}

// sums the partial tiles of the balanced tail and applies the epilogue; one thread per output element of a tail tile
__global__ __launch_bounds__(256) void conv_tail_epilogue_kernel(ConvParams p, int BM, int BN)
{
    const int64_t per_tile = (int64_t)BM * BN, total = (int64_t)(p.mt * p.nt - p.tail_first) * per_tile;
    unsigned omx = 0u;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e / per_tile), within = (int)(e - (int64_t)t * per_tile);
        const int tile = p.tail_first + t, tile_m = tile / p.nt, tile_n = tile % p.nt;
        const int64_t m = (int64_t)tile_m * BM + within / BN;
        const int n = tile_n * BN + within % BN;
        if (m >= p.M || n >= p.Co) continue;
        const float *src = p.partial + (int64_t)t * p.tail_splits * per_tile + within;
        float v = 0.f;
        for (int k = 0; k < p.tail_splits; ++k) v += src[(int64_t)k * per_tile];
        const int64_t o = m * p.Co + n;
        if (p.bias) v += p.bias[n];
        if (p.residual) {
            int64_t ro = o;
            if (p.res_H > 0) {
                const unsigned mm = (unsigned)m, wo = mm % (unsigned)p.Wo, tt = mm / (unsigned)p.Wo;
                const unsigned ho = tt % (unsigned)p.Ho, b = tt / (unsigned)p.Ho;
                const int rh = min((int)floorf(ho * p.res_sh), p.res_H - 1);
                const int rw = min((int)floorf(wo * p.res_sw), p.res_W - 1);
                ro = (((int64_t)b * p.res_H + rh) * p.res_W + rw) * p.Co + n;
            }
            v += p.residual[ro];
        }
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.mask_src) v = p.mask_src[o] > 0.f ? v : 0.f;
        p.y[o] = v;
        omx = max(omx, htd::mag_bits(v));
    }
    if (p.amax_out != nullptr) htd::wave_mag_out(omx, p.amax_out);
}

int g_conv_math = getenv("HTD_CONV_MATH") ? atoi(getenv("HTD_CONV_MATH")) : 1;

// Arithmetic of the matrix products: 1 = fp32 through three-way bf16 splits on the bf16 matrix pipe (X3, above),
// 0 = the fp32-input MFMA.  Both give fp32-accurate results; X3 is 2.67x the matrix throughput.  HTD_CONV_MATH / the
// setter below choose; layers whose channel count is not a multiple of 16 (the 8-channel stem) always take 0.
template <int WGM, int WGN, int TM, int TN>
void launch_cfg(const ConvParams &p, unsigned blocks, hipStream_t s)
{
    const dim3 grid(blocks, p.splits);
    const bool x3 = g_conv_math == 1;
    if (p.ntaps > 0) {           // strided data gradient sub-problem: rare, one BK is enough
        if (p.Ci % 16 == 0) {
            if (x3) hipLaunchKernelGGL((conv_igemm_kernel<16, WGM, WGN, TM, TN, true, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((conv_igemm_kernel<16, WGM, WGN, TM, TN, true>), grid, dim3(256), 0, s, p);
        } else
            hipLaunchKernelGGL((conv_igemm_kernel<8, WGM, WGN, TM, TN, true>), grid, dim3(256), 0, s, p);
    } else if (p.Ci % 32 == 0) {
        if (x3) hipLaunchKernelGGL((conv_igemm_kernel<32, WGM, WGN, TM, TN, false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_igemm_kernel<32, WGM, WGN, TM, TN, false>), grid, dim3(256), 0, s, p);
    } else if (p.Ci % 16 == 0) {
        if (x3) hipLaunchKernelGGL((conv_igemm_kernel<16, WGM, WGN, TM, TN, false, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_igemm_kernel<16, WGM, WGN, TM, TN, false>), grid, dim3(256), 0, s, p);
    } else
        hipLaunchKernelGGL((conv_igemm_kernel<8, WGM, WGN, TM, TN, false>), grid, dim3(256), 0, s, p);
}

// split-K plan shared by the launcher and the workspace query: only problems that cannot fill the chip
int plan_splits(int64_t M, int Co, int Ci, int taps)
{
    const int bk = Ci % 32 == 0 ? 32 : (Ci % 16 == 0 ? 16 : 8);
    const int total_slices = taps * (Ci / bk);
    const int64_t tiles = htd::ceil_div(M, 128) * htd::ceil_div(Co, 64);     // 128x64 tiles are used when tiles are few
    if (tiles <= 0 || tiles >= 384 || total_slices < 16) return 1;
    int64_t want = htd::ceil_div(768, tiles);
    want = std::min<int64_t>(want, total_slices / 8);
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, 16));
}

// Tile configurations: id -> (BM, BN) and the wave grid / MFMA blocks per wave of the instantiation
//   0  64x64   2x2 waves of 32x32        3  128x128  2x2 waves of 64x64
//   1  128x32  4x1 waves of 32x32        4  128x64   2x2 waves of 64x32   (one A + ... fragment pair feeds 8 MFMAs)
//   2  128x64  4x1 waves of 32x64        5  64x128   2x2 waves of 32x64
struct TileCfg { int bm, bn; };
constexpr TileCfg kTileCfg[] = {{64, 64}, {128, 32}, {128, 64}, {128, 128}, {128, 64}, {64, 128}};
constexpr int kNumTileCfg = 6;

// HTD_CONV_TUNE=1 makes the launcher re-read HTD_CONV_FORCE_TILE (a configuration id, -1 = automatic) on every call:
// the knob of tools/sweep_conv_tiles.py.  Off (the default) nothing is read after the first call.
static const bool g_tune = getenv("HTD_CONV_TUNE") != nullptr;
int forced_tile()
{
    if (!g_tune) return -1;
    const char *e = getenv("HTD_CONV_FORCE_TILE");
    const int v = e ? atoi(e) : -1;
    return (v >= 0 && v < kNumTileCfg) ? v : -1;
}

// Tile choice.  Narrow outputs (Co <= 64) keep the 4x1-wave tiles.  Otherwise every candidate is scored by
//     base(tile, K) * quantisation(tiles on 256 CUs) * useful fraction of the padded tiles
// and the best one runs.  The numbers come from tools/sweep_conv_tiles.py on MI355X (B = 4 @ 800x1344 layer set):
//   * long reductions (K >= 1024): 128x128 reaches ~122 TF/s, the two 128x64 / 64x128 forms ~120, 64x64 ~104 -- the
//     smaller the tile the more L2 -> LDS traffic and barriers per FLOP;
//   * short reductions (K <= 512, the 1x1 expansions): the 64 KB epilogue of a 128x128 tile is no longer hidden by
//     other workgroups' main loops (84 vs 92 TF/s at K = 256), the small tiles lose nothing;
//   * quantisation: t tiles on 256 CUs take ceil(t / 256) rounds unless the balanced tail (plan_tail, 3..7 rounds)
//     spreads the remainder: M = 16 800 x Co = 256 is 264 tiles of 128x128 (63 TF/s) but 1 052 of 64x64 (104 TF/s),
//     while M = 67 200 x Co = 128 is 1 050 tiles of 64x128 (100 TF/s) against 2 100 of 64x64 (81 TF/s).
static float tile_score(int cfg, int64_t M, int Co, int K, int splits)
{
    const int bm = kTileCfg[cfg].bm, bn = kTileCfg[cfg].bn;
    const int64_t tiles = htd::ceil_div(M, bm) * htd::ceil_div(Co, bn);
    const bool long_k = K >= 1024;
    float base;
    if (g_conv_math == 1) {
        // split-bf16 products: the MFMA phase of a slice is 2.7x shorter, so everything else -- L2 -> LDS traffic, the
        // operand split, barriers -- weighs more and the big tile's 2x lower traffic per FLOP decides
        // (P2 3x3: 183 / 155 / 161 / 131 TF/s for 128x128 / 128x64 / 64x128 / 64x64; K = 256: 128 / 116 / 122 / 109)
        switch (cfg) {
        case 3: base = 1.00f; break;
        case 4: base = long_k ? 0.86f : 0.90f; break;
        case 5: base = long_k ? 0.88f : 0.94f; break;
        default: base = long_k ? 0.72f : 0.82f; break;
        }
    } else {
        switch (cfg) {
        case 3: base = long_k ? 1.00f : 0.90f; break;
        case 4: base = long_k ? 0.98f : 0.97f; break;
        case 5: base = long_k ? 0.98f : 0.98f; break;
        default: base = long_k ? 0.86f : (K >= 512 ? 0.90f : 0.95f); break;
        }
    }
    const float w = (float)(tiles * splits) / 256.f;        // split-K: every tile is `splits` workgroups
    float quant;
    if (w >= 7.f) quant = w / ceilf(w);
    else if (w >= 3.f) quant = 0.97f;                         // balanced tail
    else quant = w / ceilf(w);
    const float useful = (float)((double)M * Co / ((double)tiles * bm * bn));
    return base * quant * useful;
}

// Tuned tile table: (M, Co, Ci, taps, epilogue bits) -> configuration id, filled through htd_conv2d_tile_table_set by
// htd_amd/tuning.py from the table tools/tune_conv_tiles.py measured INSIDE the train / inference step on MI355X (the
// epilogue variant and the cache state the neighbouring kernels leave matter: stand-alone sweeps mis-rank the short-K
// layers).  Problems that are not in the table fall back to the score below.
struct TileKey {
    int64_t M;
    int Co, Ci, taps, epi;
    bool operator==(const TileKey &o) const { return M == o.M && Co == o.Co && Ci == o.Ci && taps == o.taps && epi == o.epi; }
};
struct TileKeyHash {
    size_t operator()(const TileKey &k) const
    {
        uint64_t h = (uint64_t)k.M * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t)k.Co << 40) ^ ((uint64_t)k.Ci << 20) ^ ((uint64_t)k.taps << 4) ^ (uint64_t)k.epi;
        h *= 0xBF58476D1CE4E5B9ull;
        return (size_t)(h ^ (h >> 29));
    }
};
std::mutex g_tile_mutex;
std::unordered_map<TileKey, int, TileKeyHash> g_tile_table;

int table_tile(int64_t M, int Co, int Ci, int taps, int epi)
{
    std::lock_guard<std::mutex> lock(g_tile_mutex);
    if (g_tile_table.empty()) return -1;
    const auto it = g_tile_table.find(TileKey{M, Co, Ci, taps, epi});
    return it == g_tile_table.end() ? -1 : it->second;
}

// epi < 0: the problem cannot use the table (tap-table data gradients, batched GEMMs)
int choose_tile(int64_t M, int Co, int Ci, int taps, int epi, int splits, int &bm, int &bn)
{
    const int K = taps * Ci;
    int cfg = forced_tile();
    if (cfg < 0 && epi >= 0) cfg = table_tile(M, Co, Ci, taps, epi);
    if (cfg >= 0 && Co <= 64 && kTileCfg[cfg].bn > 64) cfg = -1;      // a table entry must still fit the kernel
    if (cfg < 0) {
        if (Co <= 32) cfg = 1;
        else if (Co <= 64) cfg = (htd::ceil_div(M, 128) * htd::ceil_div(Co, 64) < 4400 && M >= 2048) ? 0 : 2;
        else {
            static const int cand[4] = {3, 4, 5, 0};
            float best = -1.f;
            for (int c : cand) {
                const float sc = tile_score(c, M, Co, K, splits);
                if (sc > best * 1.005f) { best = sc; cfg = c; }      // ties go to the earlier (larger) tile
            }
        }
    }
    bm = kTileCfg[cfg].bm;
    bn = kTileCfg[cfg].bn;
    return cfg;
}

// Balanced tail: with t tiles on 256 CUs and t / 256 small, the CUs that receive ceil(t / 256) tiles set the run time.
// Keep floor(t / 256) * 256 tiles whole and cut the remaining ones into K ranges, about 256 units in all.
// -> number of whole tiles (0: no tail), splits and slices per split of the tail tiles
static const bool g_no_tail = getenv("HTD_CONV_NO_TAIL") != nullptr;
int plan_tail(int64_t M, int Co, int Ci, int taps, int bm, int bn, int &tail_splits, int &tail_sps)
{
    tail_splits = tail_sps = 0;
    if (g_no_tail || (Co & 3)) return 0;
    const int bk = Ci % 32 == 0 ? 32 : (Ci % 16 == 0 ? 16 : 8);
    const int total_slices = taps * (Ci / bk);
    const int64_t tiles = htd::ceil_div(M, bm) * htd::ceil_div(Co, bn);
    const int64_t rem = tiles % 256, whole = tiles - rem;
    // measured on the HTD layer set: pays at 3..6 whole tiles per CU (M = 16 800 layers: +10 %); with fewer the CUs are not
    // saturated and an extra resident tile costs little, with more the dispatcher's own balancing already hides it
    if (tiles < 256 * 3 || tiles >= 256 * 7 || rem == 0 || rem > 128 || total_slices < 8) return 0;
    int want = (int)std::min<int64_t>(16, std::max<int64_t>(2, (256 + rem / 2) / rem));
    want = std::min(want, total_slices / 4);
    if (want < 2) return 0;
    tail_sps = (int)htd::ceil_div(total_slices, want);
    tail_splits = (int)htd::ceil_div(total_slices, tail_sps);
    return (int)whole;
}

int launch_conv(ConvParams p, hipStream_t s, void *workspace)
{
    const int bk = (p.Ci % 32 == 0 && p.ntaps == 0) ? 32 : (p.Ci % 16 == 0 ? 16 : 8);
    const int total_slices = (p.ntaps > 0 ? p.ntaps : p.kh * p.kw) * (p.Ci / bk);
    p.splits = (workspace && p.ntaps == 0) ? plan_splits(p.M, p.Co, p.Ci, p.kh * p.kw) : 1;
    p.slices_per_split = (int)htd::ceil_div(total_slices, p.splits);
    p.splits = (int)htd::ceil_div(total_slices, p.slices_per_split);
    int bm, bn;
    const int epi = (p.ntaps > 0 || p.w_bstride > 0) ? -1 : ((p.residual ? 1 : 0) | (p.mask_src ? 2 : 0));
    const int cfg = choose_tile(p.M, p.Co, p.Ci, p.ntaps > 0 ? p.ntaps : p.kh * p.kw, epi, p.splits, bm, bn);
    p.mt = (int)htd::ceil_div(p.M, bm);
    p.nt = (int)htd::ceil_div(p.Co, bn);
    const int64_t blocks = (int64_t)p.mt * p.nt;
    HTD_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv2d: bad grid");
    HTD_REQUIRE((int64_t)p.B * p.Hx * p.Wx * p.Ci < (1ll << 31) && (int64_t)p.Co * p.kh * p.kw * p.Ci < (1ll << 31) &&
                    p.M < (1ll << 31),
                "conv2d: operand larger than 2^31 elements (32-bit element offsets)");
    p.partial = (float *)workspace;
    static const int cmajor_env = getenv("HTD_CONV_CMAJOR") ? atoi(getenv("HTD_CONV_CMAJOR")) : 1;
    p.cmajor = (p.ntaps == 0 && p.kh * p.kw > 1) ? cmajor_env : 0;
    unsigned launch_blocks = (unsigned)blocks;
    p.tail_first = p.tail_splits = p.tail_sps = 0;
    if (workspace && p.ntaps == 0 && p.splits == 1 && p.w_bstride == 0) {
        const int whole = plan_tail(p.M, p.Co, p.Ci, p.kh * p.kw, bm, bn, p.tail_splits, p.tail_sps);
        if (p.tail_splits > 0) {
            p.tail_first = whole;
            launch_blocks = (unsigned)(whole + (blocks - whole) * p.tail_splits);
        }
    }
    switch (cfg) {
    case 0: launch_cfg<2, 2, 1, 1>(p, launch_blocks, s); break;
    case 1: launch_cfg<4, 1, 1, 1>(p, launch_blocks, s); break;
    case 2: launch_cfg<4, 1, 1, 2>(p, launch_blocks, s); break;
    case 3: launch_cfg<2, 2, 2, 2>(p, launch_blocks, s); break;
    case 4: launch_cfg<2, 2, 2, 1>(p, launch_blocks, s); break;
    default: launch_cfg<2, 2, 1, 2>(p, launch_blocks, s); break;
    }
    if (p.tail_splits > 0) {
        const int64_t elems = (blocks - p.tail_first) * (int64_t)bm * bn;
        hipLaunchKernelGGL(conv_tail_epilogue_kernel, dim3((unsigned)htd::ceil_div(elems, 256)), dim3(256), 0, s, p, bm, bn);
    }
    if (p.splits > 1) {
        const unsigned rb = (unsigned)std::min<int64_t>(htd::ceil_div(p.M * p.Co, 256), 4096);
        hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3(rb), dim3(256), 0, s, p);
    }
    return htd::check_launch("conv2d");
}

}  // namespace

namespace htd {
int conv_math() { return g_conv_math; }      // conv_x3.hip: its kernels exist for the split-bf16 arithmetic only
}

// res_h / res_w > 0: residual is a [B][res_h][res_w][Co] map added through nearest-neighbour up-sampling to the
// output size (0, 0: residual has the output's shape).
static int conv2d_fwd_impl(const float *x, const float *w, const float *bias, const float *residual, int res_h, int res_w, float *y,
                           int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int relu,
                           float *amax_out, void *workspace, void *stream)
{
    HTD_REQUIRE((res_h > 0) == (res_w > 0) && res_h >= 0 && (res_h == 0 || ((Co & 3) == 0 && residual)),
                "conv2d_fwd: bad residual up-sampling arguments");
    HTD_REQUIRE(B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0 && pad >= 0,
                "conv2d_fwd: bad sizes B=%d H=%d W=%d Ci=%d Co=%d k=%dx%d s=%d p=%d d=%d", B, H, W, Ci, Co, kh, kw,
                stride, pad, dil);
    HTD_REQUIRE(Ci % 8 == 0, "conv2d_fwd: Ci=%d must be a multiple of 8 (pad the stem input to 8 channels)", Ci);
    HTD_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
    ConvParams p{};
    p.x = x; p.w = w; p.bias = bias; p.residual = residual; p.mask_src = nullptr; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.dil = dil;
    p.Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    p.Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    HTD_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d_fwd: empty output");
    p.Hx = H; p.Wx = W; p.relu = relu;
    p.amax_out = amax_out;
    p.M = (int64_t)B * p.Ho * p.Wo;
    if (res_h > 0) {
        p.res_H = res_h; p.res_W = res_w;
        p.res_sh = (float)res_h / (float)p.Ho; p.res_sw = (float)res_w / (float)p.Wo;
    }
    return launch_conv(p, (hipStream_t)stream, workspace);
}

extern "C" int htd_conv2d_fwd(const float *x, const float *w, const float *bias, const float *residual, int res_h,
                              int res_w, float *y, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride,
                              int pad, int dil, int relu, void *workspace, void *stream)
{
    return conv2d_fwd_impl(x, w, bias, residual, res_h, res_w, y, B, H, W, Ci, Co, kh, kw, stride, pad, dil, relu, nullptr, workspace,
                           stream);
}

// htd_conv2d_fwd that also leaves max |y| in *amax_out (a device scalar holding zero or an earlier maximum on entry): the layers
// this kernel keeps (strided, dilated, skinny) hand their output to layers that run on the H2 arithmetic (htd_conv2d_fwd_x3h)
extern "C" int htd_conv2d_fwd_amax(const float *x, const float *w, const float *bias, const float *residual, int res_h, int res_w,
                                   float *y, float *amax_out, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                                   int dil, int relu, void *workspace, void *stream)
{
    HTD_REQUIRE(amax_out, "conv2d_fwd: null maximum");
    return conv2d_fwd_impl(x, w, bias, residual, res_h, res_w, y, B, H, W, Ci, Co, kh, kw, stride, pad, dil, relu, amax_out, workspace,
                           stream);
}

// bytes of split-K workspace htd_conv2d_fwd / htd_conv2d_bwd_data may use for this problem (0 = none needed);
// passing workspace = NULL is always allowed and disables split-K.
extern "C" int64_t htd_conv2d_workspace_bytes(int64_t M, int Co, int Ci, int kh, int kw)
{
    const int splits = plan_splits(M, Co, Ci, kh * kw);
    if (splits > 1) return (int64_t)splits * M * Co * 4;
    int64_t need = 0;
    for (int epi = 0; epi < 4; ++epi) {          // the caller does not say which epilogue the launch will have
        int bm, bn, ts, tsps;
        choose_tile(M, Co, Ci, kh * kw, epi, 1, bm, bn);
        const int whole = plan_tail(M, Co, Ci, kh * kw, bm, bn, ts, tsps);
        if (ts == 0) continue;
        const int64_t tiles = htd::ceil_div(M, bm) * htd::ceil_div(Co, bn);
        need = std::max(need, (tiles - whole) * ts * (int64_t)bm * bn * 4);
    }
    return need;
}

// Tuned tile table (see choose_tile).  cfg: 0 64x64, 1 128x32, 2 128x64 (4x1 waves), 3 128x128, 4 128x64 (2x2 waves),
// 5 64x128; cfg < 0 erases the entry.  epi: bit 0 = residual / accum operand present, bit 1 = mask_src present.
extern "C" int htd_conv2d_tile_table_set(int64_t M, int Co, int Ci, int taps, int epi, int cfg)
{
    HTD_REQUIRE(M > 0 && Co > 0 && Ci > 0 && taps > 0 && epi >= 0 && epi < 4 && cfg < kNumTileCfg,
                "tile_table_set: bad entry M=%lld Co=%d Ci=%d taps=%d epi=%d cfg=%d", (long long)M, Co, Ci, taps, epi, cfg);
    std::lock_guard<std::mutex> lock(g_tile_mutex);
    if (cfg < 0) g_tile_table.erase(TileKey{M, Co, Ci, taps, epi});
    else g_tile_table[TileKey{M, Co, Ci, taps, epi}] = cfg;
    return HTD_OK;
}

extern "C" int htd_conv2d_tile_table_clear()
{
    std::lock_guard<std::mutex> lock(g_tile_mutex);
    g_tile_table.clear();
    return HTD_OK;
}

// the configuration id a launch of this problem would use now (table, then score); splits as planned for it
extern "C" int htd_conv2d_tile_query(int64_t M, int Co, int Ci, int taps, int epi)
{
    int bm, bn;
    return choose_tile(M, Co, Ci, taps, epi, plan_splits(M, Co, Ci, taps), bm, bn);
}

// 1: fp32 via three-way bf16 splits on the bf16 matrix pipe (default), 0: fp32-input MFMA.  Returns the previous mode.
extern int g_wgrad_math;
extern "C" int htd_conv2d_set_math(int mode)
{
    const int prev = g_conv_math;
    if (mode == 0 || mode == 1) g_conv_math = g_wgrad_math = mode;
    return prev;
}

// Batched NT GEMM on the same kernel: c[g] = a[g] @ b[g]^T, a [G][M][K], b [G][N][K], c [G][M][N].
extern "C" int htd_bgemm_nt(const float *a, const float *b, float *c, int G, int M, int N, int K, void *stream)
{
    HTD_REQUIRE(G > 0 && M > 0 && N > 0 && K > 0, "bgemm_nt: bad sizes G=%d M=%d N=%d K=%d", G, M, N, K);
    HTD_REQUIRE(K % 8 == 0, "bgemm_nt: K=%d must be a multiple of 8", K);
    HTD_REQUIRE(G == 1 || M % 128 == 0, "bgemm_nt: M=%d must be a multiple of 128 when G > 1 (tiles may not straddle groups)", M);
    HTD_REQUIRE(a && b && c, "bgemm_nt: null pointer");
    ConvParams p{};
    p.x = a; p.w = b; p.y = c;
    p.B = G; p.H = M; p.W = 1; p.Ci = K; p.Co = N; p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0; p.dil = 1;
    p.Ho = M; p.Wo = 1; p.Hx = M; p.Wx = 1; p.relu = 0;
    p.w_bstride = (int64_t)N * K;
    p.M = (int64_t)G * M;
    return launch_conv(p, (hipStream_t)stream, nullptr);
}

// The same over zero-padded groups: counts[g] (device, int64) entries of group g are real.  limit bits: 1 = rows of a / c,
// 2 = rows of b = columns of c, 4 = the reduction index; what lies beyond the count must be zero in the operands (it is
// not read) and is written as zeros in c.  PGraph's three contractions (htd_bbox_head.py:210,213-216) at B = 64 x 512
// proposals pad 256 groups to the largest image: 584 GFLOP issued for 123 GFLOP of real groups without this.
extern "C" int htd_bgemm_nt_counts(const float *a, const float *b, float *c, int G, int M, int N, int K,
                                   const int64_t *counts, int limit, void *stream)
{
    HTD_REQUIRE(G > 0 && M > 0 && N > 0 && K > 0, "bgemm_nt_counts: bad sizes G=%d M=%d N=%d K=%d", G, M, N, K);
    HTD_REQUIRE(K % 8 == 0 && M % 128 == 0, "bgemm_nt_counts: K=%d %% 8, M=%d %% 128", K, M);
    HTD_REQUIRE(a && b && c && counts && limit >= 0 && limit < 8, "bgemm_nt_counts: bad arguments");
    ConvParams p{};
    p.x = a; p.w = b; p.y = c;
    p.B = G; p.H = M; p.W = 1; p.Ci = K; p.Co = N; p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0; p.dil = 1;
    p.Ho = M; p.Wo = 1; p.Hx = M; p.Wx = 1; p.relu = 0;
    p.w_bstride = (int64_t)N * K;
    p.M = (int64_t)G * M;
    p.group_count = counts;
    p.lim = limit;
    return launch_conv(p, (hipStream_t)stream, nullptr);
}

// Data gradient: gx[B][H][W][Ci] from gy[B][Ho][Wo][Co] and wT[Ci][kh][kw][Co] = spatially flipped,
// channel-transposed weights (htd_conv2d_flip_weights).  mask_src (may be NULL): gx is zeroed where
// mask_src <= 0 -- the ReLU of the layer that produced the conv input, fused into this epilogue.
// accum (may be NULL; stride 1 only): added to the data gradient before the mask (the other branch of a residual join).
static int conv2d_bwd_data_impl(const float *gy, const float *wT, const float *mask_src, const float *accum, float *gx, int B, int H,
                                int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, float *amax_out, void *workspace,
                                void *stream)
{
    HTD_REQUIRE(!accum || stride == 1, "conv2d_bwd_data: accum needs stride 1");
    HTD_REQUIRE(B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0 && pad >= 0,
                "conv2d_bwd_data: bad sizes");
    HTD_REQUIRE(Co % 8 == 0, "conv2d_bwd_data: Co=%d must be a multiple of 8", Co);
    HTD_REQUIRE(gy && wT && gx, "conv2d_bwd_data: null pointer");
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    ConvParams p{};
    p.x = gy; p.w = wT; p.bias = nullptr; p.residual = accum; p.mask_src = mask_src; p.y = gx;
    p.B = B; p.Ci = Co; p.Co = Ci; p.kh = kh; p.kw = kw; p.dil = dil; p.relu = 0;
    p.amax_out = amax_out;               // (strided form: every parity class's launch raises the same scalar; the rest is zeros)
    p.Hx = Ho; p.Wx = Wo;
    HTD_REQUIRE(kh == kw, "conv2d_bwd_data: square kernels only");
    if (stride == 1) {
        // a stride-1 correlation of gy with the flipped weights: pad' = dil*(k-1) - pad
        p.stride = 1;
        p.pad = dil * (kh - 1) - pad;
        HTD_REQUIRE(p.pad >= 0, "conv2d_bwd_data: pad > dil*(k-1) unsupported");
        p.H = Ho; p.W = Wo; p.Ho = H; p.Wo = W;
        p.M = (int64_t)B * H * W;
        return launch_conv(p, (hipStream_t)stream, workspace);
    }
    // stride s: output pixels fall into s*s parity classes; class (ph, pw) only receives taps with
    // (ph + pad - ky*dil) % s == 0, and those form a dense stride-1 problem on the sub-grid hi = ph + s*i.
    // Every tap belongs to exactly one class, so the total work is 1/s^2 of the zero-stuffed formulation.
    HTD_REQUIRE(kh * kw <= 16 * stride * stride, "conv2d_bwd_data: kernel too large for the tap table");
    bool need_zero = false;
    for (int cls = 0; cls < stride * stride && !need_zero; ++cls) {
        const int ph = cls / stride, pw = cls % stride;
        int ny = 0, nx = 0;
        for (int k = 0; k < kh; ++k) {
            ny += ((ph + pad - k * dil) % stride + stride) % stride == 0;
            nx += ((pw + pad - k * dil) % stride + stride) % stride == 0;
        }
        need_zero = (ny == 0 || nx == 0);
    }
    if (need_zero) {     // classes without taps (1x1 stride-2 shortcuts) are plain zeros
        if (hipMemsetAsync(gx, 0, (size_t)B * H * W * Ci * 4, (hipStream_t)stream) != hipSuccess) {
            htd::set_error("conv2d_bwd_data: memset failed");
            return HTD_ERR_LAUNCH;
        }
    }
    for (int cls = 0; cls < stride * stride; ++cls) {
        const int ph = cls / stride, pw = cls % stride;
        if (ph >= H || pw >= W) continue;
        ConvParams q = p;
        q.ntaps = 0;
        for (int ky = 0; ky < kh; ++ky) {
            const int ry = ph + pad - ky * dil;
            if (((ry % stride) + stride) % stride != 0) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int rx = pw + pad - kx * dil;
                if (((rx % stride) + stride) % stride != 0) continue;
                q.tap_dy[q.ntaps] = ry >= 0 ? ry / stride : -((-ry) / stride);
                q.tap_dx[q.ntaps] = rx >= 0 ? rx / stride : -((-rx) / stride);
                q.tap_w[q.ntaps] = (kh - 1 - ky) * kw + (kw - 1 - kx);      // index into the flipped weights
                ++q.ntaps;
            }
        }
        if (q.ntaps == 0) continue;
        q.stride = 1; q.pad = 0;
        q.H = Ho; q.W = Wo;                                  // bounds of the gy map
        q.Ho = (H - ph + stride - 1) / stride; q.Wo = (W - pw + stride - 1) / stride;
        q.o_h0 = ph; q.o_w0 = pw; q.o_step = stride; q.oH = H; q.oW = W;
        q.M = (int64_t)B * q.Ho * q.Wo;
        const int st = launch_conv(q, (hipStream_t)stream, nullptr);
        if (st) return st;
    }
    return HTD_OK;
}

extern "C" int htd_conv2d_bwd_data(const float *gy, const float *wT, const float *mask_src, const float *accum, float *gx,
                                   int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                                   void *workspace, void *stream)
{
    return conv2d_bwd_data_impl(gy, wT, mask_src, accum, gx, B, H, W, Ci, Co, kh, kw, stride, pad, dil, nullptr, workspace, stream);
}

// htd_conv2d_bwd_data that also leaves max |gx| in *amax_out (zero or an earlier maximum on entry), see htd_conv2d_fwd_amax
extern "C" int htd_conv2d_bwd_data_amax(const float *gy, const float *wT, const float *mask_src, const float *accum, float *gx,
                                        float *amax_out, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                                        int dil, void *workspace, void *stream)
{
    HTD_REQUIRE(amax_out, "conv2d_bwd_data: null maximum");
    return conv2d_bwd_data_impl(gy, wT, mask_src, accum, gx, B, H, W, Ci, Co, kh, kw, stride, pad, dil, amax_out, workspace, stream);
}

namespace {
__global__ __launch_bounds__(256) void flip_weights_kernel(const float *__restrict__ w, float *__restrict__ wT, int Co,
                                                           int taps, int Ci, int Cop)
{
    // w[co][t][ci] -> wT[ci][taps-1-t][co], rows of wT padded with zeros to Cop >= Co; 32x32 LDS transpose per
    // (t, co-tile, ci-tile)
    __shared__ float tile[32][33];
    const int t = blockIdx.z, co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Co && ci < Ci) ? w[((int64_t)co * taps + t) * Ci + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Ci && co < Cop) wT[((int64_t)ci * taps + (taps - 1 - t)) * Cop + co] = tile[tx][r];
    }
}

// y[row][0..Cp) = x[row][0..C) followed by zeros (the reduction channels of a skinny head's data gradient)
__global__ __launch_bounds__(256) void pad_channels_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t total,
                                                           int C, int Cp)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        y[i] = c < C ? x[(i / Cp) * C + c] : 0.f;
    }
}
}  // namespace

extern "C" int htd_conv2d_flip_weights(const float *w, float *wT, int Co, int kh, int kw, int Ci, void *stream)
{
    HTD_REQUIRE(w && wT && Co > 0 && Ci > 0 && kh > 0 && kw > 0, "flip_weights: bad arguments");
    dim3 grid((unsigned)htd::ceil_div(Ci, 32), (unsigned)htd::ceil_div(Co, 32), (unsigned)(kh * kw));
    hipLaunchKernelGGL(flip_weights_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wT, Co, kh * kw, Ci, Co);
    return htd::check_launch("flip_weights");
}

extern "C" int htd_conv2d_flip_weights_padded(const float *w, float *wT, int Co, int Co_padded, int kh, int kw, int Ci,
                                              void *stream)
{
    HTD_REQUIRE(w && wT && Co > 0 && Co_padded >= Co && Ci > 0 && kh > 0 && kw > 0, "flip_weights_padded: bad arguments");
    dim3 grid((unsigned)htd::ceil_div(Ci, 32), (unsigned)htd::ceil_div(Co_padded, 32), (unsigned)(kh * kw));
    hipLaunchKernelGGL(flip_weights_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wT, Co, kh * kw, Ci, Co_padded);
    return htd::check_launch("flip_weights_padded");
}

extern "C" int htd_pad_channels(const float *x, float *y, int64_t rows, int C, int C_padded, void *stream)
{
    HTD_REQUIRE(x && y && rows > 0 && C > 0 && C_padded >= C, "pad_channels: bad arguments");
    const int64_t total = rows * C_padded;
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 8192);
    hipLaunchKernelGGL(pad_channels_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, total, C, C_padded);
    return htd::check_launch("pad_channels");
}
