// fp32 convolution on the bf16 matrix pipe with PRE-SPLIT weights and a halo-staged activation operand (round 3).
//
// Same arithmetic as the X3 form of conv_igemm_kernel (conv_fwd.hip): every fp32 operand is the sum of three bf16 numbers,
// a product is six v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the result is fp32-accurate.  What changes is where the
// split happens.  conv_igemm_kernel cuts every staged float4 into its three pieces while a K slice is written to LDS -- once
// per tile AND per filter tap: an input element of a 3x3 layer is split 9 taps x (Co / BN) times, a weight once per M tile
// (2 100 times on a P2-sized map), and that vector work is serialised with the MFMAs (matrix pipe 36 % busy).
// Here
//   * the weights are split ONCE PER STEP by htd_conv2d_x3_planes into an image [tap][Ci/16][3 planes x 2 halves][Co][8 bf16]
//     whose 16-byte chunks go global -> LDS with global_load_lds_dwordx4: no vector instruction, no register, no ds_write
//     on the B side of the loop;
//   * the activations of a stride-1 "same" convolution are staged as a HALO RUN: the BM consecutive output pixels of a tile
//     read, for filter row ky, the BM + kw - 1 consecutive input pixels  m0 - pad + (ky - pad) * W ...  (NHWC: a pixel shift
//     is a row shift of the operand), so one staged-and-split run serves the kw taps of the row -- the split work and the
//     L2 -> LDS traffic of a 3x3 layer drop 3x; image borders are handled per lane: a lane whose tap falls outside the map
//     reads a row of zeros instead (one v_cndmask on the LDS address);
//   * 1x1 layers (any stride) and Linear layers run on the same kernel with kw = 1.
// K loop: step = (16-channel slice, filter row); per step one A run (double-buffered in LDS, loaded into registers one step
// ahead, split when written), per tap one B tile (double-buffered, LDS-DMA one tap ahead), ONE barrier per tap.
// LDS images are chunk-major, [chunk][row][16 B]: the 16 lanes of a ds_read_b128 group read 256 consecutive bytes
// (conflict-free without padding, which LDS-DMA could not honour), and the split's ds_write_b64 are conflict-free with a
// row pitch = 4 (mod 8).
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include <stdlib.h>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

constexpr int XK = 16;            // channels per K slice
constexpr int NCH = 6;            // 16-byte chunks of a row slice: 3 planes x 2 halves of 8 channels

// ---- "H2" arithmetic (round 4): fp32 products through TWO fp16 pieces per operand, three matrix instructions instead of six ----
// a = (a0 + a1) / sa with a0 = fp16(sa a), a1 = fp16(sa a - a0): 11 + 11 significant bits, |a - (a0 + a1) / sa| <= 2^-22 |a|, and
// a b ~ (a0 b0 + a0 b1 + a1 b0) / (sa sb); the dropped a1 b1 is <= 2^-22 |a b| (2^-24 rms), the size of the terms the bf16 form
// drops.  fp16 has 5 exponent bits, so the operands are block-scaled by powers of two (exact): sa from the largest magnitude of
// the WHOLE activation tensor (a device scalar, htd_absmax, so that sa |a| < 2^15), sb[n] per output channel of the weights;
// the epilogue multiplies column n by 1 / (sa sb[n]).  Elements more than 2^29 below their tensor's maximum fall into fp16's
// subnormals (kept by the matrix pipe: tools/micro/mfma_split_products.hip) and lose relative precision from there on -- their
// ABSOLUTE error stays <= 2^-40 of the tensor maximum, far below fp32's rounding of any sum the large elements take part in.
// Why: under the socket power cap the six-product form's matrix-pipe roof is 284 algorithmic TF/s, this one's 539
// (same micro-benchmark), and the 3x3 layers ran at 75 % of the former.
struct H2Scale { float s, inv; };
__device__ __forceinline__ H2Scale h2_scale(const float *amax)
{
    const unsigned E = (__float_as_uint(*amax) >> 23) & 0xffu;        // amax = 1.m x 2^(E - 127) < 2^(E - 126)
    int e = E == 0u ? 126 : (E == 255u ? 0 : 141 - (int)E);            // 2^e amax in [2^14, 2^15)
    e = e > 126 ? 126 : e;
    return H2Scale{__uint_as_float((unsigned)(127 + e) << 23), __uint_as_float((unsigned)(127 - e) << 23)};
}
// two (scaled) floats -> two packed fp16 pairs (element 0 in the low half): the value and what it left over
__device__ __forceinline__ void split2hx2(float a, float b, unsigned &h, unsigned &l)
{
    union { f16x2 v; unsigned u; } c;
    c.v = __builtin_convertvector(f32x2{a, b}, f16x2);
    h = c.u;
    const f32x2 back = __builtin_convertvector(c.v, f32x2);
    c.v = __builtin_convertvector(f32x2{a - back[0], b - back[1]}, f16x2);
    l = c.u;
}

// two floats -> three packed bf16 pairs (element 0 in the low half), see conv_fwd.hip
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

// s_waitcnt vmcnt(N): all but the wave's N youngest vector-memory operations are done.  volatile asm statements keep their
// order; landed() after the wait names the registers the inline-asm loads fill, so no use of them is scheduled above it and
// the compiler has no reason to copy them earlier (a wait that took them as operands in two branches made hipcc insert
// v_mov copies AHEAD of one of the waits: stale data).
#ifndef HTD_X3P_EARLY
#define HTD_X3P_EARLY 1
#endif
#ifndef HTD_X3H_NB
#define HTD_X3H_NB 2             // weight-tile buffers of the H2 kernels (prefetch distance + 1)
#endif
#ifndef HTD_X3H_FRAG_FIRST
#define HTD_X3H_FRAG_FIRST 1     // H2: a tap's fragment reads ahead of its bookkeeping (0: where the MFMAs are)
#endif
#ifndef HTD_X3H_OCC4
#define HTD_X3H_OCC4 0          // 1: the 128x128 H2 tile at four workgroups per CU (128 VGPRs)
#endif

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void landed(f32x4 &a) { asm volatile("" : "+v"(a)::"memory"); }

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes at `src` go to LDS byte address lds_dst + 16 l (lds_dst wave-uniform, in M0).
// Inline asm for the same reason as the loads above: hipcc puts s_waitcnt vmcnt(0) in front of any LDS access it cannot
// prove disjoint from a pending LDS-DMA of its own -- with a ring of B buffers that is every fragment read, i.e. the
// prefetched tile was waited for in the middle of the current tap's MFMAs.  M0 is saved and restored (hipcc owns it).
__device__ __forceinline__ void lds_dma16(const void *src, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
// ds_write_b64 the compiler does not see: it puts s_waitcnt vmcnt(0) in front of its own LDS stores while an LDS-DMA is in
// flight (possible overlap), which would drain the prefetched B tiles at every tap.  lds_barrier() waits lgkmcnt(0).
template <int OFF>
__device__ __forceinline__ void lds_store8(unsigned addr, unsigned lo, unsigned hi)
{
    const u32x2 v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

struct X3Params {
    const float *x;            // [B][Hx][Wx][Ci] fp32
    const uint4 *wp;           // weight planes [kh*kw][Ci/16][6][Cop] x 16 B
    const float *bias, *residual, *mask_src;
    float *y;
    int Hx, Wx;                // input map
    int Ho, Wo;                // output map (== input map when kw > 1)
    int Ci, Co, Cop;
    int kh;                    // filter rows; the filter width is the template parameter KW
    int stride;                // kw == 1 only
    // kw == 1, ntl > 0: TAP-LIST mode (strided 3x3 layers and the parity classes of their data gradients on the 1x1 loop): the
    // step's kh "rows" are ntl = kh taps; tap t multiplies input pixel (ho * stride + tl_dy[t], wo * stride + tl_dx[t]) -- zeros
    // outside the map -- with tap tl_w[t] of the weight image (whose tiles are [tap][slice] for every filter shape)
    int ntl;
    int tl_dy[9], tl_dx[9], tl_w[9];
    // o_step > 0: output row m = pixel (ho, wo) of the launch's Ho x Wo sub-grid is written to pixel (o_h0 + o_step ho, o_w0 +
    // o_step wo) of an oH x oW map (one parity class of a strided data gradient); residual / mask_src are read there too
    int o_step, o_h0, o_w0, oH, oW;
    int relu;
    int res_H, res_W;          // > 0: residual is a coarser map read through nearest up-sampling (FPN top-down)
    float res_sh, res_sw;
    int64_t M;                 // B*Ho*Wo
    int mt, nt;
    int ncs;                   // Ci / 16
    // Work decomposition (plan_x3p).  Tiles [0, tiles_a) -- region A, output rows < m_rem0 -- are cut along K into splits_a
    // ranges of sps_a steps, the later tiles -- region B -- into splits_b ranges of sps_b steps; one workgroup per (tile,
    // range).  A region with one range writes y; otherwise its workgroups write partial sums, [split][row][Co] per region
    // (region A first), which conv_x3p_splitk_epilogue_kernel adds in split order.
    int tiles_a, splits_a, sps_a, splits_b, sps_b;
    int64_t m_rem0;
    float *partial;
    // Activation planes (round 4).  xp: the A operand ALREADY split -- [Ci/16][6][xp_rows] x 16 B, chunk = plane * 2 + half as in
    // the weight image, row = pixel -- read by conv_x3q_kernel with LDS-DMA (1x1, stride 1 only).  yp: the epilogue also
    // writes the planes of the final values it stores to y, in the same layout with yp_rows rows per chunk array, for a 1x1
    // consumer of y (Co % 16 == 0).  Rows >= M of an image are never written and never matter: a 1x1 output row depends on its
    // own input row only.
    const uint4 *xp;
    uint4 *yp;
    int64_t xp_rows, yp_rows;
    // H2 arithmetic: amax = device scalar holding max |x| of the A operand's tensor, wscale = the weight image's per-column
    // 1 / sb[n] ([Cop] floats behind the planes).  NULL: the six-product bf16 form.
    const float *amax, *wscale;
    // amax_out != NULL: the epilogue also leaves max |y| of the values it stores in this device scalar (zero, or an earlier maximum,
    // on entry) -- the `amax` of whoever consumes y on the H2 arithmetic, without a pass over y.  (An `amax` that was NOT the
    // tensor's maximum -- a caller's bug -- overflows fp16 in the split, the output holds infinities and amax_out says so: NaN / inf
    // from a finite input maximum is what dense.h2_check() looks for.)
    float *amax_out;
};

// the block's largest stored magnitude -> *out, one atomic per workgroup.
// Bits of non-negative floats order like unsigned integers; NaN sorts above everything and stays.
__device__ __forceinline__ void block_absmax_out(float mx, float *out, float *red, unsigned seen)
{
    unsigned bits = (mx != mx) ? 0x7fc00000u : __float_as_uint(mx);
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o));
    __syncthreads();                                   // `red` aliases the epilogue's LDS tile
    if ((threadIdx.x & 63) == 0) reinterpret_cast<unsigned *>(red)[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned *r = reinterpret_cast<const unsigned *>(red);
        bits = max(max(r[0], r[1]), max(r[2], r[3]));
        // `seen`: what the scalar held when this workgroup looked (early, off the critical path; possibly stale, which costs one
        // atomic more).  Thousands of atomics on ONE address serialise in the L2 -- 16 K of them took 30 us -- so a workgroup
        // that cannot raise the maximum stays away; the atomic itself returns nothing and nobody waits for it.
        if (bits > seen) __hip_atomic_fetch_max(reinterpret_cast<unsigned *>(out), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the same per wavefront, no barrier: for the short reduce kernels, whose workgroups live for one element per thread
__device__ __forceinline__ void wave_absmax_out(float mx, float *out, unsigned seen)
{
    unsigned bits = (mx != mx) ? 0x7fc00000u : __float_as_uint(mx);
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o));
    if ((threadIdx.x & 63) == 0 && bits > seen)
        __hip_atomic_fetch_max(reinterpret_cast<unsigned *>(out), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float absmax4(float mx, float4 v)
{
    // (fmaxf drops NaNs: carry them by hand)
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    const bool nan = (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
    return nan ? __uint_as_float(0x7fc00000u) : ((mx != mx) ? mx : fmaxf(mx, m));
}

// planes of four consecutive channels n .. n + 3 (n % 4 == 0) of row m: 8 bytes into each of the three plane chunks
__device__ __forceinline__ void emit_planes4(uint4 *yp, int64_t rows, int64_t m, int n, float4 v)
{
    unsigned h0, m0, l0, h1, m1, l1;
    split3x2(v.x, v.y, h0, m0, l0);
    split3x2(v.z, v.w, h1, m1, l1);
    char *d = reinterpret_cast<char *>(yp + ((int64_t)(n >> 4) * NCH + ((n >> 3) & 1)) * rows + m) + ((n >> 2) & 1) * 8;
    *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(d + 2 * rows * 16) = make_uint2(m0, m1);
    *reinterpret_cast<uint2 *>(d + 4 * rows * 16) = make_uint2(l0, l1);
}

template <int BM, int KW>
struct Geo {
    static constexpr int RUN = BM + KW - 1;                       // staged rows of a step
    static constexpr int PITCH = ((RUN + 1 + 3) / 8) * 8 + 4;     // rows per chunk array: >= RUN + 1 (zero row), = 4 mod 8
    static constexpr int PASSES = (RUN + 63) / 64;                // 64 rows x 4 float4 per pass of the 256 threads
};

// ---- epilogue (conv_fwd.hip): accumulators through LDS, 16-byte stores along output rows, fused bias / residual / ReLU /
// producer's ReLU mask (+ the bf16 planes of the stored values for a 1x1 consumer, X3Params::yp).  Shared by conv_x3p_kernel and
// conv_x3q_kernel.  D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// element offset of output row m, column n in y (and in residual / mask_src)
__device__ __forceinline__ int64_t x3_out_off(const X3Params &p, int64_t m, int n)
{
    if (p.o_step == 0) return m * p.Co + n;
    const unsigned mm = (unsigned)m, wo = mm % (unsigned)p.Wo, t = mm / (unsigned)p.Wo;
    const unsigned ho = t % (unsigned)p.Ho, b = t / (unsigned)p.Ho;
    return (((int64_t)b * p.oH + p.o_h0 + p.o_step * (int)ho) * p.oW + p.o_w0 + p.o_step * (int)wo) * p.Co + n;
}

template <int WGM, int WGN, int TM, int TN, bool MF16, typename Acc>
__device__ __forceinline__ void x3_epilogue(const X3Params &p, Acc &acc, uint4 *lds, int64_t m0, int n0, bool part, bool region_b,
                                            int split)
{
    constexpr int BN = WGN * TN * 32;
    constexpr int EPI_STRIDE = BN + 4, EPI_ROWS = WGM * 32;
    constexpr int RBLK = MF16 ? 16 : 32;
    constexpr int CB = TN * 32 / RBLK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int lrow = MF16 ? (lane & 15) : frow;
    const int kg = lane >> 4;
    float *le = reinterpret_cast<float *>(lds);
    const bool vec_ok = (p.Co & 3) == 0;
    constexpr int V = BN / 4, RPP = 256 / V;
    float omax = 0.f;                                  // largest stored magnitude (X3Params::amax_out)
    const unsigned oseen = (p.amax_out != nullptr && !part) ? *reinterpret_cast<const volatile unsigned *>(p.amax_out) : 0u;
    // H2: column n of the accumulators carries sa sb[n] (exact powers of two)
    const bool scaled = p.wscale != nullptr;
    float4 cscale = make_float4(1.f, 1.f, 1.f, 1.f);
    if (scaled) {
        const float ia = h2_scale(p.amax).inv;
        const float4 w4 = *reinterpret_cast<const float4 *>(p.wscale + n0 + (tid % V) * 4);      // (n < Cop: the image's padded columns)
        cscale = make_float4(w4.x * ia, w4.y * ia, w4.z * ia, w4.w * ia);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if constexpr (MF16) {       // D of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        le[(wm * 32 + ib * 16 + 4 * kg + r) * EPI_STRIDE + wn * TN * 32 + j * 16 + lrow] = acc[2 * i + ib][j][r];
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    le[row * EPI_STRIDE + (wn * TN + j) * 32 + frow] = acc[i][j][r];
                }
        }
        __syncthreads();
        const int c4 = tid % V;
        const int n = n0 + c4 * 4;
        constexpr int NPASS = EPI_ROWS / RPP;
        if (vec_ok && !part && p.res_H == 0) {
            // The common form (16-byte columns, final values, residual / gradient sum at the output's own resolution): four passes
            // at a time, every global load of the four issued before the first value is used.  The general loop below tests
            // five kernel-uniform flags per pass and hipcc turns each into a branch around ONE load, i.e. 8 dependent memory
            // round trips per thread and row block -- the whole run time of the short-K layers (layer1 conv3: 4 K steps)
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
                    o[u] = x3_out_off(p, ok[u] ? m : 0, n) & -(int64_t)ok[u];   // rows / columns past the end: element 0, read and dropped
                    v[u] = *reinterpret_cast<const float4 *>(le + row * EPI_STRIDE + c4 * 4);
                    if (scaled) { v[u].x *= cscale.x; v[u].y *= cscale.y; v[u].z *= cscale.z; v[u].w *= cscale.w; }
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
                        omax = absmax4(omax, v[u]);
                    }
                if (p.yp != nullptr) {           // bf16 planes of the same values for a 1x1 consumer
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int row = tid / V + (pass0 + u) * RPP;
                        if (ok[u]) emit_planes4(p.yp, p.yp_rows, m0 + ((row >> 5) * TM + i) * 32 + (row & 31), n, v[u]);
                    }
                }
            }
            if (i + 1 < TM) __syncthreads();
            continue;
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = tid / V + pass * RPP;
            const int64_t m = m0 + ((row >> 5) * TM + i) * 32 + (row & 31);
            if (m >= p.M || n >= p.Co) continue;
            float4 v = *reinterpret_cast<const float4 *>(le + row * EPI_STRIDE + c4 * 4);
            if (scaled) { v.x *= cscale.x; v.y *= cscale.y; v.z *= cscale.z; v.w *= cscale.w; }
            const int64_t o = x3_out_off(p, m, n);
            if (part) {
                float *dst = region_b ? p.partial + (p.splits_a > 1 ? (int64_t)p.splits_a * p.m_rem0 * p.Co : 0) +
                                            ((int64_t)split * (p.M - p.m_rem0) + (m - p.m_rem0)) * p.Co + n
                                      : p.partial + ((int64_t)split * p.m_rem0 + m) * p.Co + n;
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
                if (p.mask_src) {
                    const float4 mv = *reinterpret_cast<const float4 *>(p.mask_src + o);
                    v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
                    v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4 *>(p.y + o) = v;
                omax = absmax4(omax, v);
                if (p.yp != nullptr) emit_planes4(p.yp, p.yp_rows, m, n, v);
            } else {
                auto put = [&](int e, float t) {
                    if (n + e >= p.Co) return;
                    t += p.bias ? p.bias[n + e] : 0.f;
                    if (p.residual) t += p.residual[o + e];
                    if (p.relu) t = fmaxf(t, 0.f);
                    if (p.mask_src) t = p.mask_src[o + e] > 0.f ? t : 0.f;
                    p.y[o + e] = t;
                    omax = absmax4(omax, make_float4(t, 0.f, 0.f, 0.f));
                };
                put(0, v.x); put(1, v.y); put(2, v.z); put(3, v.w);
            }
        }
        if (i + 1 < TM) __syncthreads();
    }
    if (p.amax_out != nullptr && !part) block_absmax_out(omax, p.amax_out, le, oseen);
}

// MF16: the products run on v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.  Its 32 k per instruction carry TWO of
// the six plane products at once -- the lane groups k = 0..15 and k = 16..31 read different planes: [a0|a1].[b0|b1] = a0 b0 +
// a1 b1, [a0|a1].[b1|b0] = a0 b1 + a1 b0, [a0|a2].[b2|b0] = a0 b2 + a2 b0 -- so a 16-channel slice takes three instructions of
// 16 cycles per 16x16 block: the same matrix-pipe cycles per FLOP, but the chip holds a higher clock on this shape under its
// power cap (MI355X_MICROARCH.md, "DVFS give-back" item 7).
// NB: LDS buffers of the B tile = prefetch distance + 1.  The LDS-DMA of tap t + NB - 1 is issued at the start of tap t and must
// have landed at the end of tap t + NB - 2; it stays in flight across the barriers in between, which the compiler's
// __syncthreads() would not allow (it drains vmcnt(0) while an LDS-DMA is outstanding): the loop uses raw s_barrier with its
// own counted s_waitcnt, and the activation loads are inline asm so that hipcc has no vector-memory result of its own to wait
// for inside the loop (cdna_hip_programming.md section 5, "Pipelining across barriers").  Small tiles, whose tap is a few
// hundred matrix-pipe cycles, need the distance: with one tap of prefetch every tap waited for its weights (~1 us).
// resident workgroups per CU (= waves per SIMD of a 4-wave workgroup): by registers 3 (128x128) or 4, by the 160 KB of LDS
template <int WGM, int WGN, int TM, int TN, int KW, int NB, bool H2 = false>
constexpr int x3p_occupancy()
{
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NCHL = H2 ? 4 : NCH;               // chunk arrays in LDS: H2 keeps two planes
    constexpr int main_bytes = (2 * NCHL * Geo<BM, KW>::PITCH + NB * NCHL * BN) * 16;
    constexpr int epi_bytes = WGM * 32 * (BN + 4) * 4;
    constexpr int by_lds = 163840 / (main_bytes > epi_bytes ? main_bytes : epi_bytes);
    constexpr int by_regs = TM * TN >= 4 ? (H2 && HTD_X3H_OCC4 ? 4 : 3) : 4;
    return by_lds < by_regs ? by_lds : by_regs;
}

template <int WGM, int WGN, int TM, int TN, int KW, bool MF16, int NB, bool H2 = false>
__global__ __launch_bounds__(256, (x3p_occupancy<WGM, WGN, TM, TN, KW, NB, H2>())) void conv_x3p_kernel(X3Params p)
{
    static_assert(!(H2 && MF16), "H2: 32x32x16 form only");
    constexpr int NCHL = H2 ? 4 : NCH;               // chunk arrays per LDS tile (the weight image in memory always has NCH)
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    using G = Geo<BM, KW>;
    constexpr int PADX = (KW - 1) / 2;
    constexpr int A_VEC = NCHL * G::PITCH;                        // uint4 per A buffer
    constexpr int B_VEC = NCHL * BN;                              // uint4 per B buffer
    constexpr int MAIN_VEC = 2 * A_VEC + NB * B_VEC;
    constexpr int EPI_STRIDE = BN + 4, EPI_ROWS = WGM * 32;
    constexpr int EPI_VEC = EPI_ROWS * EPI_STRIDE / 4;
    constexpr int LDS_VEC = MAIN_VEC > EPI_VEC ? MAIN_VEC : EPI_VEC;
    __shared__ uint4 lds[LDS_VEC];
    uint4 *const lA = lds, *const lB = lds + 2 * A_VEC;
    const unsigned lds_a0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lA;      // LDS byte address of the A buffers

    // XCD-aware tile order (conv_fwd.hip): the blocks of one XCD walk a contiguous run of tiles, N tiles of an M tile adjacent.
    // Blocks past the whole tiles are (remainder tile, K split) pairs in plain order.
    int bid = blockIdx.x, split = 0, sps = 0;
    bool part = false;
    const bool region_b = bid >= p.tiles_a * p.splits_a;
    if (region_b) {
        const int r = bid - p.tiles_a * p.splits_a;
        bid = p.tiles_a + r / p.splits_b;
        split = r % p.splits_b;
        part = p.splits_b > 1;
        sps = p.sps_b;
    } else if (p.splits_a > 1) {
        split = bid % p.splits_a;
        bid = bid / p.splits_a;
        part = true;
        sps = p.sps_a;
    } else {
        const int q = p.tiles_a / 8, r = p.tiles_a % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.nt, tile_n = bid % p.nt;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int PADY = (p.kh - 1) / 2;

    // ---- A staging coordinates: thread (vrow, vcol) handles row vrow + 64 i of the run, channels 4 vcol .. + 3 of the slice
    const int vcol = tid & 3, vrow = tid >> 2;
    // halo mode: row j of the run is input pixel m0 - PADX + j (+ (ky - PADY) * W): one base, rows 64 apart per pass;
    // 1x1 mode: row j is output pixel m0 + j, whose input pixel is decoded once per pass (strided layers)
    const int a_g0 = (int)m0 + vrow - PADX;
    unsigned a_off[KW > 1 ? 1 : G::PASSES];            // element offsets mod 2^32
    bool a_in[KW > 1 ? 1 : G::PASSES];
    unsigned a_tm[KW > 1 ? 1 : G::PASSES];             // tap-list mode: bit t = tap t of this row lies inside the map
    if constexpr (KW > 1) {
        a_off[0] = (unsigned)a_g0 * (unsigned)p.Ci + vcol * 4;
        a_in[0] = true;
        a_tm[0] = 0u;
    } else {
#pragma unroll
        for (int i = 0; i < G::PASSES; ++i) {
            const int64_t m = m0 + vrow + 64 * i;
            a_in[i] = vrow + 64 * i < G::RUN && m < p.M;
            const unsigned mm = a_in[i] ? (unsigned)m : 0u;
            const unsigned wo = mm % (unsigned)p.Wo, t = mm / (unsigned)p.Wo;
            const unsigned ho = t % (unsigned)p.Ho, b = t / (unsigned)p.Ho;
            a_off[i] = ((b * (unsigned)p.Hx + ho * (unsigned)p.stride) * (unsigned)p.Wx + wo * (unsigned)p.stride) *
                           (unsigned)p.Ci + vcol * 4;
            unsigned tm = 0x1ffu;
            if (p.ntl > 0) {
                tm = 0u;
                const int hb = (int)(ho * (unsigned)p.stride), wb = (int)(wo * (unsigned)p.stride);
                for (int q = 0; q < p.ntl; ++q)
                    tm |= ((unsigned)(hb + p.tl_dy[q]) < (unsigned)p.Hx && (unsigned)(wb + p.tl_dx[q]) < (unsigned)p.Wx) ? (1u << q) : 0u;
            }
            a_tm[i] = tm;
        }
    }

    // ---- per-lane tap validity of the MFMA rows (halo mode): bit ky * KW + kx of vmask[i] for row block i
    // (32x32x16: block = 32 rows, lane row = lane & 31, k half = lane >> 5; 16x16x32: 16 rows, lane & 15, k group = lane >> 4)
    constexpr int RBLK = MF16 ? 16 : 32;
    constexpr int RB = TM * 32 / RBLK, CB = TN * 32 / RBLK;          // row / column blocks of the wave tile
    const int lrow = MF16 ? (lane & 15) : frow;
    const int kg = lane >> 4;
    unsigned vmask[RB];
    const int a_frag0 = (wm * TM * 32 + lrow) * 16;   // byte offset of the lane's fragment row in a chunk array: block 0, tap 0
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int r = wm * TM * 32 + i * RBLK + lrow;
        vmask[i] = 0xffffffffu;
        if constexpr (KW > 1) {
            const int64_t m = m0 + r;
            const unsigned mm = m < p.M ? (unsigned)m : 0u;
            const int xx = (int)(mm % (unsigned)p.Wx), yy = (int)((mm / (unsigned)p.Wx) % (unsigned)p.Hx);
            unsigned v = 0u;
            for (int ky = 0; ky < p.kh; ++ky)
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    const bool ok = m < p.M && (unsigned)(yy + ky - PADY) < (unsigned)p.Hx &&
                                    (unsigned)(xx + kx - PADX) < (unsigned)p.Wx;
                    v |= ok ? (1u << (ky * KW + kx)) : 0u;
                }
            vmask[i] = v;
        }
    }
    const int zero_frag = G::RUN * 16;                            // the zero row of a chunk array
    // chunk (= plane * 2 + channel half) the lane reads for each operand:
    //   32x32x16: plane q -> chunk 2 q + (lane >> 5)
    //   16x16x32: operand [p|q] -> plane p for k groups 0, 1 and plane q for k groups 2, 3, channel half = k group & 1
    const int a_c0 = MF16 ? kg * G::PITCH * 16 : fhalf * G::PITCH * 16;                                // [a0|a1] / a0
    const int a_c1 = MF16 ? ((kg >> 1) * 4 + (kg & 1)) * G::PITCH * 16 : 0;                            // [a0|a2]
    const int b_row = (wn * TN * 32 + lrow) * 16;
    const int b_c0 = (MF16 ? kg * BN * 16 : fhalf * BN * 16) + b_row;                                  // [b0|b1] / b0
    const int b_c1 = MF16 ? ((1 - (kg >> 1)) * 2 + (kg & 1)) * BN * 16 + b_row : 0;                    // [b1|b0]
    const int b_c2 = MF16 ? (((kg >> 1) ? 0 : 4) + (kg & 1)) * BN * 16 + b_row : 0;                    // [b2|b0]

    const int total_steps = p.ncs * p.kh;
    const int s_begin = part ? split * sps : 0;
    const int s_end = part ? min(total_steps, s_begin + sps) : total_steps;

    // zero rows of both A buffers (never overwritten: the staging writes rows < RUN only)
    if (tid < 2 * NCHL) lA[(tid / NCHL) * A_VEC + (tid % NCHL) * G::PITCH + G::RUN] = make_uint4(0u, 0u, 0u, 0u);

    // ---- A staging registers.  A pass (64 rows x one float4 per thread) is IN FLIGHT FOR A WHOLE TAP: issued at the start of
    // one tap, waited for and split into LDS at the end of the NEXT one, so the matrix work of two taps (and of the
    // co-resident workgroups) covers the HBM latency; with the wait in the same tap, a 64-row tile -- a few hundred matrix
    // cycles per tap -- stalled ~1 us per tap on it.
    //   KW > 1: tap kx issues pass kx + 1 of the NEXT step's run (the last tap: pass 0 of the run after it) and stores pass kx;
    //   KW = 1: a step is one tap; it issues the NP passes of the run two steps ahead and stores the run of the next step.
    // Two register sets alternate tap by tap: the tap loop is unrolled by two so that the set is a literal.  hipcc does not
    // know that a load into these registers is outstanding, so it must never have a reason to MOVE them: every asm statement
    // that touches them ties them through "+v", none sits under a branch (a pass that does not exist is loaded from element 0
    // and dropped at store time), and the chain init -> load -> landed -> load ... keeps one physical register per slot
    // around the loop.  tools/x3p_check_isa.py (tests/test_build.py) checks the ISA of every instantiation for such moves.
    constexpr int NP = G::PASSES;
    constexpr int NPT = KW > 1 ? 1 : NP;                          // staging registers per set = A loads per tap
    // 128x128 tiles (24 MFMAs per tap): the staged pass is waited for after ONE tap and split between this tap's MFMAs; the
    // smaller tiles keep it in flight for two taps and split behind the MFMAs (their taps are too short to cover the load:
    // -3 % on the 128x64 layers with the early form, +5..9 % on the 128x128 3x3 layers)
    constexpr bool EARLY = HTD_X3P_EARLY && TM * TN >= 4;
    static_assert(KW == 1 || NP <= KW, "one pass per tap");
    f32x4 ra[2 * NPT];
#pragma unroll
    for (int i = 0; i < 2 * NPT; ++i) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned ra_ok = 0u;
    float h2s = 1.f;
    if constexpr (H2) h2s = h2_scale(p.amax).s;
    // (no divisions inside the K loop: the step's channel slice / filter row and the prefetch pointers advance incrementally)
    // (kw == 1: `shift` is the step's tap index -- tap-list mode looks its validity up -- and koff holds the tap's pixel offset)
    auto load_pass = [&](int shift, int koff, int i, int slot, bool live) __attribute__((always_inline)) {   // pass i of the run at pixel shift `shift`
        bool ok;
        unsigned off;
        if constexpr (KW > 1) {
            const int g = a_g0 + 64 * i + shift;
            ok = live && vrow + 64 * i < G::RUN && g >= 0 && (int64_t)g < p.M;
            off = a_off[0] + (unsigned)(64 * i) * (unsigned)p.Ci + (unsigned)koff;
            ra_ok = ok ? (ra_ok | (1u << i)) : (ra_ok & ~(1u << i));
        } else {
            ok = live && a_in[i] && ((a_tm[i] >> shift) & 1u);
            off = a_off[i] + (unsigned)koff;
            ra_ok = ok ? (ra_ok | (1u << slot)) : (ra_ok & ~(1u << slot));
        }
        // out of range / not live: element 0, zeroed or dropped at store time.  Inline asm: hipcc must not count this load (see NB above)
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(ra[slot]) : "v"(p.x + (off & (0u - (unsigned)ok))) : "memory");
    };
    auto store_pass = [&](int buf, int i, int slot) __attribute__((always_inline)) {   // registers -> three bf16 planes, chunk-major
        int j = vrow + 64 * i;
        if constexpr (EARLY) {
            // rows past the run go to the spare row behind the zero row instead of skipping the stores: a predicated region
            // (the branch over them) cannot take this tap's MFMAs
            static_assert(G::PITCH >= G::RUN + 2, "a spare row behind the zero row");
            j = (G::RUN % 64 == 0 || j < G::RUN) ? j : G::RUN + 1;
        } else {
            if (G::RUN % 64 != 0 && j >= G::RUN) return;
        }
        bool ok;
        if constexpr (KW > 1) ok = (ra_ok >> i) & 1u;
        else ok = (ra_ok >> slot) & 1u;
        const float4 v = make_float4(ok ? ra[slot][0] : 0.f, ok ? ra[slot][1] : 0.f, ok ? ra[slot][2] : 0.f, ok ? ra[slot][3] : 0.f);
        // chunk = plane * 2 + (vcol >> 1); 8 bytes at half (vcol & 1) of the row's 16
        const unsigned d = lds_a0 + (unsigned)(buf * A_VEC * 16 + ((vcol >> 1) * G::PITCH + j) * 16 + (vcol & 1) * 8);
        if constexpr (H2) {
            unsigned h0, l0, h1, l1;
            split2hx2(v.x * h2s, v.y * h2s, h0, l0);
            split2hx2(v.z * h2s, v.w * h2s, h1, l1);
            lds_store8<0>(d, h0, h1);
            lds_store8<2 * G::PITCH * 16>(d, l0, l1);
        } else {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3x2(v.x, v.y, h0, m0_, l0);
            split3x2(v.z, v.w, h1, m1, l1);
            lds_store8<0>(d, h0, h1);
            lds_store8<2 * G::PITCH * 16>(d, m0_, m1);
            lds_store8<4 * G::PITCH * 16>(d, l0, l1);
        }
    };
    // B tile of (step s, tap kx) -> LDS buffer `buf`, LDS-DMA: instruction idx = chunk * (BN / 64) + half covers 64 rows
    // (H2: the image's first four chunks -- two planes -- only)
    constexpr int NI = (H2 ? 4 : NCH) * BN / 64;                  // LDS-DMA instructions per B tile, dealt round-robin to the waves
    const uint4 *bsrc[(NI + 3) / 4];                              // the lane's source of the wave's k-th instruction, tile (tap 0, slice 0)
#pragma unroll
    for (int k = 0; k < (NI + 3) / 4; ++k) {
        const int idx = (wave + 4 * k) % NI;
        bsrc[k] = p.wp + (int64_t)(idx / (BN / 64)) * p.Cop + (idx % (BN / 64)) * 64 + n0 + lane;
    }
    const unsigned lds_b0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lB;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // provably wave-uniform (an SGPR) for the M0 operand
    auto load_b = [&](unsigned tile_off, int buf) {                // tile_off: uint4 index of the (tap, slice) tile in the plane image
#pragma unroll
        for (int k = 0; k < (NI + 3) / 4; ++k) {
            const int idx = wave_u + 4 * k;
            if (NI % 4 != 0 && idx >= NI) continue;
            // instruction idx = chunk * (BN / 64) + half covers rows half * 64 .. + 63 of chunk array `chunk`: LDS offset idx * 1 KiB
            lds_dma16(bsrc[k] + tile_off, lds_b0 + (unsigned)((buf * B_VEC + idx * 64) * 16));
        }
    };

    auto lds_barrier = [&]() {             // LDS writes of this wave done, then the workgroup barrier (no vmcnt drain)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    constexpr int DIST = NB - 1;                                  // prefetch distance of the B tiles, in taps
    constexpr int NI_MIN = NI / 4;                                // LDS-DMA instructions every wave issues per B tile
    static_assert(NI_MIN >= 1, "every wave issues at least one LDS-DMA per B tile");
    // At the end of a tap the A pass issued in the PREVIOUS tap and the B tile of the next tap must have landed.  A tap issues
    // its B tile first, then its A pass(es): with DIST = 1 the B tile is this tap's own and only the A loads behind it may stay
    // outstanding; with DIST >= 2 it is older than everything this tap issued.  Vector-memory operations retire in order.
    auto wait_tap = [&](bool b_issued) __attribute__((always_inline)) {
        if (DIST >= 2 && b_issued) wait_vm<NPT + NI_MIN>();
        else wait_vm<NPT>();
    };
    // the next B tile to issue: step, tap of the filter (ky * KW + kx), channel slice, buffer, offset in the plane image
    const int ntap = p.kh * KW;
    const unsigned tile_stride = (unsigned)(NCH * p.Cop);         // uint4 per (tap, slice) tile
    int ps = s_begin, pbuf = 0;
    int pcs = s_begin / p.kh, ptap = (s_begin - pcs * p.kh) * KW;
    const bool tap_list = KW == 1 && p.ntl > 0;
    unsigned poff = (unsigned)((tap_list ? p.tl_w[ptap] : ptap) * p.ncs + pcs) * tile_stride;
    auto issue_b = [&]() -> bool {
        const bool any = ps < s_end;
        if (any) load_b(poff, pbuf);
        pbuf = pbuf + 1 == NB ? 0 : pbuf + 1;
        ++ptap;
        poff += (unsigned)p.ncs * tile_stride;
        if (ptap % KW == 0) ++ps;
        if (ptap == ntap) { ptap = 0; ++pcs; poff = (unsigned)pcs * tile_stride; }
        if constexpr (KW == 1)
            if (tap_list) poff = (unsigned)(p.tl_w[ptap] * p.ncs + pcs) * tile_stride;
        return any;
    };
    // element offset of (tap / filter row t, channel slice c) relative to the run's base pixel
    auto tap_koff = [&](int t, int c) -> int {
        if constexpr (KW == 1)
            if (tap_list) return (p.tl_dy[t] * p.Wx + p.tl_dx[t]) * p.Ci + c * XK;
        return (t - PADY) * p.Wx * p.Ci + c * XK;
    };

    using acc_t = typename std::conditional<MF16, f32x4, f32x16>::type;
    acc_t acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < (MF16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

    // the matrix work of one tap: fragments of A buffer `abuf` at row shift kx against B buffer `bbuf`
    // H2: the fragment reads of a tap are issued FIRST in the tap (phase 0), ahead of the tile / pass bookkeeping and the wait for
    // the staged pass, so that their LDS latency runs under those ~100 scalar and vector instructions; the MFMAs (phase 1) follow
    f16x8 hfa[RB][2], hfb[CB][2];
    auto mma_tap = [&](int abuf, int bbuf, int kx, int tap0, int phase = 2) __attribute__((always_inline)) {
        const char *la = reinterpret_cast<const char *>(lA + abuf * A_VEC);
        const char *lb = reinterpret_cast<const char *>(lB + bbuf * B_VEC);
        int arow[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            arow[i] = a_frag0 + (i * RBLK + kx) * 16;
            if constexpr (KW > 1) arow[i] = ((vmask[i] >> (tap0 + kx)) & 1u) ? arow[i] : zero_frag;
        }
        if constexpr (MF16) {
            bf16x8 fa01[RB], fa02[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                fa01[i] = *reinterpret_cast<const bf16x8 *>(la + arow[i] + a_c0);
                fa02[i] = *reinterpret_cast<const bf16x8 *>(la + arow[i] + a_c1);
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const bf16x8 fb01 = *reinterpret_cast<const bf16x8 *>(lb + b_c0 + j * 16 * 16);
                const bf16x8 fb10 = *reinterpret_cast<const bf16x8 *>(lb + b_c1 + j * 16 * 16);
                const bf16x8 fb20 = *reinterpret_cast<const bf16x8 *>(lb + b_c2 + j * 16 * 16);
#pragma unroll
                for (int i = 0; i < RB; ++i) {      // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa02[i], fb20, acc[i][j], 0, 0, 0);   // a0 b2 + a2 b0
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa01[i], fb10, acc[i][j], 0, 0, 0);   // a0 b1 + a1 b0
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa01[i], fb01, acc[i][j], 0, 0, 0);   // a0 b0 + a1 b1
                }
            }
        } else if constexpr (H2) {
            if (phase != 1) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        hfa[i][q] = *reinterpret_cast<const f16x8 *>(la + arow[i] + a_c0 + q * 2 * G::PITCH * 16);
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        hfb[j][q] = *reinterpret_cast<const f16x8 *>(lb + b_c0 + j * 32 * 16 + q * 2 * BN * 16);
            }
            if (phase != 0) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < CB; ++j) {      // smallest terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hfa[i][1], hfb[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hfa[i][0], hfb[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hfa[i][0], hfb[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            bf16x8 fa[RB][3], fb[CB][3];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    fa[i][q] = *reinterpret_cast<const bf16x8 *>(la + arow[i] + a_c0 + q * 2 * G::PITCH * 16);
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    fb[j][q] = *reinterpret_cast<const bf16x8 *>(lb + b_c0 + j * 32 * 16 + q * 2 * BN * 16);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) {      // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                }
        }
    };

    int s = s_begin;                                              // current step: channel slice cs, filter row ky
    int cs = s_begin / p.kh, ky = s_begin - cs * p.kh;
    int cs1 = cs, ky1 = ky;                                       // step s + 1
    if (++ky1 == p.kh) { ky1 = 0; ++cs1; }
    int cs2 = cs1, ky2 = ky1;                                     // step s + 2
    if (++ky2 == p.kh) { ky2 = 0; ++cs2; }
    // prologue (every split has at least one step): the first B tiles, the whole run of the first step through the staging
    // registers, then the first pass(es) of the next run in flight in set 1 -- the loop's first tap stores from set 1
#pragma unroll
    for (int d = 0; d < DIST; ++d) issue_b();
    {
        const int shift0 = (ky - PADY) * p.Wx, koff0 = tap_koff(ky, cs);
        if constexpr (KW == 1) {
#pragma unroll
            for (int i = 0; i < NP; ++i) load_pass(ky, koff0, i, i, true);
            wait_vm<0>();
#pragma unroll
            for (int i = 0; i < NP; ++i) landed(ra[i]);
#pragma unroll
            for (int i = 0; i < NP; ++i) store_pass(0, i, i);
#pragma unroll
            for (int i = 0; i < NP; ++i) load_pass(ky1, tap_koff(ky1, cs1), i, NP + i, s_begin + 1 < s_end);
        } else {
            load_pass(shift0, koff0, 0, 0, true);
            load_pass(shift0, koff0, 1, 1, true);
            wait_vm<0>();
            landed(ra[0]);
            landed(ra[1]);
            store_pass(0, 0, 0);
            store_pass(0, 1, 1);
            if constexpr (NP > 2) {
                load_pass(shift0, koff0, 2, 0, true);
                wait_vm<0>();
                landed(ra[0]);
                store_pass(0, 2, 0);
            }
            const int sh = (ky1 - PADY) * p.Wx;
            load_pass(sh, sh * p.Ci + cs1 * XK, 0, 1, s_begin + 1 < s_end);
        }
    }
    lds_barrier();

    int bbuf = 0, abuf = 0, kx = 0;
    int n1shift = (ky1 - PADY) * p.Wx, n1koff = tap_koff(ky1, cs1);
    int n2shift = (ky2 - PADY) * p.Wx, n2koff = tap_koff(ky2, cs2);
    // one tap; P (a literal at the call sites) is the register set this tap LOADS into, the other one is stored
    // (TAIL: the split's last tap when their number is odd -- it has nothing to put in flight)
    auto tap = [&](int P, bool TAIL) __attribute__((always_inline)) {
        const bool more = s + 1 < s_end, more2 = s + 2 < s_end;
        if constexpr (H2 && HTD_X3H_FRAG_FIRST) mma_tap(abuf, bbuf, kx, ky * KW, 0);
        const bool issued = issue_b();
        if (!TAIL) {
            if constexpr (KW == 1) {
#pragma unroll
                for (int i = 0; i < NP; ++i) load_pass(ky2, n2koff, i, P * NP + i, more2);
            } else {
                const bool last = kx == KW - 1;
                load_pass(last ? n2shift : n1shift, last ? n2koff : n1koff, last ? 0 : kx + 1, P, last ? more2 : more);
            }
        }
        if constexpr (EARLY) {
            // the other set (issued one tap ago) has landed: only this tap's own B tile and A pass(es) may stay outstanding.  Its
            // split then runs BETWEEN the MFMAs of this tap (group barriers below) instead of behind them -- the vector and the
            // matrix work of one wave overlap (PMC r03: they co-executed in 9 % of the matrix cycles)
            if (TAIL) wait_vm<0>();
            else if (issued) wait_vm<NPT + NI_MIN>();
            else wait_vm<NPT>();
#pragma unroll
            for (int i = 0; i < NPT; ++i) landed(ra[(P ^ 1) * NPT + i]);
            mma_tap(abuf, bbuf, kx, ky * KW, (H2 && HTD_X3H_FRAG_FIRST) ? 1 : 2);
            // (not under `if (more)`: a predicated region cannot take MFMAs; in the last step the stores put unused rows into the idle buffer)
            if constexpr (KW == 1) {
#pragma unroll
                for (int i = 0; i < NP; ++i) store_pass(abuf ^ 1, i, (P ^ 1) * NP + i);
            } else {
                store_pass(abuf ^ 1, kx, P ^ 1);
            }
            constexpr int NM = RB * CB * ((MF16 || H2) ? 3 : 6);      // MFMAs of the tap
            constexpr int PER = (NPT * 36 + NM - 1) / NM;             // ~36 vector instructions per staged pass
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
            for (int e = 0; e < NM; ++e) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
            }
            // the next tap's B tile has landed; this tap's A passes stay in flight, and with two taps of prefetch (DIST >= 2) so
            // does the tile this tap issued (it is younger than the one needed next: vector-memory operations retire in order)
            if (!TAIL) {
                if (DIST >= 2 && issued) wait_vm<NPT + NI_MIN>();
                else wait_vm<NPT>();
            }
        } else {
            mma_tap(abuf, bbuf, kx, ky * KW, (H2 && HTD_X3H_FRAG_FIRST) ? 1 : 2);
            // the other set (issued one tap ago) and the next tap's B tile have landed
            if (TAIL) wait_vm<0>();
            else wait_tap(issued);
#pragma unroll
            for (int i = 0; i < NPT; ++i) landed(ra[(P ^ 1) * NPT + i]);
            if (more) {
                if constexpr (KW == 1) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) store_pass(abuf ^ 1, i, (P ^ 1) * NP + i);
                } else {
                    store_pass(abuf ^ 1, kx, P ^ 1);
                }
            }
        }
        lds_barrier();               // every wave is done with this tap's buffers; the next tap's are complete
        bbuf = bbuf + 1 == NB ? 0 : bbuf + 1;
        if (++kx == KW) {            // next step (no divisions: slice / filter row advance incrementally)
            kx = 0;
            ++s;
            abuf ^= 1;
            cs = cs1; ky = ky1;
            cs1 = cs2; ky1 = ky2;
            if (++ky2 == p.kh) { ky2 = 0; ++cs2; }
            n1shift = n2shift; n1koff = n2koff;
            n2shift = (ky2 - PADY) * p.Wx;
            n2koff = tap_koff(ky2, cs2);
        }
    };
    const int taps_total = (s_end - s_begin) * KW;
    int t = 0;
#pragma unroll 1
    for (; t + 1 < taps_total; t += 2) {
        tap(0, false);
        tap(1, false);
    }
    if (t < taps_total) tap(0, true);
    // the last taps' loads fetch nothing that is used, but they still WRITE their registers when they land: drain them before
    // the epilogue may reuse the registers
    wait_vm<0>();
#pragma unroll
    for (int i = 0; i < 2 * NPT; ++i) landed(ra[i]);

    x3_epilogue<WGM, WGN, TM, TN, MF16>(p, acc, lds, m0, n0, part, region_b, split);
}

// ---- conv_x3q_kernel (round 4): the 1x1 / stride-1 form with BOTH operands pre-split ------------------------------------
// conv_x3p_kernel splits its A slices inside the K loop: global -> registers -> three bf16 planes -> ds_write, once per (M tile,
// N tile) -- an activation of layer3's conv3 (256 -> 1024) is split Co / BN = 8 times, and the loop carries 3.4 vector + 2.3
// scalar instructions per MFMA and the split's ds_writes next to the fragment reads (matrix pipe 39 % busy on
// that layer).  Here the producer of the activation map has already written its planes (X3Params::yp of ITS epilogue, or
// act_planes_kernel): [Ci/16][6][rows] x 16 B, the weight image's layout with pixels for output channels.  A K step then
// is 6 * (BM + BN) / 64 LDS-DMA instructions dealt round-robin to the four waves, one barrier, the fragment reads and the
// MFMAs: no vector arithmetic, no staging register, no ds_write in the loop.  NS stages of (A, B) tiles in LDS; the tiles of
// step s + NS - 1 are issued at the start of step s.  Work plan, tile order, fragment addressing, MFMA order and epilogue are
// conv_x3p_kernel's (the products and their summation order are the same: the results are bit-identical).
template <int WGM, int WGN, int TM, int TN, int NS>
constexpr int x3q_occupancy()
{
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int main_bytes = NS * NCH * (BM + BN) * 16;
    constexpr int epi_bytes = WGM * 32 * (BN + 4) * 4;
    constexpr int by_lds = 163840 / (main_bytes > epi_bytes ? main_bytes : epi_bytes);
    constexpr int by_regs = TM * TN >= 4 ? 3 : 4;
    return by_lds < by_regs ? by_lds : by_regs;
}

template <int WGM, int WGN, int TM, int TN, bool MF16, int NS>
__global__ __launch_bounds__(256, (x3q_occupancy<WGM, WGN, TM, TN, NS>())) void conv_x3q_kernel(X3Params p)
{
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int A_VEC = NCH * BM, B_VEC = NCH * BN;             // uint4 per A / B tile: [chunk][row]
    constexpr int MAIN_VEC = NS * (A_VEC + B_VEC);
    constexpr int EPI_VEC = WGM * 32 * (BN + 4) / 4;
    constexpr int LDS_VEC = MAIN_VEC > EPI_VEC ? MAIN_VEC : EPI_VEC;
    __shared__ uint4 lds[LDS_VEC];
    uint4 *const lA = lds, *const lB = lds + NS * A_VEC;
    const unsigned lds_a0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lA;
    const unsigned lds_b0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lB;

    // tile / K-range of this workgroup: conv_x3p_kernel's decomposition (plan_x3p)
    int bid = blockIdx.x, split = 0, sps = 0;
    bool part = false;
    const bool region_b = bid >= p.tiles_a * p.splits_a;
    if (region_b) {
        const int r = bid - p.tiles_a * p.splits_a;
        bid = p.tiles_a + r / p.splits_b;
        split = r % p.splits_b;
        part = p.splits_b > 1;
        sps = p.sps_b;
    } else if (p.splits_a > 1) {
        split = bid % p.splits_a;
        bid = bid / p.splits_a;
        part = true;
        sps = p.sps_a;
    } else {
        const int q = p.tiles_a / 8, r = p.tiles_a % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.nt, tile_n = bid % p.nt;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int frow = lane & 31, fhalf = lane >> 5;
    constexpr int RBLK = MF16 ? 16 : 32;
    constexpr int RB = TM * 32 / RBLK, CB = TN * 32 / RBLK;
    const int lrow = MF16 ? (lane & 15) : frow;
    const int kg = lane >> 4;
    // fragment addressing as in conv_x3p_kernel, with BM rows per A chunk array
    const int a_row = (wm * TM * 32 + lrow) * 16;
    const int a_c0 = (MF16 ? kg * BM * 16 : fhalf * BM * 16) + a_row;                                  // [a0|a1] / a0
    const int a_c1 = MF16 ? ((kg >> 1) * 4 + (kg & 1)) * BM * 16 + a_row : 0;                          // [a0|a2]
    const int b_row = (wn * TN * 32 + lrow) * 16;
    const int b_c0 = (MF16 ? kg * BN * 16 : fhalf * BN * 16) + b_row;                                  // [b0|b1] / b0
    const int b_c1 = MF16 ? ((1 - (kg >> 1)) * 2 + (kg & 1)) * BN * 16 + b_row : 0;                    // [b1|b0]
    const int b_c2 = MF16 ? (((kg >> 1) ? 0 : 4) + (kg & 1)) * BN * 16 + b_row : 0;                    // [b2|b0]

    const int total_steps = p.ncs;                                // kh = kw = 1
    const int s_begin = part ? split * sps : 0;
    const int s_end = part ? min(total_steps, s_begin + sps) : total_steps;

    // LDS-DMA instructions of one stage: A tile first (NIA), then the B tile (NIB); instruction idx covers 64 rows of one chunk
    // array = 1 KiB of LDS; wave w issues idx = w, w + 4, ...
    constexpr int NIA = NCH * BM / 64, NIB = NCH * BN / 64, NI = NIA + NIB, NIW = (NI + 3) / 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const uint4 *src[NIW];                                        // the lane's source of the wave's k-th instruction at slice 0
    unsigned dst[NIW], adv[NIW];                                  // LDS byte offset inside stage 0; source advance per slice
#pragma unroll
    for (int k = 0; k < NIW; ++k) {
        const int idx = wave_u + 4 * k;
        if (idx < NIA) {
            const int chunk = idx / (BM / 64), half = idx % (BM / 64);
            src[k] = p.xp + (int64_t)chunk * p.xp_rows + m0 + half * 64 + lane;
            dst[k] = lds_a0 + (unsigned)((chunk * BM + half * 64) * 16);
            adv[k] = (unsigned)(NCH * p.xp_rows);
        } else {
            const int j = (idx < NI ? idx : NIA) - NIA, chunk = j / (BN / 64), half = j % (BN / 64);
            src[k] = p.wp + (int64_t)chunk * p.Cop + n0 + half * 64 + lane;
            dst[k] = lds_b0 + (unsigned)((chunk * BN + half * 64) * 16);
            adv[k] = (unsigned)(NCH * p.Cop);
        }
    }
    auto issue = [&](int s, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NIW; ++k) {
            const int idx = wave_u + 4 * k;
            if (NI % 4 != 0 && idx >= NI) continue;
            const unsigned st = (unsigned)stage * (unsigned)((idx < NIA ? A_VEC : B_VEC) * 16);
            lds_dma16(src[k] + (int64_t)s * adv[k], dst[k] + st);
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    using acc_t = typename std::conditional<MF16, f32x4, f32x16>::type;
    acc_t acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < (MF16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

    auto mma = [&](int stage) __attribute__((always_inline)) {
        const char *la = reinterpret_cast<const char *>(lA + stage * A_VEC);
        const char *lb = reinterpret_cast<const char *>(lB + stage * B_VEC);
        if constexpr (MF16) {
            bf16x8 fa01[RB], fa02[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                fa01[i] = *reinterpret_cast<const bf16x8 *>(la + a_c0 + i * 16 * 16);
                fa02[i] = *reinterpret_cast<const bf16x8 *>(la + a_c1 + i * 16 * 16);
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const bf16x8 fb01 = *reinterpret_cast<const bf16x8 *>(lb + b_c0 + j * 16 * 16);
                const bf16x8 fb10 = *reinterpret_cast<const bf16x8 *>(lb + b_c1 + j * 16 * 16);
                const bf16x8 fb20 = *reinterpret_cast<const bf16x8 *>(lb + b_c2 + j * 16 * 16);
#pragma unroll
                for (int i = 0; i < RB; ++i) {      // smallest terms first (conv_x3p_kernel's order)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa02[i], fb20, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa01[i], fb10, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa01[i], fb01, acc[i][j], 0, 0, 0);
                }
            }
        } else {
            bf16x8 fa[RB][3], fb[CB][3];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) fa[i][q] = *reinterpret_cast<const bf16x8 *>(la + a_c0 + i * 32 * 16 + q * 2 * BM * 16);
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) fb[j][q] = *reinterpret_cast<const bf16x8 *>(lb + b_c0 + j * 32 * 16 + q * 2 * BN * 16);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                }
        }
    };

    // The tiles of step s + NS - 1 are issued at the start of step s into the stage that step s - 1 read (the barrier at the end
    // of s - 1 ordered those reads first) and must have landed at the end of step s + NS - 2.  Vector-memory operations retire
    // in order, so at the end of step s all but this wave's youngest NS - 2 stages must be done: NI / 4 is the least any wave
    // issues per stage (a wave with one instruction more waits for part of a younger stage too); in the tail, where fewer
    // stages are outstanding, everything is waited for.  ONE barrier per step.
    constexpr int KEEP = (NS - 2) * (NI / 4);
#pragma unroll
    for (int d = 0; d < NS - 1; ++d)
        if (s_begin + d < s_end) issue(s_begin + d, d);
    if (s_begin + NS - 2 < s_end) wait_vm<KEEP>();
    else wait_vm<0>();
    lds_barrier();
    int stage = 0, pstage = NS - 1;
#pragma unroll 1
    for (int s = s_begin; s < s_end; ++s) {
        const bool more = s + NS - 1 < s_end;
        if (more) issue(s + NS - 1, pstage);
        mma(stage);
        if (more) wait_vm<KEEP>();
        else wait_vm<0>();
        lds_barrier();               // every wave is done with this step's stage; the next step's is complete
        stage = stage + 1 == NS ? 0 : stage + 1;
        pstage = pstage + 1 == NS ? 0 : pstage + 1;
    }
    x3_epilogue<WGM, WGN, TM, TN, MF16>(p, acc, lds, m0, n0, part, region_b, split);
}

// sums the K-range partials of the regions that have them, in range order, and applies the epilogue
__global__ __launch_bounds__(256) void conv_x3p_splitk_epilogue_kernel(X3Params p)
{
    const int64_t lo = p.splits_a > 1 ? 0 : p.m_rem0 * p.Co, hi = p.splits_b > 1 ? p.M * p.Co : p.m_rem0 * p.Co;
    const int64_t na = p.m_rem0 * p.Co, nb = (p.M - p.m_rem0) * p.Co;
    const float *pb = p.partial + (p.splits_a > 1 ? (int64_t)p.splits_a * na : 0);
    float omax = 0.f;
    const unsigned oseen = p.amax_out != nullptr ? *reinterpret_cast<const volatile unsigned *>(p.amax_out) : 0u;
    for (int64_t o = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < hi; o += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (o < na) {
            for (int k = 0; k < p.splits_a; ++k) v += p.partial[(int64_t)k * na + o];
        } else {
            for (int k = 0; k < p.splits_b; ++k) v += pb[(int64_t)k * nb + (o - na)];
        }
        const int n = (int)(o % p.Co);
        if (p.bias) v += p.bias[n];
        if (p.residual) {
            int64_t ro = p.o_step == 0 ? o : x3_out_off(p, o / p.Co, n);
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
        const int64_t oo = p.o_step == 0 ? o : x3_out_off(p, o / p.Co, n);
        if (p.mask_src) v = p.mask_src[oo] > 0.f ? v : 0.f;
        p.y[oo] = v;
        omax = absmax4(omax, make_float4(v, 0.f, 0.f, 0.f));
    }
    __shared__ float ored[4];
    if (p.amax_out != nullptr) block_absmax_out(omax, p.amax_out, ored, oseen);
}

// the same for Co % 4 == 0, four channels per thread: float4 loads of the partials (independent across the ranges), 32-bit
// index arithmetic, one division per four outputs -- these launches are a few MB each and were latency, not bandwidth
__global__ __launch_bounds__(256) void conv_x3p_splitk_epilogue_vec_kernel(X3Params p)
{
    const int C4 = p.Co >> 2;
    const int64_t lo = p.splits_a > 1 ? 0 : p.m_rem0 * C4, hi = p.splits_b > 1 ? p.M * C4 : p.m_rem0 * C4;
    const int64_t na = p.m_rem0 * C4, nb = (p.M - p.m_rem0) * C4;
    const float4 *pa = reinterpret_cast<const float4 *>(p.partial);
    const float4 *pb = pa + (p.splits_a > 1 ? (int64_t)p.splits_a * na : 0);
    float omax = 0.f;
    const unsigned oseen = p.amax_out != nullptr ? *reinterpret_cast<const volatile unsigned *>(p.amax_out) : 0u;
    for (int64_t q = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < hi; q += (int64_t)gridDim.x * blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool in_a = q < na;
        const float4 *src = in_a ? pa + q : pb + (q - na);
        const int64_t stride = in_a ? na : nb;
        const int splits = in_a ? p.splits_a : p.splits_b;
#pragma unroll 4
        for (int k = 0; k < splits; ++k) {
            const float4 t = src[(int64_t)k * stride];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const unsigned m = (unsigned)(q / C4), n = (unsigned)(q - (int64_t)m * C4) * 4u;
        const int64_t o = x3_out_off(p, (int64_t)m, (int)n);
        if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        if (p.residual) {
            int64_t ro = o;
            if (p.res_H > 0) {
                const unsigned wo = m % (unsigned)p.Wo, t = m / (unsigned)p.Wo;
                const unsigned ho = t % (unsigned)p.Ho, b = t / (unsigned)p.Ho;
                const int rh = min((int)floorf(ho * p.res_sh), p.res_H - 1);
                const int rw = min((int)floorf(wo * p.res_sw), p.res_W - 1);
                ro = (((int64_t)b * p.res_H + rh) * p.res_W + rw) * p.Co + n;
            }
            const float4 rv = *reinterpret_cast<const float4 *>(p.residual + ro);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (p.mask_src) {
            const float4 mv = *reinterpret_cast<const float4 *>(p.mask_src + o);
            v.x = mv.x > 0.f ? v.x : 0.f; v.y = mv.y > 0.f ? v.y : 0.f;
            v.z = mv.z > 0.f ? v.z : 0.f; v.w = mv.w > 0.f ? v.w : 0.f;
        }
        *reinterpret_cast<float4 *>(p.y + o) = v;
        omax = absmax4(omax, v);
        if (p.yp != nullptr) emit_planes4(p.yp, p.yp_rows, (int64_t)m, (int)n, v);
    }
    __shared__ float ored[4];
    if (p.amax_out != nullptr) block_absmax_out(omax, p.amax_out, ored, oseen);
}

// ---- weight planes --------------------------------------------------------------------------------------------------
// w [Co][taps][Ci] fp32 (KRSC) -> planes [taps][K/16][6][Np] x 16 B, chunk c = plane * 2 + half holds the bf16 piece `plane` of
// the 8 reduction channels 16 cs + 8 half .. + 7 of output channel n (rows n >= N are zeros).
//   transposed = 0 (forward operand):       N = Co, K = Ci, tap t      <- w[n][t][k]
//   transposed = 1 (data-gradient operand): N = Ci, K = Co, tap t      <- w[k][taps - 1 - t][n]   (flipped, transposed filter)
// One thread per (tap, slice, n): 16 loads, 6 x 16-byte stores that are contiguous across the threads of a wavefront.
__device__ __forceinline__ void x3_planes_element(const float *__restrict__ w, uint4 *__restrict__ out, int Co, int taps,
                                                  int Ci, int Np, int transposed, int64_t e)
{
    const int N = transposed ? Ci : Co, K = transposed ? Co : Ci;
    const int ncs = K / XK;
    const int n = (int)(e % Np);
    const int64_t t2 = e / Np;
    const int cs = (int)(t2 % ncs), tap = (int)(t2 / ncs);
    float v[XK];
    if (n < N) {
        if (!transposed) {
            const float4 *src = reinterpret_cast<const float4 *>(w + ((int64_t)n * taps + tap) * Ci + cs * XK);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 f = src[q];
                v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < XK; ++k) v[k] = w[((int64_t)(cs * XK + k) * taps + (taps - 1 - tap)) * Ci + n];
        }
    } else {
#pragma unroll
        for (int k = 0; k < XK; ++k) v[k] = 0.f;
    }
    unsigned pl[3][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split3x2(v[2 * k], v[2 * k + 1], pl[0][k], pl[1][k], pl[2][k]);
    uint4 *dst = out + ((int64_t)(tap * ncs + cs) * NCH) * Np + n;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            dst[(int64_t)(q * 2 + h) * Np] = make_uint4(pl[q][4 * h], pl[q][4 * h + 1], pl[q][4 * h + 2], pl[q][4 * h + 3]);
}

__global__ __launch_bounds__(256) void x3_planes_kernel(const float *__restrict__ w, uint4 *__restrict__ out, int Co, int taps,
                                                        int Ci, int Np, int transposed)
{
    const int K = transposed ? Co : Ci;
    const int64_t total = (int64_t)taps * (K / XK) * Np;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
        x3_planes_element(w, out, Co, taps, Ci, Np, transposed, e);
}

// one launch for the plane images of many weights (every plain convolution / Linear of neck and heads at the start of a
// step, every folded weight of a backbone stage after its BN fold): workgroup b serves entry l with block0[l] <= b < block0[l + 1]
struct PlanesDesc {
    const float *w;
    uint4 *out;
    int Co, taps, Ci, transposed;
    int64_t block0;
};

__global__ __launch_bounds__(256) void x3_planes_many_kernel(const PlanesDesc *__restrict__ descs, int n)
{
    int lo = 0, hi = n - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {                    // last entry whose first block is <= b
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const PlanesDesc d = descs[lo];
    const int N = d.transposed ? d.Ci : d.Co, K = d.transposed ? d.Co : d.Ci;
    const int Np = ((N + 127) / 128) * 128;
    const int64_t total = (int64_t)d.taps * (K / XK) * Np;
    const int64_t e = (b - d.block0) * 256 + threadIdx.x;
    if (e < total) x3_planes_element(d.w, d.out, d.Co, d.taps, d.Ci, Np, d.transposed, e);
}

inline int planes_np(int N) { return (int)htd::ceil_div(N, 128) * 128; }

// ---- H2 weight image: the same [taps][K/16][6][Np] x 16 B layout with the two fp16 pieces of sb[n] w in chunks 0..3 (chunks 4, 5
// unused) and, behind the planes, Np floats 1 / sb[n] (what the epilogue multiplies by), Np floats sb[n] and the chunk maxima.
// Pass 1: sb[n] = the power of two that puts row n's largest magnitude into [2^14, 2^15).
// Pass 1, one workgroup per (64 output rows, chunk of H2_ECH reduction elements): the largest magnitude of every row within the
// chunk -> part[chunk][Np] (behind the scales).  transposed (rows are strided columns of w): lane = row, the four waves share
// the chunk, every load is 64 consecutive floats; forward operand (a row is contiguous): a wave takes 16 rows one after the
// other, its lanes stride along the row.  Pass 2 is the planes kernel, which folds the chunks of its row.
constexpr int H2_ECH = 1024;
__host__ __device__ inline int h2_chunks(int taps, int K) { return (taps * K + H2_ECH - 1) / H2_ECH; }

__device__ __forceinline__ void x3h_rowmax_part(const float *__restrict__ w, float *__restrict__ part, int Co, int taps, int Ci, int Np,
                                                int transposed, int n0, int chunk, float *red /* [4][64] */)
{
    const int N = transposed ? Ci : Co, K = transposed ? Co : Ci;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e0 = chunk * H2_ECH, e1 = min(taps * K, e0 + H2_ECH);
    if (transposed) {
        const int n = n0 + lane;
        float mx[4] = {0.f, 0.f, 0.f, 0.f};
        if (n < N) {
            int e = e0 + wave;
            for (; e + 12 < e1; e += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) mx[u] = fmaxf(mx[u], fabsf(w[(int64_t)(e + 4 * u) * Ci + n]));
            }
            for (; e < e1; e += 4) mx[0] = fmaxf(mx[0], fabsf(w[(int64_t)e * Ci + n]));
        }
        red[wave * 64 + lane] = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
    } else {
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wave * 16 + r;
            float mx = 0.f;
            if (n < N) {            // (rows and chunks are whole float4s: Ci % 4 == 0, chunk bounds multiples of 4)
                const float4 *row = reinterpret_cast<const float4 *>(w + (int64_t)n * taps * Ci);
                for (int e = (e0 >> 2) + lane; e < (e1 >> 2); e += 64) {
                    const float4 v = row[e];
                    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            }
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) red[wave * 16 + r] = mx;
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const float mx = transposed ? fmaxf(fmaxf(red[lane], red[64 + lane]), fmaxf(red[128 + lane], red[192 + lane])) : red[lane];
        part[(int64_t)chunk * Np + n0 + lane] = mx;
    }
}

// the row's scale from its chunk maxima (pass 2, by whoever needs it)
__device__ __forceinline__ H2Scale x3h_row_scale(const float *__restrict__ part, int nch, int Np, int n)
{
    float mx = 0.f;
    for (int c = 0; c < nch; ++c) mx = fmaxf(mx, part[(int64_t)c * Np + n]);
    return h2_scale(&mx);
}

__device__ __forceinline__ void x3h_planes_element(const float *__restrict__ w, uint4 *__restrict__ out, float *__restrict__ scales,
                                                   int Co, int taps, int Ci, int Np, int transposed, int64_t e)
{
    const int N = transposed ? Ci : Co, K = transposed ? Co : Ci;
    const int ncs = K / XK;
    const int n = (int)(e % Np);
    const int64_t t2 = e / Np;
    const int cs = (int)(t2 % ncs), tap = (int)(t2 / ncs);
    float v[XK];
    const H2Scale rs = x3h_row_scale(scales + 2 * Np, h2_chunks(taps, K), Np, n);
    if (tap == 0 && cs == 0) {
        scales[n] = rs.inv;
        scales[Np + n] = rs.s;
    }
    if (n < N) {
        const float sb = rs.s;
        if (!transposed) {
            const float4 *src = reinterpret_cast<const float4 *>(w + ((int64_t)n * taps + tap) * Ci + cs * XK);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 f = src[q];
                v[4 * q] = f.x * sb; v[4 * q + 1] = f.y * sb; v[4 * q + 2] = f.z * sb; v[4 * q + 3] = f.w * sb;
            }
        } else {
#pragma unroll
            for (int k = 0; k < XK; ++k) v[k] = w[((int64_t)(cs * XK + k) * taps + (taps - 1 - tap)) * Ci + n] * sb;
        }
    } else {
#pragma unroll
        for (int k = 0; k < XK; ++k) v[k] = 0.f;
    }
    unsigned pl[2][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split2hx2(v[2 * k], v[2 * k + 1], pl[0][k], pl[1][k]);
    uint4 *dst = out + ((int64_t)(tap * ncs + cs) * NCH) * Np + n;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            dst[(int64_t)(q * 2 + h) * Np] = make_uint4(pl[q][4 * h], pl[q][4 * h + 1], pl[q][4 * h + 2], pl[q][4 * h + 3]);
}

__host__ __device__ inline int64_t planes_vec(int taps, int K, int Np) { return (int64_t)taps * (K / XK) * NCH * Np; }      // uint4 of the planes

__global__ __launch_bounds__(256) void x3h_rowscale_kernel(const float *__restrict__ w, float *__restrict__ scales, int Co, int taps,
                                                           int Ci, int Np, int transposed)
{
    __shared__ float red[256];
    const int groups = Np / 64;
    x3h_rowmax_part(w, scales + 2 * Np, Co, taps, Ci, Np, transposed, (blockIdx.x % groups) * 64, blockIdx.x / groups, red);
}

__global__ __launch_bounds__(256) void x3h_planes_kernel(const float *__restrict__ w, uint4 *__restrict__ out,
                                                         float *__restrict__ scales, int Co, int taps, int Ci, int Np,
                                                         int transposed)
{
    const int K = transposed ? Co : Ci;
    const int64_t total = (int64_t)taps * (K / XK) * Np;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
        x3h_planes_element(w, out, scales, Co, taps, Ci, Np, transposed, e);
}

// many images, two launches (row scales, then planes): PlanesDesc with `transposed` bit 1 set marks an H2 entry -- the bf16 and
// the H2 images of a step are made by the same two tables; row0 = prefix sum of Np over the H2 entries
struct PlanesDescH {
    const float *w;
    uint4 *out;
    int Co, taps, Ci, transposed;
    int64_t block0, row0;
};

__global__ __launch_bounds__(256) void x3h_rowscale_many_kernel(const PlanesDescH *__restrict__ descs, int n)
{
    __shared__ float red[256];
    int lo = 0, hi = n - 1;
    const int64_t b = blockIdx.x;                        // row0: prefix sum of (Np / 64) * chunks over the entries
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].row0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const PlanesDescH d = descs[lo];
    const int tr = d.transposed & 1;
    const int N = tr ? d.Ci : d.Co, K = tr ? d.Co : d.Ci;
    const int Np = ((N + 127) / 128) * 128;
    float *scales = reinterpret_cast<float *>(d.out + planes_vec(d.taps, K, Np));
    const int groups = Np / 64, r = (int)(b - d.row0);
    x3h_rowmax_part(d.w, scales + 2 * Np, d.Co, d.taps, d.Ci, Np, tr, (r % groups) * 64, r / groups, red);
}

__global__ __launch_bounds__(256) void x3h_planes_many_kernel(const PlanesDescH *__restrict__ descs, int n)
{
    int lo = 0, hi = n - 1;
    const int64_t b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const PlanesDescH d = descs[lo];
    const int tr = d.transposed & 1;
    const int N = tr ? d.Ci : d.Co, K = tr ? d.Co : d.Ci;
    const int Np = ((N + 127) / 128) * 128;
    const int64_t total = (int64_t)d.taps * (K / XK) * Np;
    const int64_t e = (b - d.block0) * 256 + threadIdx.x;
    float *scales = reinterpret_cast<float *>(d.out + planes_vec(d.taps, K, Np));
    if (e < total) x3h_planes_element(d.w, d.out, scales, d.Co, d.taps, d.Ci, Np, tr, e);
}

// max |x| over a tensor into a device scalar that is ZERO (or an earlier maximum) on entry: one atomic per wavefront on the
// bits of the non-negative value (they order like unsigned integers; a NaN sorts above everything and stays)
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out)
{
    const int64_t n4 = n >> 2;
    float mx = 0.f;
    unsigned nan = 0u;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {            // four loads in flight per thread
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        const float m0 = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
        const float m1 = fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)));
        const float m2 = fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)));
        const float m3 = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
        mx = fmaxf(fmaxf(mx, fmaxf(m0, m1)), fmaxf(m2, m3));
        // (fmaxf drops NaNs.  x - x is NaN exactly when x is NaN or infinite: sum the lot, test once, look closer only then)
        const float t = (a.x - a.x) + (a.y - a.y) + (a.z - a.z) + (a.w - a.w) + (b.x - b.x) + (b.y - b.y) + (b.z - b.z) + (b.w - b.w) +
                        (c.x - c.x) + (c.y - c.y) + (c.z - c.z) + (c.w - c.w) + (d.x - d.x) + (d.y - d.y) + (d.z - d.z) + (d.w - d.w);
        if (t != t) {
            nan |= (a.x != a.x) | (a.y != a.y) | (a.z != a.z) | (a.w != a.w) | (b.x != b.x) | (b.y != b.y) | (b.z != b.z) | (b.w != b.w) |
                   (c.x != c.x) | (c.y != c.y) | (c.z != c.z) | (c.w != c.w) | (d.x != d.x) | (d.y != d.y) | (d.z != d.z) | (d.w != d.w);
        }
    }
    for (; i < n4; i += stride) {
        const float4 v = x4[i];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float t = x[(n4 << 2) + threadIdx.x];
        mx = fmaxf(mx, fabsf(t));
        nan |= t != t;
    }
    unsigned bits = nan ? 0x7fc00000u : __float_as_uint(mx);
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o));
    if ((threadIdx.x & 63) == 0 && bits != 0u) atomicMax(out, bits);
}

// fp32 map x [M][C] (NHWC pixels x channels, C % 16 == 0) -> activation planes [C/16][6][rows] x 16 B (X3Params::xp): what a
// producer's epilogue writes through X3Params::yp, as a pass of its own for maps whose producer is not one of these kernels.
// One thread per (pixel, 8 channels): two float4 in, three 16-byte chunks out, consecutive lanes = consecutive pixels.
__global__ __launch_bounds__(256) void act_planes_kernel(const float *__restrict__ x, uint4 *__restrict__ out, int64_t M, int C,
                                                         int64_t rows)
{
    const int c8n = C >> 3;
    const int64_t total = M * c8n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t m = e % M;
        const int c8 = (int)(e / M);
        const float4 a = *reinterpret_cast<const float4 *>(x + m * C + c8 * 8);
        const float4 b = *reinterpret_cast<const float4 *>(x + m * C + c8 * 8 + 4);
        unsigned h[4], md[4], l[4];
        split3x2(a.x, a.y, h[0], md[0], l[0]);
        split3x2(a.z, a.w, h[1], md[1], l[1]);
        split3x2(b.x, b.y, h[2], md[2], l[2]);
        split3x2(b.z, b.w, h[3], md[3], l[3]);
        uint4 *d = out + ((int64_t)(c8 >> 1) * NCH + (c8 & 1)) * rows + m;
        d[0] = make_uint4(h[0], h[1], h[2], h[3]);
        d[2 * rows] = make_uint4(md[0], md[1], md[2], md[3]);
        d[4 * rows] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

inline int64_t act_rows(int64_t M) { return htd::ceil_div(M, (int64_t)128) * 128; }

// Tile configurations (BM x BN): 0 64x64, 1 128x128, 2 128x64, 3 64x128 -- all 2x2 waves
struct XCfg { int bm, bn; };
constexpr XCfg kXCfg[] = {{64, 64}, {128, 128}, {128, 64}, {64, 128}};

// HTD_X3P=0: every layer keeps conv_igemm_kernel.  HTD_X3P_TUNE=1 (tools/sweep_x3p.py, A/B runs inside one process) makes the
// library re-read HTD_X3P and HTD_X3P_FORCE_TILE on every call; otherwise nothing is read after load.
static const bool g_x3p_tune = getenv("HTD_X3P_TUNE") != nullptr;
static const bool g_x3p_off = getenv("HTD_X3P") != nullptr && atoi(getenv("HTD_X3P")) == 0;
bool x3p_off()
{
    if (!g_x3p_tune) return g_x3p_off;
    const char *e = getenv("HTD_X3P");
    return e && atoi(e) == 0;
}
int forced_cfg()
{
    if (!g_x3p_tune) return -1;
    const char *e = getenv("HTD_X3P_FORCE_TILE");
    const int v = e ? atoi(e) : -1;
    return (v >= 0 && v < 4) ? v : -1;
}

// tune mode: HTD_X3P_BASE_SPLITS / HTD_X3P_REM_SPLITS force the two split factors of plan_x3p (0 = planned)
int forced_splits(const char *name)
{
    if (!g_x3p_tune) return 0;
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
}

// Work decomposition of one launch.  A layer whose tiles all fit on the chip at once (l3 / l4 of the backbone, P4-P6, the FC
// layers: a few hundred tiles for 256 CUs) runs as long as its most loaded CU -- 526 tiles put three on some CUs and two on
// the others, 68 % of the chip's time -- and a CU with a single workgroup cannot hide its own latencies.  So
//   * every tile is cut along K into `base` ranges when there are too few tiles for ~3 workgroups per CU, and
//   * the (tile, range) units beyond the last multiple of 256 -- the remainder -- are cut `rem` times more, into about 256
//     small workgroups that spread over every CU.
// Partial sums are added in range order by the epilogue kernel: deterministic, like the classic split-K this generalises.
// The factors minimise a small time model (rounds x unit time x a latency penalty for < 3 resident workgroups + the
// partial-sum traffic); tools/sweep_x3p.py measures it against forced factors.
struct XPlan {
    int tiles_a, splits_a, sps_a, splits_b, sps_b;
    int64_t m_rem0, grid, partial_floats;
};
constexpr int kXOcc[4] = {5, 3, 4, 4};           // resident workgroups per CU of the four tiles (x3p_occupancy)

XPlan plan_x3p(int cfg, int64_t M, int Co, int total_steps, int kw, bool have_ws)
{
    const int bm = kXCfg[cfg].bm, bn = kXCfg[cfg].bn;
    const int64_t mt = htd::ceil_div(M, bm), nt = htd::ceil_div(Co, bn), tiles = mt * nt;
    XPlan pl{(int)tiles, 1, total_steps, 1, total_steps, M, tiles, 0};
    const int fbase = forced_splits("HTD_X3P_BASE_SPLITS"), frem = forced_splits("HTD_X3P_REM_SPLITS");
    if (!have_ws || total_steps < 8 || tiles > 256 * kXOcc[cfg]) return pl;   // more than one resident round: dispatch balances
    const double tile_s = 2.0 * bm * bn * total_steps * XK * kw / (200e12 / 256);      // one tile on one CU at the large-layer rate
    // model constants; the HTD_X3P_PLAN_* variables (tune mode) override them for sweeps (tools/sweep_x3p_plan.sh)
    const int f_ov = forced_splits("HTD_X3P_PLAN_OV"), f_l1 = forced_splits("HTD_X3P_PLAN_LAT1"), f_l2 = forced_splits("HTD_X3P_PLAN_LAT2");
    const int f_mr = forced_splits("HTD_X3P_PLAN_MAXREM");
    const double ov_taps = f_ov > 0 ? f_ov : 8.0, lat1 = f_l1 > 0 ? f_l1 * 0.01 : 1.45, lat2 = f_l2 > 0 ? f_l2 * 0.01 : 1.12;
    const int max_rem = f_mr > 0 ? f_mr : 16;
    const double ov = ov_taps / (double)(total_steps * kw);                           // prologue + partial epilogue, in tile times
    // the reduce pass: one more dependent launch (~4 us in the queue) plus reading the partials back; HTD_X3P_PLAN_LAUNCH_US (tune
    // mode) overrides the constant for experiments
    const int forced_us = forced_splits("HTD_X3P_PLAN_LAUNCH_US");
    const double launch_s = forced_us > 0 ? forced_us * 1e-6 : 4e-6;
    double best_t = 1e30;
    for (int base = 1; base <= 8; ++base) {
        if (fbase > 0 && base != fbase) continue;
        if (base > 1 && total_steps / base < 8) break;
        const int64_t units = tiles * base;
        int64_t ua = (units / 256) * 256;
        ua -= ua % (nt * base);                                       // whole rows of tiles
        const int64_t ta = ua / base, r = tiles - ta, m0 = std::min<int64_t>((ta / nt) * bm, M);
        for (int rem = 1; rem <= max_rem; ++rem) {
            if (frem > 0 && rem != frem) continue;
            if (r == 0 && rem > 1) break;
            const int sb = base * rem;
            if (sb > 1 && total_steps / sb < 3) break;
            const double t_a = (double)(ua / 256) * (1.0 / base + (base > 1 ? ov : 0.0));
            const double t_b = r ? (double)htd::ceil_div(r * sb, 256) * (1.0 / sb + (sb > 1 ? ov : 0.0)) : 0.0;
            const double resident = std::min<double>((double)(ua + r * sb) / 256.0, kXOcc[cfg]);
            const double latency = resident < 1.5 ? lat1 : (resident < 2.5 ? lat2 : 1.0);
            const int64_t pf = (base > 1 ? (int64_t)base * m0 * Co : 0) + (sb > 1 ? (int64_t)sb * (M - m0) * Co : 0);
            if (pf * 4 > (128ll << 20)) continue;
            const double t = (t_a + t_b) * latency * tile_s + (double)pf * 8.0 / 4e12 + (pf ? launch_s : 0.0);
            if (t < best_t * 0.98) {
                best_t = t;
                const int spa = (int)htd::ceil_div(total_steps, base), spb = (int)htd::ceil_div(total_steps, sb);
                pl = XPlan{(int)ta, (int)htd::ceil_div(total_steps, spa), spa, (int)htd::ceil_div(total_steps, spb), spb, m0, 0, 0};
                pl.grid = ta * pl.splits_a + r * pl.splits_b;
                pl.partial_floats = (pl.splits_a > 1 ? (int64_t)pl.splits_a * m0 * Co : 0) +
                                    (pl.splits_b > 1 ? (int64_t)pl.splits_b * (M - m0) * Co : 0);
            }
        }
    }
    return pl;
}

// wave quantisation on 256 CUs x the useful fraction of the padded tiles x a per-tile base efficiency
float cfg_score(int cfg, int64_t M, int Co)
{
    const int bm = kXCfg[cfg].bm, bn = kXCfg[cfg].bn;
    const int64_t tiles = htd::ceil_div(M, bm) * htd::ceil_div(Co, bn);
    static const float base[4] = {0.80f, 1.00f, 0.90f, 0.92f};
    const float w = (float)tiles / 256.f;
    // one resident round: the remainder is spread by plan_x3p (a small loss); more: the last round is partly empty
    const float quant = tiles <= 256 * kXOcc[cfg] ? (w < 1.f ? 0.85f : 0.95f) : w / ceilf(w) * 0.3f + 0.7f;
    const float useful = (float)((double)M * Co / ((double)tiles * bm * bn));
    return base[cfg] * quant * useful;
}

// Tuned tile table, (M, Co, Ci, taps, epilogue bits) -> configuration id: measured inside the train / inference step by
// tools/tune_conv_tiles.py --kernel x3p, shipped as htd_amd/tuning/conv_x3p_tiles_gfx950.json and pushed in at load
// (htd_conv2d_x3p_tile_table_set) -- what cuDNN's algorithm search does for the reference, without a search at run time.
struct XKey {
    int64_t M;
    int Co, Ci, taps, epi;
    bool operator==(const XKey &o) const { return M == o.M && Co == o.Co && Ci == o.Ci && taps == o.taps && epi == o.epi; }
};
struct XKeyHash {
    size_t operator()(const XKey &k) const
    {
        uint64_t h = (uint64_t)k.M * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t)k.Co << 40) ^ ((uint64_t)k.Ci << 20) ^ ((uint64_t)k.taps << 4) ^ (uint64_t)k.epi;
        h *= 0xBF58476D1CE4E5B9ull;
        return (size_t)(h ^ (h >> 29));
    }
};
std::mutex g_xtable_mutex;
std::unordered_map<XKey, int, XKeyHash> g_xtable;

int table_cfg(int64_t M, int Co, int Ci, int taps, int epi)
{
    std::lock_guard<std::mutex> lock(g_xtable_mutex);
    if (g_xtable.empty()) return -1;
    const auto it = g_xtable.find(XKey{M, Co, Ci, taps, epi});
    return it == g_xtable.end() ? -1 : it->second;
}

int choose_cfg(int64_t M, int Co, int Ci, int taps, int epi)
{
    int cfg = forced_cfg();
    if (cfg < 0) cfg = table_cfg(M, Co, Ci, taps, epi);
    if (cfg >= 0) return cfg;
    float best = -1.f;
    static const int cand[4] = {1, 3, 2, 0};
    for (int c : cand) {
        const float sc = cfg_score(c, M, Co);
        if (sc > best * 1.005f) { best = sc; cfg = c; }
    }
    return cfg;
}

// HTD_X3P_MFMA=16 / 32 forces the instruction shape (0: per filter width, launch_tile_nb); re-read per call under HTD_X3P_TUNE
static const int g_x3p_mfma = getenv("HTD_X3P_MFMA") ? atoi(getenv("HTD_X3P_MFMA")) : 0;
int x3p_mfma()
{
    if (!g_x3p_tune) return g_x3p_mfma;
    const char *e = getenv("HTD_X3P_MFMA");
    return e ? atoi(e) : 0;
}

// HTD_X3P_NB (tune mode): B buffers, 0 = the per-tile default
int x3p_nb()
{
    if (!g_x3p_tune) return 0;
    const char *e = getenv("HTD_X3P_NB");
    return e ? atoi(e) : 0;
}

template <int TM, int TN, int NB>
void launch_tile_nb(const X3Params &p, int kw, dim3 grid, hipStream_t s)
{
    // the instruction shape per filter width (tools/sweep_x3p.py): 16x16x32 on the 1x1 layers, 32x32x16 on the 3x3 ones
    const int mf = x3p_mfma() ? x3p_mfma() : (kw == 1 ? 16 : 32);
    if (mf == 32) {
        if (kw == 1) hipLaunchKernelGGL((conv_x3p_kernel<2, 2, TM, TN, 1, false, NB>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_x3p_kernel<2, 2, TM, TN, 3, false, NB>), grid, dim3(256), 0, s, p);
    } else {
        if (kw == 1) hipLaunchKernelGGL((conv_x3p_kernel<2, 2, TM, TN, 1, true, NB>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_x3p_kernel<2, 2, TM, TN, 3, true, NB>), grid, dim3(256), 0, s, p);
    }
}

// Two B buffers (one tap of prefetch) everywhere: measured with three and four (tools/sweep_x3p.py, SWEEP_NB), the longer
// prefetch never paid for the workgroup per CU its LDS costs -- co-resident workgroups already cover one tap of latency.
// -DHTD_X3P_DEEP builds the deeper rings for that experiment.
template <int TM, int TN>
void launch_tile(const X3Params &p, int kw, dim3 grid, hipStream_t s)
{
#ifdef HTD_X3P_DEEP
    const int nb = x3p_nb();
    if (nb == 3) return launch_tile_nb<TM, TN, 3>(p, kw, grid, s);
    if (nb >= 4) return launch_tile_nb<TM, TN, 4>(p, kw, grid, s);
#endif
    launch_tile_nb<TM, TN, 2>(p, kw, grid, s);
}

// HTD_X3Q_MFMA=16 / 32, HTD_X3Q_NS=2 / 3 (tune mode re-reads them per call): instruction shape and LDS stages of conv_x3q_kernel
static const int g_x3q_mfma = getenv("HTD_X3Q_MFMA") ? atoi(getenv("HTD_X3Q_MFMA")) : 0;
static const int g_x3q_ns = getenv("HTD_X3Q_NS") ? atoi(getenv("HTD_X3Q_NS")) : 0;
int x3q_env(const char *name, int at_load)
{
    if (!g_x3p_tune) return at_load;
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
}

template <int TM, int TN>
void launch_tile_q(const X3Params &p, dim3 grid, hipStream_t s)
{
    // defaults: 16x16x32 (the shape conv_x3p_kernel runs its 1x1 layers on, so a plane-fed layer returns the bits of the
    // fp32-fed one; also the fastest, tools/bench_planes.py) and two LDS stages (three cost a resident workgroup)
    const int mf = x3q_env("HTD_X3Q_MFMA", g_x3q_mfma), ns = x3q_env("HTD_X3Q_NS", g_x3q_ns);
    if (mf != 32) {
        if (ns == 3) hipLaunchKernelGGL((conv_x3q_kernel<2, 2, TM, TN, true, 3>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_x3q_kernel<2, 2, TM, TN, true, 2>), grid, dim3(256), 0, s, p);
    } else {
        if (ns == 3) hipLaunchKernelGGL((conv_x3q_kernel<2, 2, TM, TN, false, 3>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_x3q_kernel<2, 2, TM, TN, false, 2>), grid, dim3(256), 0, s, p);
    }
}

int launch_x3p(X3Params p, int kw, hipStream_t s, void *workspace)
{
    const int total_steps = p.ncs * p.kh;
    const int epi = (p.residual ? 1 : 0) | (p.mask_src ? 2 : 0) | (p.amax ? 4 : 0);          // bit 2: the H2 kernels have their own entries
    // (tap-list launches key the table with taps + 100: a strided 3x3 layer has the M, Co, Ci and tap count of its stride-1 neighbour)
    const int cfg = choose_cfg(p.M, p.Co, p.Ci, p.kh * kw + (p.ntl > 0 ? 100 : 0), epi);
    const XPlan pl = plan_x3p(cfg, p.M, p.Co, total_steps, kw, workspace != nullptr);
    p.tiles_a = pl.tiles_a; p.splits_a = pl.splits_a; p.sps_a = pl.sps_a; p.splits_b = pl.splits_b; p.sps_b = pl.sps_b;
    p.m_rem0 = pl.m_rem0;
    p.partial = (float *)workspace;
    p.mt = (int)htd::ceil_div(p.M, kXCfg[cfg].bm);
    p.nt = (int)htd::ceil_div(p.Co, kXCfg[cfg].bn);
    HTD_REQUIRE(pl.grid > 0 && pl.grid < (1ll << 31), "conv2d_x3p: bad grid");
    const dim3 grid((unsigned)pl.grid);
    if (p.amax != nullptr && kw == 3) {        // H2 arithmetic
        switch (cfg) {
        case 0: hipLaunchKernelGGL((conv_x3p_kernel<2, 2, 1, 1, 3, false, HTD_X3H_NB, true>), grid, dim3(256), 0, s, p); break;
        case 1: hipLaunchKernelGGL((conv_x3p_kernel<2, 2, 2, 2, 3, false, HTD_X3H_NB, true>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_x3p_kernel<2, 2, 2, 1, 3, false, HTD_X3H_NB, true>), grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((conv_x3p_kernel<2, 2, 1, 2, 3, false, HTD_X3H_NB, true>), grid, dim3(256), 0, s, p); break;
        }
    } else if (p.amax != nullptr) {
        switch (cfg) {
        case 0: hipLaunchKernelGGL((conv_x3p_kernel<2, 2, 1, 1, 1, false, HTD_X3H_NB, true>), grid, dim3(256), 0, s, p); break;
        case 1: hipLaunchKernelGGL((conv_x3p_kernel<2, 2, 2, 2, 1, false, HTD_X3H_NB, true>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_x3p_kernel<2, 2, 2, 1, 1, false, HTD_X3H_NB, true>), grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((conv_x3p_kernel<2, 2, 1, 2, 1, false, HTD_X3H_NB, true>), grid, dim3(256), 0, s, p); break;
        }
    } else if (p.xp != nullptr) {   // A operand pre-split: conv_x3q_kernel (1x1, stride 1)
        switch (cfg) {
        case 0: launch_tile_q<1, 1>(p, grid, s); break;
        case 1: launch_tile_q<2, 2>(p, grid, s); break;
        case 2: launch_tile_q<2, 1>(p, grid, s); break;
        default: launch_tile_q<1, 2>(p, grid, s); break;
        }
    } else {
        switch (cfg) {
        case 0: launch_tile<1, 1>(p, kw, grid, s); break;
        case 1: launch_tile<2, 2>(p, kw, grid, s); break;
        case 2: launch_tile<2, 1>(p, kw, grid, s); break;
        default: launch_tile<1, 2>(p, kw, grid, s); break;
        }
    }
    if (pl.partial_floats > 0) {
        const int64_t rows = (p.splits_a > 1 ? p.m_rem0 : 0) + (p.splits_b > 1 ? p.M - p.m_rem0 : 0);
        if ((p.Co & 3) == 0) {      // (the main kernel writes its partial rows with float4 stores under the same condition)
            // (with amax_out: one atomic per workgroup on one address -- fewer, longer workgroups)
            const unsigned rb = (unsigned)std::min<int64_t>(htd::ceil_div(rows * (p.Co >> 2), 256), p.amax_out ? 1024 : 4096);
            hipLaunchKernelGGL(conv_x3p_splitk_epilogue_vec_kernel, dim3(rb), dim3(256), 0, s, p);
        } else {
            const unsigned rb = (unsigned)std::min<int64_t>(htd::ceil_div(rows * p.Co, 256), p.amax_out ? 1024 : 4096);
            hipLaunchKernelGGL(conv_x3p_splitk_epilogue_kernel, dim3(rb), dim3(256), 0, s, p);
        }
    }
    return htd::check_launch("conv2d_x3p");
}

// strided 3x3 layers (the first block of a ResNet stage, backbones/resnet.py:260-300 `conv2` at stride 2): H2 only, tap-list mode
bool x3h_strided_ok(int Ci, int Co, int kh, int kw, int stride, int pad)
{
    static const bool on = !(getenv("HTD_X3H_STRIDED") && atoi(getenv("HTD_X3H_STRIDED")) == 0);
    return on && Ci % XK == 0 && Co >= 33 && kh == 3 && kw == 3 && stride == 2 && pad == 1;
}

bool x3p_shape_ok(int Ci, int Co, int kh, int kw, int stride, int pad, int dil)
{
    if (Ci % XK != 0 || Co < 33 || dil != 1 || kh != kw) return false;
    if (kh == 1) return pad == 0 && stride >= 1;
    return kh == 3 && stride == 1 && pad == 1;
}

}  // namespace

// 1 when htd_conv2d_fwd_x3p / htd_conv2d_bwd_data_x3p take this layer (else use htd_conv2d_fwd / htd_conv2d_bwd_data)
extern "C" int htd_conv2d_x3p_supported(int Ci, int Co, int kh, int kw, int stride, int pad, int dil)
{
    return (!x3p_off() && htd::conv_math() == 1 && x3p_shape_ok(Ci, Co, kh, kw, stride, pad, dil)) ? 1 : 0;
}

// H2 arithmetic (two fp16 pieces per operand, three products): 1 when htd_conv2d_fwd_x3h / htd_conv2d_bwd_data_x3h take the layer.
// HTD_CONV_H2=0 switches it off (every layer on the six-product bf16 form); htd_conv2d_set_h2 does the same at run time.
int g_conv_h2 = getenv("HTD_CONV_H2") ? atoi(getenv("HTD_CONV_H2")) : 1;
extern "C" int htd_conv2d_set_h2(int on)
{
    const int prev = g_conv_h2;
    if (on == 0 || on == 1) g_conv_h2 = on;
    return prev;
}
extern "C" int htd_conv2d_x3h_supported(int Ci, int Co, int kh, int kw, int stride, int pad, int dil)
{
    return (g_conv_h2 == 1 && htd_conv2d_x3p_supported(Ci, Co, kh, kw, stride, pad, dil)) ? 1 : 0;
}
// 1 when htd_conv2d_fwd_x3h also takes this STRIDED layer (3x3, stride 2, pad 1: nine taps on the kernel's 1x1 loop; workspace:
// htd_conv2d_x3p_workspace_bytes(M, Co, Ci, 9, 1)).  The six-product form has no such path: without a maximum for the input the
// layer stays on htd_conv2d_fwd.
extern "C" int htd_conv2d_x3h_strided_supported(int Ci, int Co, int kh, int kw, int stride, int pad, int dil)
{
    return (g_conv_h2 == 1 && !x3p_off() && htd::conv_math() == 1 && dil == 1 && x3h_strided_ok(Ci, Co, kh, kw, stride, pad)) ? 1 : 0;
}

// Tuned tile table of conv_x3p_kernel (see choose_cfg).  cfg: 0 64x64, 1 128x128, 2 128x64, 3 64x128; < 0 erases the entry.
// epi: bit 0 = residual / accum operand present, bit 1 = mask_src present, bit 2 = the launch runs on the H2 arithmetic.
extern "C" int htd_conv2d_x3p_tile_table_set(int64_t M, int Co, int Ci, int taps, int epi, int cfg)
{
    HTD_REQUIRE(M > 0 && Co > 0 && Ci > 0 && taps > 0 && epi >= 0 && epi < 8 && cfg < 4,
                "x3p_tile_table_set: bad entry M=%lld Co=%d Ci=%d taps=%d epi=%d cfg=%d", (long long)M, Co, Ci, taps, epi, cfg);
    std::lock_guard<std::mutex> lock(g_xtable_mutex);
    if (cfg < 0) g_xtable.erase(XKey{M, Co, Ci, taps, epi});
    else g_xtable[XKey{M, Co, Ci, taps, epi}] = cfg;
    return HTD_OK;
}

extern "C" int htd_conv2d_x3p_tile_table_clear()
{
    std::lock_guard<std::mutex> lock(g_xtable_mutex);
    g_xtable.clear();
    return HTD_OK;
}

// the configuration id a launch of this problem would use now (forced tile, table, then score)
extern "C" int htd_conv2d_x3p_tile_query(int64_t M, int Co, int Ci, int taps, int epi)
{
    return choose_cfg(M, Co, Ci, taps, epi);
}

// the work decomposition a launch of this problem would use with tile configuration cfg (0..3) and a workspace:
// out = {tiles_a, splits_a, steps_a, splits_b, steps_b, first row of region B, grid, partial floats}
extern "C" int htd_conv2d_x3p_plan_query(int cfg, int64_t M, int Co, int Ci, int kh, int kw, int64_t *out)
{
    HTD_REQUIRE(cfg >= 0 && cfg < 4 && M > 0 && Co > 0 && Ci >= XK && Ci % XK == 0 && kh > 0 && kw > 0 && out,
                "x3p_plan_query: bad arguments");
    const XPlan pl = plan_x3p(cfg, M, Co, (Ci / XK) * kh, kw, true);
    out[0] = pl.tiles_a; out[1] = pl.splits_a; out[2] = pl.sps_a; out[3] = pl.splits_b; out[4] = pl.sps_b;
    out[5] = pl.m_rem0; out[6] = pl.grid; out[7] = pl.partial_floats;
    return HTD_OK;
}

extern "C" int64_t htd_conv2d_x3_planes_bytes(int Co, int kh, int kw, int Ci, int transposed)
{
    const int N = transposed ? Ci : Co, K = transposed ? Co : Ci;
    if (N <= 0 || K <= 0 || K % XK != 0 || kh <= 0 || kw <= 0) return 0;
    // (+ the H2 image's row scales, 1 / sb and sb, and the chunk maxima they are made from)
    return (int64_t)kh * kw * (K / XK) * NCH * planes_np(N) * 16 + (int64_t)(2 + h2_chunks(kh * kw, K)) * planes_np(N) * 4;
}

extern "C" int htd_conv2d_x3_planes(const float *w, void *planes, int Co, int kh, int kw, int Ci, int transposed, void *stream)
{
    HTD_REQUIRE(w && planes && Co > 0 && Ci > 0 && kh > 0 && kw > 0, "x3_planes: bad arguments");
    const int N = transposed ? Ci : Co, K = transposed ? Co : Ci;
    HTD_REQUIRE(K % XK == 0, "x3_planes: reduction length %d must be a multiple of 16", K);
    HTD_REQUIRE(transposed || Ci % 4 == 0, "x3_planes: Ci %% 4");
    const int Np = planes_np(N);
    const int64_t total = (int64_t)kh * kw * (K / XK) * Np;
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 16384);
    hipLaunchKernelGGL(x3_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (uint4 *)planes, Co, kh * kw, Ci, Np,
                       transposed);
    return htd::check_launch("x3_planes");
}

// The H2 image of one weight (layout of htd_conv2d_x3_planes, fp16 pieces in chunks 0..3, row scales behind the planes).
extern "C" int htd_conv2d_x3h_planes(const float *w, void *planes, int Co, int kh, int kw, int Ci, int transposed, void *stream)
{
    HTD_REQUIRE(w && planes && Co > 0 && Ci > 0 && kh > 0 && kw > 0, "x3h_planes: bad arguments");
    const int N = transposed ? Ci : Co, K = transposed ? Co : Ci;
    HTD_REQUIRE(K % XK == 0, "x3h_planes: reduction length %d must be a multiple of 16", K);
    HTD_REQUIRE(transposed || Ci % 4 == 0, "x3h_planes: Ci %% 4");
    const int Np = planes_np(N);
    float *scales = reinterpret_cast<float *>((uint4 *)planes + planes_vec(kh * kw, K, Np));
    hipLaunchKernelGGL(x3h_rowscale_kernel, dim3((unsigned)(Np / 64 * h2_chunks(kh * kw, K))), dim3(256), 0, (hipStream_t)stream, w,
                       scales, Co, kh * kw, Ci, Np, transposed);
    const int64_t total = (int64_t)kh * kw * (K / XK) * Np;
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 16384);
    hipLaunchKernelGGL(x3h_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (uint4 *)planes, scales, Co, kh * kw,
                       Ci, Np, transposed);
    return htd::check_launch("x3h_planes");
}

// desc: device array of n entries { const float *w; void *planes; int Co, taps, Ci, transposed; int64_t block0, row0; } (48
// bytes); block0 as in htd_conv2d_x3_planes_many, row0 = prefix sum of (Np / 64) * ceil(taps * K / 1024) over the entries (the
// workgroups of the row-maximum pass; Np = N rounded up to 128), total_rows = its end.
extern "C" int htd_conv2d_x3h_planes_many(const void *desc, int n, int64_t total_blocks, int64_t total_rows, void *stream)
{
    static_assert(sizeof(PlanesDescH) == 48, "PlanesDescH layout is part of the ABI");
    HTD_REQUIRE(desc && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31) && total_rows > 0 && total_rows < (1ll << 31),
                "x3h_planes_many: bad arguments");
    hipLaunchKernelGGL(x3h_rowscale_many_kernel, dim3((unsigned)total_rows), dim3(256), 0, (hipStream_t)stream,
                       (const PlanesDescH *)desc, n);
    hipLaunchKernelGGL(x3h_planes_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PlanesDescH *)desc, n);
    return htd::check_launch("x3h_planes_many");
}

// *amax = max(*amax, max |x[i]|): *amax must hold zero (or an earlier maximum) on entry; NaN in x makes it NaN.
extern "C" int htd_absmax(const float *x, int64_t n, float *amax, void *stream)
{
    HTD_REQUIRE(x && amax && n > 0 && ((uintptr_t)x & 15) == 0, "absmax: bad arguments");
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(n >> 2, (int64_t)256 * 8) + 1, 2048);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, (unsigned *)amax);
    return htd::check_launch("absmax");
}

// desc: device array of n entries { const float *w; void *planes; int Co, taps, Ci, transposed; int64_t block0; } (40 bytes),
// block0 = prefix sum of ceil(taps * (K / 16) * Np / 256) over the entries, total_blocks = its end.
extern "C" int htd_conv2d_x3_planes_many(const void *desc, int n, int64_t total_blocks, void *stream)
{
    static_assert(sizeof(PlanesDesc) == 40, "PlanesDesc layout is part of the ABI");
    HTD_REQUIRE(desc && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "x3_planes_many: bad arguments");
    hipLaunchKernelGGL(x3_planes_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PlanesDesc *)desc, n);
    return htd::check_launch("x3_planes_many");
}

// bytes of partial sums a launch of this problem may need: the largest over the tile configurations (the table / a forced
// tile picks one at launch time)
extern "C" int64_t htd_conv2d_x3p_workspace_bytes(int64_t M, int Co, int Ci, int kh, int kw)
{
    int64_t need = 0;
    for (int cfg = 0; cfg < 4; ++cfg) need = std::max(need, plan_x3p(cfg, M, Co, (Ci / XK) * kh, kw, true).partial_floats * 4);
    return need;
}

// rows per chunk array / bytes of the activation planes of a map with M pixels and C channels (C % 16 == 0)
extern "C" int64_t htd_act_planes_rows(int64_t M) { return M > 0 ? act_rows(M) : 0; }

extern "C" int64_t htd_act_planes_bytes(int64_t M, int C)
{
    if (M <= 0 || C <= 0 || C % XK != 0) return 0;
    return (int64_t)(C / XK) * NCH * act_rows(M) * 16;
}

// planes of the fp32 map x [M][C]: what the convolutions below write through `yplanes` for the values they store to y
extern "C" int htd_act_planes(const float *x, void *planes, int64_t M, int C, void *stream)
{
    HTD_REQUIRE(x && planes && M > 0 && C > 0 && C % XK == 0, "act_planes: bad arguments (C %% 16)");
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(M * (C >> 3), (int64_t)256), 65536);
    hipLaunchKernelGGL(act_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (uint4 *)planes, M, C, act_rows(M));
    return htd::check_launch("act_planes");
}

// htd_conv2d_fwd_x3p with the activation planes of round 4:
//   xplanes  != NULL: the input map's planes (htd_act_planes layout, M = B*H*W pixels); 1x1 / stride 1 / pad 0 layers only;
//                     x is then not read and may be NULL (conv_x3q_kernel: both operands by LDS-DMA)
//   yplanes  != NULL: the planes of y (Co % 16 == 0), written next to y by the same epilogue
extern "C" int htd_conv2d_fwd_x3q(const float *x, const void *xplanes, const void *wplanes, const float *bias,
                                  const float *residual, int res_h, int res_w, float *y, void *yplanes, float *amax_out, int B, int H,
                                  int W, int Ci, int Co, int kh, int kw, int stride, int pad, int relu, void *workspace, void *stream)
{
    HTD_REQUIRE((x || xplanes) && wplanes && y && B > 0 && H > 0 && W > 0, "conv2d_fwd_x3q: bad arguments");
    HTD_REQUIRE(x3p_shape_ok(Ci, Co, kh, kw, stride, pad, 1), "conv2d_fwd_x3q: unsupported layer Ci=%d Co=%d k=%dx%d s=%d p=%d", Ci,
                Co, kh, kw, stride, pad);
    HTD_REQUIRE(!xplanes || (kh == 1 && stride == 1), "conv2d_fwd_x3q: input planes serve 1x1 / stride-1 layers only");
    HTD_REQUIRE(!yplanes || Co % XK == 0, "conv2d_fwd_x3q: output planes need Co %% 16 == 0 (Co=%d)", Co);
    HTD_REQUIRE((res_h > 0) == (res_w > 0) && res_h >= 0 && (res_h == 0 || ((Co & 3) == 0 && residual)),
                "conv2d_fwd_x3q: bad residual up-sampling arguments");
    X3Params p{};
    p.x = x; p.wp = (const uint4 *)wplanes; p.bias = bias; p.residual = residual; p.y = y;
    p.Hx = H; p.Wx = W;
    p.Ho = (H + 2 * pad - kh) / stride + 1;
    p.Wo = (W + 2 * pad - kw) / stride + 1;
    p.Ci = Ci; p.Co = Co; p.Cop = planes_np(Co); p.kh = kh; p.stride = stride; p.relu = relu;
    p.M = (int64_t)B * p.Ho * p.Wo;
    p.ncs = Ci / XK;
    p.xp = (const uint4 *)xplanes; p.xp_rows = act_rows(p.M);
    p.yp = (uint4 *)yplanes; p.yp_rows = act_rows(p.M);
    p.amax_out = amax_out;
    HTD_REQUIRE((int64_t)B * H * W * Ci < (1ll << 31) && p.M * Co < (1ll << 40), "conv2d_fwd_x3q: operand too large");
    if (res_h > 0) {
        p.res_H = res_h; p.res_W = res_w;
        p.res_sh = (float)res_h / (float)p.Ho; p.res_sw = (float)res_w / (float)p.Wo;
    }
    return launch_x3p(p, kw, (hipStream_t)stream, workspace);
}

extern "C" int htd_conv2d_fwd_x3p(const float *x, const void *wplanes, const float *bias, const float *residual, int res_h,
                                  int res_w, float *y, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                                  int relu, void *workspace, void *stream)
{
    HTD_REQUIRE(x, "conv2d_fwd_x3p: bad arguments");
    return htd_conv2d_fwd_x3q(x, nullptr, wplanes, bias, residual, res_h, res_w, y, nullptr, nullptr, B, H, W, Ci, Co, kh, kw, stride,
                              pad, relu, workspace, stream);
}

// gx[B][H][W][Ci] from gy[B][Ho][Wo][Co] and the transposed planes of the layer's weights (htd_conv2d_x3_planes(...,
// transposed = 1)); stride 1 only (kh = 1: pad 0; kh = 3: pad 1).  mask_src / accum as htd_conv2d_bwd_data.
// gyplanes != NULL (1x1 layers): the planes of gy, gy itself is then not read; gxplanes != NULL: the planes of gx (Ci % 16 == 0).
extern "C" int htd_conv2d_bwd_data_x3q(const float *gy, const void *gyplanes, const void *wplanesT, const float *mask_src,
                                       const float *accum, float *gx, void *gxplanes, float *amax_out, int B, int H, int W, int Ci,
                                       int Co, int kh, int kw, int pad, void *workspace, void *stream)
{
    HTD_REQUIRE((gy || gyplanes) && wplanesT && gx && B > 0 && H > 0 && W > 0, "conv2d_bwd_data_x3q: bad arguments");
    HTD_REQUIRE(x3p_shape_ok(Co, Ci, kh, kw, 1, pad, 1), "conv2d_bwd_data_x3q: unsupported layer Ci=%d Co=%d k=%dx%d p=%d", Ci, Co,
                kh, kw, pad);
    HTD_REQUIRE(!gyplanes || kh == 1, "conv2d_bwd_data_x3q: gradient planes serve 1x1 layers only");
    HTD_REQUIRE(!gxplanes || Ci % XK == 0, "conv2d_bwd_data_x3q: output planes need Ci %% 16 == 0 (Ci=%d)", Ci);
    X3Params p{};
    p.x = gy; p.wp = (const uint4 *)wplanesT; p.residual = accum; p.mask_src = mask_src; p.y = gx;
    p.Hx = H; p.Wx = W; p.Ho = H; p.Wo = W;          // stride 1, same size
    p.Ci = Co; p.Co = Ci; p.Cop = planes_np(Ci); p.kh = kh; p.stride = 1;
    p.M = (int64_t)B * H * W;
    p.ncs = Co / XK;
    p.xp = (const uint4 *)gyplanes; p.xp_rows = act_rows(p.M);
    p.yp = (uint4 *)gxplanes; p.yp_rows = act_rows(p.M);
    p.amax_out = amax_out;
    HTD_REQUIRE((int64_t)B * H * W * Co < (1ll << 31), "conv2d_bwd_data_x3q: operand too large");
    return launch_x3p(p, kw, (hipStream_t)stream, workspace);
}

extern "C" int htd_conv2d_bwd_data_x3p(const float *gy, const void *wplanesT, const float *mask_src, const float *accum,
                                       float *gx, int B, int H, int W, int Ci, int Co, int kh, int kw, int pad, void *workspace,
                                       void *stream)
{
    HTD_REQUIRE(gy, "conv2d_bwd_data_x3p: bad arguments");
    return htd_conv2d_bwd_data_x3q(gy, nullptr, wplanesT, mask_src, accum, gx, nullptr, nullptr, B, H, W, Ci, Co, kh, kw, pad,
                                   workspace, stream);
}

// The 3x3 layers on the H2 arithmetic (two fp16 pieces per operand, three matrix instructions per product block instead of six;
// see h2_scale above).  amax: device scalar >= max |x| of the whole input tensor (htd_absmax); wplanes: htd_conv2d_x3h_planes.
extern "C" int htd_conv2d_fwd_x3h(const float *x, const float *amax, const void *wplanes, const float *bias, const float *residual,
                                  int res_h, int res_w, float *y, void *yplanes, float *amax_out, int B, int H, int W, int Ci,
                                  int Co, int kh, int kw, int stride, int pad, int relu, void *workspace, void *stream)
{
    HTD_REQUIRE(!yplanes || Co % XK == 0, "conv2d_fwd_x3h: output planes need Co %% 16 == 0 (Co=%d)", Co);
    HTD_REQUIRE(x && amax && wplanes && y && B > 0 && H > 0 && W > 0, "conv2d_fwd_x3h: bad arguments");
    const bool taps = x3h_strided_ok(Ci, Co, kh, kw, stride, pad);      // 3x3 / stride 2: nine taps on the 1x1 loop
    HTD_REQUIRE(taps || x3p_shape_ok(Ci, Co, kh, kw, stride, pad, 1), "conv2d_fwd_x3h: unsupported layer Ci=%d Co=%d k=%dx%d s=%d p=%d",
                Ci, Co, kh, kw, stride, pad);
    HTD_REQUIRE(!taps || (!yplanes && res_h == 0), "conv2d_fwd_x3h: strided 3x3 layers write no planes and take no up-sampled residual");
    HTD_REQUIRE((res_h > 0) == (res_w > 0) && res_h >= 0 && (res_h == 0 || ((Co & 3) == 0 && residual)),
                "conv2d_fwd_x3h: bad residual up-sampling arguments");
    X3Params p{};
    p.x = x; p.wp = (const uint4 *)wplanes; p.bias = bias; p.residual = residual; p.y = y;
    p.Hx = H; p.Wx = W;
    p.Ho = (H + 2 * pad - kh) / stride + 1;
    p.Wo = (W + 2 * pad - kw) / stride + 1;
    p.Ci = Ci; p.Co = Co; p.Cop = planes_np(Co); p.kh = kh; p.stride = stride; p.relu = relu;
    p.M = (int64_t)B * p.Ho * p.Wo;
    p.ncs = Ci / XK;
    p.amax = amax;
    p.wscale = reinterpret_cast<const float *>(p.wp + planes_vec(kh * kw, Ci, p.Cop));
    p.yp = (uint4 *)yplanes; p.yp_rows = act_rows(p.M);
    p.amax_out = amax_out;
    HTD_REQUIRE((int64_t)B * H * W * Ci < (1ll << 31) && p.M * Co < (1ll << 40), "conv2d_fwd_x3h: operand too large");
    if (res_h > 0) {
        p.res_H = res_h; p.res_W = res_w;
        p.res_sh = (float)res_h / (float)p.Ho; p.res_sw = (float)res_w / (float)p.Wo;
    }
    if (taps) {
        p.kh = p.ntl = kh * kw;
        for (int t = 0; t < p.ntl; ++t) { p.tl_dy[t] = t / kw - pad; p.tl_dx[t] = t % kw - pad; p.tl_w[t] = t; }
        return launch_x3p(p, 1, (hipStream_t)stream, workspace);
    }
    return launch_x3p(p, kw, (hipStream_t)stream, workspace);
}

extern "C" int htd_conv2d_bwd_data_x3h(const float *gy, const float *amax, const void *wplanesT, const float *mask_src,
                                       const float *accum, float *gx, void *gxplanes, float *amax_out, int B, int H, int W,
                                       int Ci, int Co, int kh, int kw, int pad, void *workspace, void *stream)
{
    HTD_REQUIRE(!gxplanes || Ci % XK == 0, "conv2d_bwd_data_x3h: output planes need Ci %% 16 == 0 (Ci=%d)", Ci);
    HTD_REQUIRE(gy && amax && wplanesT && gx && B > 0 && H > 0 && W > 0, "conv2d_bwd_data_x3h: bad arguments");
    HTD_REQUIRE(x3p_shape_ok(Co, Ci, kh, kw, 1, pad, 1), "conv2d_bwd_data_x3h: unsupported layer Ci=%d Co=%d k=%dx%d p=%d", Ci, Co, kh,
                kw, pad);
    X3Params p{};
    p.x = gy; p.wp = (const uint4 *)wplanesT; p.residual = accum; p.mask_src = mask_src; p.y = gx;
    p.Hx = H; p.Wx = W; p.Ho = H; p.Wo = W;          // stride 1, same size
    p.Ci = Co; p.Co = Ci; p.Cop = planes_np(Ci); p.kh = kh; p.stride = 1;
    p.M = (int64_t)B * H * W;
    p.ncs = Co / XK;
    p.amax = amax;
    p.wscale = reinterpret_cast<const float *>(p.wp + planes_vec(kh * kw, Co, p.Cop));
    p.yp = (uint4 *)gxplanes; p.yp_rows = act_rows(p.M);
    p.amax_out = amax_out;
    HTD_REQUIRE((int64_t)B * H * W * Co < (1ll << 31), "conv2d_bwd_data_x3h: operand too large");
    return launch_x3p(p, kw, (hipStream_t)stream, workspace);
}

// ---- strided data gradients on the H2 kernel (tap-list mode of conv_x3p_kernel's 1x1 loop) -----------------------------------
// gx [B][H][W][Ci] of a stride-s layer falls into s*s parity classes of output pixels; class (ph, pw) receives only the filter taps
// with (ph + pad - ky) % s == 0 (likewise kx), and those form a dense stride-1 problem on the sub-grid hi = ph + s i over gy:
// one launch per class with its tap list and a strided output map (X3Params::o_step), classes without taps are zeros
// (conv_fwd.hip::htd_conv2d_bwd_data does the same on conv_igemm_kernel).  Reference role: cuDNN's backward-data behind the
// strided `conv2` and `downsample` convolutions of backbones/resnet.py:260-300,330-350.
namespace {
struct StridedClass { int ntaps; int dy[9], dx[9], w[9]; int Ho, Wo; };
int strided_classes(int H, int W, int kh, int kw, int stride, int pad, StridedClass *cls)          // -> number of classes with taps
{
    int n = 0;
    for (int c = 0; c < stride * stride; ++c) {
        const int ph = c / stride, pw = c % stride;
        StridedClass &q = cls[c];
        q.ntaps = 0;
        q.Ho = ph < H ? (H - ph + stride - 1) / stride : 0;
        q.Wo = pw < W ? (W - pw + stride - 1) / stride : 0;
        if (q.Ho == 0 || q.Wo == 0) continue;
        for (int ky = 0; ky < kh; ++ky) {
            const int ry = ph + pad - ky;
            if (((ry % stride) + stride) % stride != 0) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int rx = pw + pad - kx;
                if (((rx % stride) + stride) % stride != 0) continue;
                q.dy[q.ntaps] = ry >= 0 ? ry / stride : -((-ry) / stride);
                q.dx[q.ntaps] = rx >= 0 ? rx / stride : -((-rx) / stride);
                q.w[q.ntaps] = (kh - 1 - ky) * kw + (kw - 1 - kx);          // tap of the flipped / transposed image
                ++q.ntaps;
            }
        }
        n += q.ntaps > 0;
    }
    return n;
}
bool x3h_strided_dgrad_ok(int Ci, int Co, int kh, int kw, int stride, int pad)
{
    static const bool on = !(getenv("HTD_X3H_STRIDED") && atoi(getenv("HTD_X3H_STRIDED")) == 0);
    return on && Co % XK == 0 && Ci >= 33 && stride == 2 && ((kh == 3 && kw == 3 && pad == 1) || (kh == 1 && kw == 1 && pad == 0));
}
}  // namespace

extern "C" int htd_conv2d_bwd_data_x3h_strided_supported(int Ci, int Co, int kh, int kw, int stride, int pad, int dil)
{
    return (g_conv_h2 == 1 && !x3p_off() && htd::conv_math() == 1 && dil == 1 && x3h_strided_dgrad_ok(Ci, Co, kh, kw, stride, pad)) ? 1 : 0;
}

extern "C" int64_t htd_conv2d_bwd_data_x3h_strided_workspace_bytes(int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad)
{
    if (!x3h_strided_dgrad_ok(Ci, Co, kh, kw, stride, pad) || B <= 0 || H <= 0 || W <= 0) return 0;
    StridedClass cls[4];
    strided_classes(H, W, kh, kw, stride, pad, cls);
    int64_t need = 0;
    for (int c = 0; c < stride * stride; ++c)
        if (cls[c].ntaps > 0)
            need = std::max(need, htd_conv2d_x3p_workspace_bytes((int64_t)B * cls[c].Ho * cls[c].Wo, Ci, Co, cls[c].ntaps, 1));
    return need;
}

// gy [B][Ho][Wo][Co] (Ho, Wo = the layer's output size), amax: device scalar holding max |gy|, wplanesT: the transposed H2 image of
// the layer's weights (htd_conv2d_x3h_planes, transposed = 1), mask_src (may be NULL): gx is zeroed where mask_src <= 0,
// amax_out (may be NULL): max |gx| is left there.  workspace: htd_conv2d_bwd_data_x3h_strided_workspace_bytes (NULL: no K ranges).
extern "C" int htd_conv2d_bwd_data_x3h_strided(const float *gy, const float *amax, const void *wplanesT, const float *mask_src,
                                               float *gx, float *amax_out, int B, int H, int W, int Ci, int Co, int kh, int kw,
                                               int stride, int pad, void *workspace, void *stream)
{
    HTD_REQUIRE(gy && amax && wplanesT && gx && B > 0 && H > 0 && W > 0, "conv2d_bwd_data_x3h_strided: bad arguments");
    HTD_REQUIRE(x3h_strided_dgrad_ok(Ci, Co, kh, kw, stride, pad), "conv2d_bwd_data_x3h_strided: unsupported layer Ci=%d Co=%d k=%dx%d s=%d p=%d",
                Ci, Co, kh, kw, stride, pad);
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    HTD_REQUIRE(Ho > 0 && Wo > 0 && (int64_t)B * Ho * Wo * Co < (1ll << 31) && (int64_t)B * H * W * Ci < (1ll << 40),
                "conv2d_bwd_data_x3h_strided: bad sizes");
    StridedClass cls[4];
    const int live = strided_classes(H, W, kh, kw, stride, pad, cls);
    if (live < stride * stride) {       // classes without taps (the 1x1 shortcut: three of four) are plain zeros
        if (hipMemsetAsync(gx, 0, (size_t)B * H * W * Ci * 4, (hipStream_t)stream) != hipSuccess) {
            htd::set_error("conv2d_bwd_data_x3h_strided: memset failed");
            return HTD_ERR_LAUNCH;
        }
    }
    X3Params p{};
    p.x = gy; p.wp = (const uint4 *)wplanesT; p.mask_src = mask_src; p.y = gx;
    p.Hx = Ho; p.Wx = Wo;
    p.Ci = Co; p.Co = Ci; p.Cop = planes_np(Ci); p.stride = 1;
    p.ncs = Co / XK;
    p.amax = amax;
    p.wscale = reinterpret_cast<const float *>(p.wp + planes_vec(kh * kw, Co, p.Cop));
    p.amax_out = amax_out;
    p.o_step = stride; p.oH = H; p.oW = W;
    for (int c = 0; c < stride * stride; ++c) {
        const StridedClass &q = cls[c];
        if (q.ntaps == 0) continue;
        X3Params r = p;
        r.Ho = q.Ho; r.Wo = q.Wo;
        r.M = (int64_t)B * q.Ho * q.Wo;
        r.kh = r.ntl = q.ntaps;
        for (int t = 0; t < q.ntaps; ++t) { r.tl_dy[t] = q.dy[t]; r.tl_dx[t] = q.dx[t]; r.tl_w[t] = q.w[t]; }
        r.o_h0 = c / stride; r.o_w0 = c % stride;
        const int st = launch_x3p(r, 1, (hipStream_t)stream, workspace);
        if (st) return st;
    }
    return HTD_OK;
}

