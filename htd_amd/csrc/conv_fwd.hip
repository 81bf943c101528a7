// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   Y[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] + R[m][n] )
//   m = (b, ho, wo) output pixel, n = output channel, k = (kh, kw, ci)
//
// NHWC activations make every A row a contiguous run of Ci floats per filter tap, and KRSC weights make
// every B row contiguous in the same k order, so both operands are staged global -> registers -> LDS as
// 16-byte vectors with no im2col buffer (im2col happens in the address arithmetic; padding = zero fill).
//
// Tiling (64-wide wavefronts): block = 256 threads = 4 waves in a 2x2 grid, each wave owns a 64x64
// accumulator tile = 2x2 MFMA 32x32 blocks (64 accumulator VGPRs).  K is walked in BK-float slices,
// double-buffered in LDS (one barrier per slice).  LDS rows are padded by 16 B so the ds_read_b128
// fragment reads (lane = row, two K-halves per wave) are bank-conflict free for BK in {8,16,32}.
// The MFMA k index is permuted (lane half h takes floats 4h..4h+3 of an 8-float group) -- legal because A
// and B use the same permutation -- which lets one ds_read_b128 per operand block feed four MFMAs.
//
// The same kernel serves: forward conv, Linear layers (1x1 on "pixels" = rows), and the data gradient
// (forward conv of gy with spatially flipped, channel-transposed weights; stride-2 data gradients pass
// `in_dilate`, which treats gy as zero-stuffed without materialising it).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct ConvParams {
    const float *x, *w, *bias, *residual, *mask_src;
    float *y;
    int B, H, W, Ci, Co, kh, kw, stride, pad, dil, Ho, Wo;
    int in_dilate;      // >1: the input is a zero-stuffed view of x (x sample every in_dilate pixels)
    int Hx, Wx;         // physical size of x when in_dilate > 1
    int relu;
    int64_t M;          // B*Ho*Wo
    int mt, nt;         // tiles along M, N
};

template <int BK>
struct Tile {
    static constexpr int BM = 128, BN = 128;
    static constexpr int LDS_STRIDE = BK + 4;                 // floats
    static constexpr int VEC_PER_ROW = BK / 4;                // float4 per row slice
    static constexpr int ROWS_PER_PASS = 256 / VEC_PER_ROW;   // rows covered by the 256 threads at once
    static constexpr int PASSES = BM / ROWS_PER_PASS;         // float4 loads per thread per operand
};

template <int BK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p)
{
    using T = Tile<BK>;
    __shared__ __attribute__((aligned(16))) float lds[2][(T::BM + T::BN) * T::LDS_STRIDE];

    // XCD-aware tile order: blocks that share an XCD (ids congruent mod 8) walk neighbouring M tiles of the
    // same N tile, so the weight panel and the overlapping input rows stay in that XCD's L2.
    const int nblk = p.mt * p.nt;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid / p.mt, tile_m = bid % p.mt;
    const int64_t m0 = (int64_t)tile_m * T::BM;
    const int n0 = tile_n * T::BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- per-thread staging coordinates
    const int vcol = tid % T::VEC_PER_ROW;            // which float4 of the K slice
    const int vrow = tid / T::VEC_PER_ROW;            // first row handled
    int a_hi0[T::PASSES], a_wi0[T::PASSES];
    int64_t a_img[T::PASSES];
    bool a_ok[T::PASSES];
#pragma unroll
    for (int i = 0; i < T::PASSES; ++i) {
        const int64_t m = m0 + vrow + i * T::ROWS_PER_PASS;
        a_ok[i] = m < p.M;
        const int64_t mm = a_ok[i] ? m : 0;
        const int wo = (int)(mm % p.Wo);
        const int64_t t = mm / p.Wo;
        const int ho = (int)(t % p.Ho);
        const int b = (int)(t / p.Ho);
        a_hi0[i] = ho * p.stride - p.pad;
        a_wi0[i] = wo * p.stride - p.pad;
        a_img[i] = (int64_t)b * p.Hx * p.Wx;
    }
    const int64_t wrow_stride = (int64_t)p.kh * p.kw * p.Ci;
    bool b_ok[T::PASSES];
    const float *b_ptr[T::PASSES];
#pragma unroll
    for (int i = 0; i < T::PASSES; ++i) {
        const int n = n0 + vrow + i * T::ROWS_PER_PASS;
        b_ok[i] = n < p.Co;
        b_ptr[i] = p.w + (int64_t)(b_ok[i] ? n : 0) * wrow_stride + vcol * 4;
    }

    const int slices_per_tap = p.Ci / BK;
    const int num_slices = p.kh * p.kw * slices_per_tap;

    float4 ra[T::PASSES], rb[T::PASSES];
    auto load_slice = [&](int s) {
        const int tap = s / slices_per_tap;
        const int ci0 = (s - tap * slices_per_tap) * BK;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int i = 0; i < T::PASSES; ++i) {
            int hi = a_hi0[i] + ky * p.dil, wi = a_wi0[i] + kx * p.dil;
            bool ok = a_ok[i] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
            if (p.in_dilate > 1) {
                ok = ok && (hi % p.in_dilate == 0) && (wi % p.in_dilate == 0);
                hi /= p.in_dilate;
                wi /= p.in_dilate;
            }
            ra[i] = ok ? *reinterpret_cast<const float4 *>(p.x + (a_img[i] + (int64_t)hi * p.Wx + wi) * p.Ci + ci0 + vcol * 4)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = b_ok[i] ? *reinterpret_cast<const float4 *>(b_ptr[i] + (int64_t)tap * p.Ci + ci0)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_slice = [&](int buf) {
        float *la = lds[buf];
        float *lb = lds[buf] + T::BM * T::LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < T::PASSES; ++i) {
            const int r = vrow + i * T::ROWS_PER_PASS;
            *reinterpret_cast<float4 *>(la + r * T::LDS_STRIDE + vcol * 4) = ra[i];
            *reinterpret_cast<float4 *>(lb + r * T::LDS_STRIDE + vcol * 4) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    load_slice(0);
    store_slice(0);
    __syncthreads();
    for (int s = 0; s < num_slices; ++s) {
        const int cur = s & 1;
        if (s + 1 < num_slices) load_slice(s + 1);
        const float *la = lds[cur] + (wm * 64 + frow) * T::LDS_STRIDE + fhalf * 4;
        const float *lb = lds[cur] + (T::BM + wn * 64 + frow) * T::LDS_STRIDE + fhalf * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const float4 *>(la + i * 32 * T::LDS_STRIDE + kk * 8);
                fb[i] = *reinterpret_cast<const float4 *>(lb + i * 32 * T::LDS_STRIDE + kk * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (s + 1 < num_slices) store_slice(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + frow;
        if (n >= p.Co) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (m >= p.M) continue;
                const int64_t o = m * p.Co + n;
                float v = acc[i][j][r] + bv;
                if (p.residual) v += p.residual[o];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.mask_src) v = p.mask_src[o] > 0.f ? v : 0.f;   // data gradient through the producer's ReLU
                p.y[o] = v;
            }
        }
    }
}

int launch_conv(ConvParams p, hipStream_t s)
{
    p.mt = (int)htd::ceil_div(p.M, 128);
    p.nt = (int)htd::ceil_div(p.Co, 128);
    const int64_t blocks = (int64_t)p.mt * p.nt;
    HTD_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv2d: bad grid");
    if (p.Ci % 32 == 0)
        hipLaunchKernelGGL(conv_igemm_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (p.Ci % 16 == 0)
        hipLaunchKernelGGL(conv_igemm_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(conv_igemm_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return htd::check_launch("conv2d");
}

}  // namespace

extern "C" int htd_conv2d_fwd(const float *x, const float *w, const float *bias, const float *residual, float *y,
                              int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                              int relu, void *stream)
{
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
    p.in_dilate = 1; p.Hx = H; p.Wx = W; p.relu = relu;
    p.M = (int64_t)B * p.Ho * p.Wo;
    return launch_conv(p, (hipStream_t)stream);
}

// Data gradient: gx[B][H][W][Ci] from gy[B][Ho][Wo][Co] and wT[Ci][kh][kw][Co] = spatially flipped,
// channel-transposed weights (htd_conv2d_flip_weights).  mask_src (may be NULL): gx is zeroed where
// mask_src <= 0 -- the ReLU of the layer that produced the conv input, fused into this epilogue.
extern "C" int htd_conv2d_bwd_data(const float *gy, const float *wT, const float *mask_src, float *gx, int B, int H,
                                   int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, void *stream)
{
    HTD_REQUIRE(B > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0 && pad >= 0,
                "conv2d_bwd_data: bad sizes");
    HTD_REQUIRE(Co % 8 == 0, "conv2d_bwd_data: Co=%d must be a multiple of 8", Co);
    HTD_REQUIRE(gy && wT && gx, "conv2d_bwd_data: null pointer");
    const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    ConvParams p{};
    p.x = gy; p.w = wT; p.bias = nullptr; p.residual = nullptr; p.mask_src = mask_src; p.y = gx;
    // a stride-1 correlation over the (zero-stuffed) gradient map: output pixel hi reads stuffed rows
    // hi + pad' - ky'*dil with pad' = dil*(kh-1) - pad; stuffed row r is gy row r/stride when r % stride == 0
    p.B = B; p.Ci = Co; p.Co = Ci; p.kh = kh; p.kw = kw; p.stride = 1; p.dil = dil;
    p.pad = dil * (kh - 1) - pad;
    HTD_REQUIRE(dil * (kw - 1) - pad == p.pad || kh == kw, "conv2d_bwd_data: square kernels only");
    HTD_REQUIRE(p.pad >= 0, "conv2d_bwd_data: pad > dil*(k-1) unsupported");
    p.in_dilate = stride; p.Hx = Ho; p.Wx = Wo;
    p.H = (Ho - 1) * stride + 1; p.W = (Wo - 1) * stride + 1;      // extent of the stuffed map
    p.Ho = H; p.Wo = W; p.relu = 0;
    p.M = (int64_t)B * H * W;
    return launch_conv(p, (hipStream_t)stream);
}

namespace {
__global__ __launch_bounds__(256) void flip_weights_kernel(const float *__restrict__ w, float *__restrict__ wT, int Co,
                                                           int taps, int Ci)
{
    // w[co][t][ci] -> wT[ci][taps-1-t][co]; 32x32 LDS transpose per (t, co-tile, ci-tile)
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
        if (ci < Ci && co < Co) wT[((int64_t)ci * taps + (taps - 1 - t)) * Co + co] = tile[tx][r];
    }
}
}  // namespace

extern "C" int htd_conv2d_flip_weights(const float *w, float *wT, int Co, int kh, int kw, int Ci, void *stream)
{
    HTD_REQUIRE(w && wT && Co > 0 && Ci > 0 && kh > 0 && kw > 0, "flip_weights: bad arguments");
    dim3 grid((unsigned)htd::ceil_div(Ci, 32), (unsigned)htd::ceil_div(Co, 32), (unsigned)(kh * kw));
    hipLaunchKernelGGL(flip_weights_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wT, Co, kh * kw, Ci);
    return htd::check_launch("flip_weights");
}
