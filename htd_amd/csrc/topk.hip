// Segmented exact top-k with sorted output: for every segment (a run of float keys) the k largest keys in descending order,
// ties by ascending position -- the result of `keys.sort(descending=True, stable=True)[:k]` (rpn_head.py:122-133 sorts every
// level's scores and keeps nms_pre of them; base_sampler / random_sampler pick the n smallest random keys the same way).
//
// A radix select over the order-preserving 32-bit image of the keys finds the k-th largest key in three histogram passes
// (11 + 11 + 10 bits) that every workgroup of the grid shares (4096 keys per workgroup, LDS histogram, one integer atomic
// per non-empty bin), the survivors -- keys above the threshold, and the first few equal to it in position order -- are
// compacted, and one workgroup per segment sorts its <= 2048 survivors in LDS (bitonic, 64-bit (key, position) words).
// Six launches for all segments of a call, every one of them fills the chip; nothing here depends on the order atomics land.
#include "common.h"

namespace {

constexpr int CHUNK = 4096;          // keys per workgroup of the select passes
constexpr int KMAX = 2048;           // survivors per segment (the LDS sort)
constexpr int NB = 2048;             // histogram bins per pass (the last pass uses 1024 of them)

struct Seg {                         // one row of the segment table (device, int64 x 4)
    int64_t start;                   // first key
    int64_t len;
    int64_t k;                       // 0 < k <= min(len, KMAX)  (len == 0: k == 0)
    int64_t out;                     // first output slot
};

// ascending unsigned image of a float: larger float <=> larger image (-0 < +0; a positive NaN is the largest)
__device__ __forceinline__ unsigned ord(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Block-wide (256 threads): the bin b, counted from the TOP, in which the running count reaches `want` (1-based):
//   above(b) < want <= above(b) + hist[b],  above(b) = sum of the bins > b.   -> b, and want - above(b) through `rest`.
__device__ int pick_bin(const unsigned *__restrict__ hist, int nb, unsigned want, unsigned &rest, unsigned *sm /* >= 258 */)
{
    const int t = threadIdx.x, per = nb / 256;
    unsigned loc[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        loc[i] = i < per ? hist[t * per + i] : 0u;
        s += loc[i];
    }
    // inclusive suffix sums over the threads (thread 255 owns the top bins)
    sm[t] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned add = t + d < 256 ? sm[t + d] : 0u;
        __syncthreads();
        sm[t] += add;
        __syncthreads();
    }
    unsigned above = sm[t] - s;                      // keys in the bins of the higher threads
    if (t == 0) { sm[256] = 0u; sm[257] = 0u; }
    __syncthreads();
    for (int i = per - 1; i >= 0; --i) {
        if (above < want && want <= above + loc[i]) {
            sm[256] = (unsigned)(t * per + i);
            sm[257] = want - above;
        }
        above += loc[i];
    }
    __syncthreads();
    rest = sm[257];
    const int b = (int)sm[256];
    __syncthreads();
    return b;
}

struct State { unsigned prefix, want; };            // per pass and segment: key bits fixed so far, rank still to find

// PASS 0: bits 31..21;  1: bits 20..10 of the keys whose top 11 bits equal the prefix;  2: bits 9..0 under a 22-bit prefix.
// chunk_tab[c] = (segment, chunk index inside the segment).
template <int PASS>
__global__ __launch_bounds__(256) void topk_hist_kernel(const float *__restrict__ keys, const Seg *__restrict__ segs,
                                                        const int2 *__restrict__ chunk_tab, unsigned *__restrict__ hist,
                                                        State *__restrict__ state, unsigned *__restrict__ chunk_hist, int S)
{
    __shared__ unsigned lh[NB];
    __shared__ unsigned sm[258];
    const int2 ct = chunk_tab[blockIdx.x];
    const int seg = ct.x;
    const Seg sg = segs[seg];
    if (sg.len <= sg.k) return;                      // everything survives: no threshold to find (uniform per block)
    const int t = threadIdx.x;
    for (int b = t; b < NB; b += 256) lh[b] = 0u;
    unsigned prefix = 0u;
    if (PASS > 0) {
        // the choice of the previous pass, recomputed by every workgroup from the finished histogram (same inputs, same answer)
        const State prev = PASS == 1 ? State{0u, (unsigned)sg.k} : state[(PASS - 2) * S + seg];
        unsigned rest;
        const int b = pick_bin(hist + ((size_t)(PASS - 1) * S + seg) * NB, NB, prev.want, rest, sm);
        prefix = (prev.prefix << 11) | (unsigned)b;
        if (t == 0) state[(PASS - 1) * S + seg] = State{prefix, rest};
    }
    __syncthreads();
    const int64_t base = (int64_t)ct.y * CHUNK;
    const float *kp = keys + sg.start;
#pragma unroll 4
    for (int j = 0; j < CHUNK / 256; ++j) {
        const int64_t i = base + j * 256 + t;
        if (i < sg.len) {
            const unsigned o = ord(kp[i]);
            if (PASS == 0) atomicAdd(&lh[o >> 21], 1u);
            else if (PASS == 1) { if ((o >> 21) == prefix) atomicAdd(&lh[(o >> 10) & 2047u], 1u); }
            else { if ((o >> 10) == prefix) atomicAdd(&lh[o & 1023u], 1u); }
        }
    }
    __syncthreads();
    unsigned *gh = hist + ((size_t)PASS * S + seg) * NB;
    for (int b = t; b < (PASS == 2 ? 1024 : NB); b += 256) {
        const unsigned c = lh[b];
        if (c) atomicAdd(&gh[b], c);
        if (PASS == 2) chunk_hist[(size_t)blockIdx.x * 1024 + b] = c;      // ties of every value, per chunk: position order
    }
}

// survivors -> cand[seg][slot] = (image of the key) << 32 | ~position: keys above the threshold in any order, keys equal to
// it by position until the segment has k
__global__ __launch_bounds__(256) void topk_compact_kernel(const float *__restrict__ keys, const Seg *__restrict__ segs,
                                                           const int2 *__restrict__ chunk_tab, const unsigned *__restrict__ hist,
                                                           const State *__restrict__ state, const unsigned *__restrict__ chunk_hist,
                                                           unsigned *__restrict__ n_above, unsigned long long *__restrict__ cand,
                                                           int S)
{
    __shared__ unsigned sm[258];
    __shared__ unsigned wave_cnt[4];
    const int2 ct = chunk_tab[blockIdx.x];
    const int seg = ct.x, t = threadIdx.x;
    const Seg sg = segs[seg];
    if (sg.len == 0) return;
    const bool all = sg.len <= sg.k;
    unsigned thr = 0u, ties_wanted = 0u, tie_base = 0u, first_tie_slot = 0u;
    if (!all) {
        unsigned rest;
        const State prev = state[1 * S + seg];
        const int b = pick_bin(hist + ((size_t)2 * S + seg) * NB, 1024, prev.want, rest, sm);
        thr = (prev.prefix << 10) | (unsigned)b;
        ties_wanted = rest;                                        // keys equal to thr that survive
        first_tie_slot = (unsigned)sg.k - rest;                    // = number of keys above thr
        // ties in the earlier chunks of this segment (chunks of a segment are consecutive in the table)
        unsigned part = 0u;
        for (int c = t; c < ct.y; c += 256) part += chunk_hist[(size_t)(blockIdx.x - ct.y + c) * 1024 + b];
        sm[t] = part;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) {
            if (t < d) sm[t] += sm[t + d];
            __syncthreads();
        }
        tie_base = sm[0];
        __syncthreads();
    }
    const int64_t base = (int64_t)ct.y * CHUNK;
    const float *kp = keys + sg.start;
    unsigned long long *out = cand + (size_t)seg * KMAX;
    const int lane = t & 63, wave = t >> 6;
    for (int j = 0; j < CHUNK / 256; ++j) {
        const int64_t i = base + j * 256 + t;
        const bool in = i < sg.len;
        const unsigned o = in ? ord(kp[i]) : 0u;
        const unsigned long long word = ((unsigned long long)o << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
        if (in && (all || o > thr)) out[atomicAdd(&n_above[seg], 1u)] = word;
        if (!all) {                                                // block-uniform
            const bool tie = in && o == thr;
            const unsigned long long m = __ballot(tie);
            if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(m);
            __syncthreads();
            unsigned before = tie_base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) before += wave_cnt[w];
            if (tie && before < ties_wanted) out[first_tie_slot + before] = word;
            tie_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
    }
}

// one workgroup per segment: bitonic sort of the k survivors, descending (key, then lower position first)
__global__ __launch_bounds__(1024) void topk_sort_kernel(const Seg *__restrict__ segs, const unsigned long long *__restrict__ cand,
                                                         int64_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    __shared__ unsigned long long w[KMAX];
    const int seg = blockIdx.x, t = threadIdx.x;
    const Seg sg = segs[seg];
    const int k = (int)sg.k;
    if (k == 0) return;
    int n = 64;
    while (n < k) n <<= 1;                                         // sort size: power of two >= k (block-uniform)
    for (int i = t; i < n; i += 1024) w[i] = i < k ? cand[(size_t)seg * KMAX + i] : 0ull;
    __syncthreads();
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = t; p < n / 2; p += 1024) {
                const int lo = 2 * p - (p & (stride - 1));         // pairs (lo, lo + stride)
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;                // descending runs first: the final order is descending
                const unsigned long long a = w[lo], b = w[hi];
                if ((a < b) == desc) { w[lo] = b; w[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < k; i += 1024) {
        const unsigned long long v = w[i];
        out_idx[sg.out + i] = (int64_t)(0xffffffffu - (unsigned)(v & 0xffffffffull));
        out_val[sg.out + i] = unord((unsigned)(v >> 32));
    }
}

}  // namespace

extern "C" int64_t htd_segmented_topk_workspace_bytes(int S, int64_t nchunks)
{
    if (S <= 0 || nchunks < 0) return 0;
    // hist[3][S][2048] u32 | state[2][S] | n_above[S] u32 | chunk_hist[nchunks][1024] u32 | cand[S][2048] u64
    return (int64_t)3 * S * NB * 4 + (int64_t)2 * S * 8 + (int64_t)S * 4 + 16 + nchunks * 1024 * 4 + (int64_t)S * KMAX * 8 + 64;
}

// segs [S][4] int64 (start, len, k, out) and chunk_tab [nchunks][2] int32 (segment, chunk inside it; the ceil(len / 4096)
// chunks of a segment consecutive and ascending) are DEVICE tables.  out_idx: position inside the segment.
extern "C" int htd_segmented_topk(const float *keys, const int64_t *segs, const int32_t *chunk_tab, int S, int64_t nchunks,
                                  int64_t *out_idx, float *out_val, void *workspace, void *stream)
{
    HTD_REQUIRE(S >= 0 && nchunks >= 0, "segmented_topk: bad sizes");
    if (S == 0) return HTD_OK;
    HTD_REQUIRE(keys && segs && out_idx && out_val && workspace && (nchunks == 0 || chunk_tab), "segmented_topk: null pointer");
    HTD_REQUIRE(nchunks < (1ll << 31), "segmented_topk: too many chunks");
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    unsigned *hist = (unsigned *)ws;
    size_t off = (size_t)3 * S * NB * 4;
    State *state = (State *)(ws + off);
    off += (size_t)2 * S * 8;
    unsigned *n_above = (unsigned *)(ws + off);
    off += ((size_t)S * 4 + 15) / 16 * 16;
    const size_t zero_bytes = off;                                 // histograms, state and counters start at zero
    unsigned *chunk_hist = (unsigned *)(ws + off);
    off += (size_t)nchunks * 1024 * 4;
    off = (off + 63) / 64 * 64;
    unsigned long long *cand = (unsigned long long *)(ws + off);
    if (hipMemsetAsync(ws, 0, zero_bytes, s) != hipSuccess) {
        htd::set_error("segmented_topk: memset failed");
        return HTD_ERR_LAUNCH;
    }
    const Seg *sg = (const Seg *)segs;
    const int2 *ct = (const int2 *)chunk_tab;
    if (nchunks > 0) {
        const dim3 g((unsigned)nchunks), b(256);
        hipLaunchKernelGGL(topk_hist_kernel<0>, g, b, 0, s, keys, sg, ct, hist, state, chunk_hist, S);
        hipLaunchKernelGGL(topk_hist_kernel<1>, g, b, 0, s, keys, sg, ct, hist, state, chunk_hist, S);
        hipLaunchKernelGGL(topk_hist_kernel<2>, g, b, 0, s, keys, sg, ct, hist, state, chunk_hist, S);
        hipLaunchKernelGGL(topk_compact_kernel, g, b, 0, s, keys, sg, ct, hist, state, chunk_hist, n_above, cand, S);
    }
    hipLaunchKernelGGL(topk_sort_kernel, dim3((unsigned)S), dim3(1024), 0, s, sg, cand, out_idx, out_val);
    return htd::check_launch("segmented_topk");
}
