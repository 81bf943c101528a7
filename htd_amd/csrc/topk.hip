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

// ascending unsigned image of a float with torch.sort's equivalences: larger float <=> larger image, -0 and +0 share an image
// (ties then go by position, as the stable sort has them), every NaN -- either sign, any payload -- is the largest key.
// Output values are read back from the keys at the winning positions, so a -0 or a NaN payload comes out as it went in.
__device__ __forceinline__ unsigned ord(float v)
{
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
    if ((u << 1) == 0u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
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
    if (sg.len <= sg.k || sg.k == 0) return;         // everything / nothing survives: no threshold to find (uniform per block)
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
        bool hit = i < sg.len;
        unsigned bin = 0u;
        if (hit) {
            const unsigned o = ord(kp[i]);
            if (PASS == 0) bin = o >> 21;
            else if (PASS == 1) { hit = (o >> 21) == prefix; bin = (o >> 10) & 2047u; }
            else { hit = (o >> 10) == prefix; bin = o & 1023u; }
        }
        // a wavefront whose keys all fall into one bin (runs of equal keys: masked-out candidates, saturated scores) adds
        // its count once instead of queueing 64 atomics on one LDS word
        const unsigned long long hits = __ballot(hit);
        if (hits) {
            const int leader = __ffsll((long long)hits) - 1;
            const unsigned lead_bin = __shfl(bin, leader, 64);
            if (__ballot(hit && bin == lead_bin) == hits) {
                if ((t & 63) == leader) atomicAdd(&lh[lead_bin], (unsigned)__popcll(hits));
            } else if (hit)
                atomicAdd(&lh[bin], 1u);
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
    if (sg.len == 0 || sg.k == 0) return;
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
__global__ __launch_bounds__(1024) void topk_sort_kernel(const Seg *__restrict__ segs, const float *__restrict__ keys,
                                                         const unsigned long long *__restrict__ cand, int64_t *__restrict__ out_idx,
                                                         float *__restrict__ out_val)
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
        const int64_t pos = (int64_t)(0xffffffffu - (unsigned)(v & 0xffffffffull));
        out_idx[sg.out + i] = pos;
        out_val[sg.out + i] = keys[sg.start + pos];
    }
}

}  // namespace

extern "C" int64_t htd_segmented_topk_workspace_bytes(int S, int64_t nchunks)
{
    if (S <= 0 || nchunks < 0) return 0;
    // hist[3][S][2048] u32 | state[2][S] | n_above[S] u32 | chunk_hist[nchunks][1024] u32 | cand[S][2048] u64
    return (int64_t)3 * S * NB * 4 + (int64_t)2 * S * 8 + (int64_t)S * 4 + 16 + nchunks * 1024 * 4 + (int64_t)S * KMAX * 8 + 64;
}

namespace {

int run_topk(const float *keys, const int64_t *segs, const int32_t *chunk_tab, int S, int64_t nchunks, int64_t *out_idx,
             float *out_val, void *workspace, hipStream_t s)
{
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
    hipLaunchKernelGGL(topk_sort_kernel, dim3((unsigned)S), dim3(1024), 0, s, sg, keys, cand, out_idx, out_val);
    return htd::check_launch("segmented_topk");
}

// ---- RandomSampler on the device (base_sampler.py:34-101, random_sampler.py:33-78)
// negated keys of the positive candidates (assigned > 0) and of the negative ones (assigned == 0); everything else -2
__global__ __launch_bounds__(256) void sample_keys_kernel(const int64_t *__restrict__ assigned, const float *__restrict__ keys,
                                                          float *__restrict__ neg_keys, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t a = assigned[i];
    const float k = -keys[i];
    neg_keys[i] = a > 0 ? k : -2.f;
    neg_keys[n + i] = a == 0 ? k : -2.f;
}

// ascending bitonic sort of n (power of two <= 2048) unsigned words in LDS, 1024 threads
__device__ void sort_ascending(unsigned *w, int n)
{
    const int t = threadIdx.x;
    for (int size = 2; size <= n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = t; p < n / 2; p += 1024) {
                const int lo = 2 * p - (p & (stride - 1)), hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const unsigned a = w[lo], b = w[hi];
                if ((a > b) == asc) { w[lo] = b; w[hi] = a; }
            }
            __syncthreads();
        }
}

// one workgroup per image: how many of the ranked candidates are drawn, the two masks, the counts and (optionally) the
// slot order of SamplingResult: drawn positives by ascending index, then drawn negatives by ascending index
__global__ __launch_bounds__(1024) void sample_finish_kernel(const int64_t *__restrict__ idx, const float *__restrict__ val,
                                                             int B, int64_t A, int kpos, int kneg, int num, float neg_pos_ub,
                                                             unsigned char *__restrict__ pos_mask,
                                                             unsigned char *__restrict__ neg_mask, int64_t *__restrict__ counts,
                                                             int64_t *__restrict__ order, int slots)
{
    __shared__ unsigned w[KMAX];
    __shared__ int cnt[2];
    const int b = blockIdx.x, t = threadIdx.x;
    const int64_t *ip = idx + (size_t)b * kpos, *in = idx + (size_t)B * kpos + (size_t)b * kneg;
    const float *vp = val + (size_t)b * kpos, *vn = val + (size_t)B * kpos + (size_t)b * kneg;
    if (t < 2) cnt[t] = 0;
    __syncthreads();
    // real candidates come before the -2 fillers in a descending ranking: the drawn ones are a prefix
    int c = 0;
    for (int j = t; j < kpos; j += 1024) c += vp[j] > -2.f;
    if (c) atomicAdd(&cnt[0], c);
    __syncthreads();
    const int n_pos = cnt[0];
    int64_t limit = num - n_pos;
    if (neg_pos_ub >= 0.f) {
        const int64_t ub = (int64_t)(neg_pos_ub * (float)(n_pos > 1 ? n_pos : 1));      // base_sampler.py:86-91
        limit = limit < ub ? limit : ub;
    }
    const int lim = (int)(limit < 0 ? 0 : (limit < kneg ? limit : kneg));
    c = 0;
    for (int j = t; j < lim; j += 1024) c += vn[j] > -2.f;
    if (c) atomicAdd(&cnt[1], c);
    __syncthreads();
    const int n_neg = cnt[1];
    for (int j = t; j < n_pos; j += 1024) pos_mask[(size_t)b * A + ip[j]] = 1;
    for (int j = t; j < n_neg; j += 1024) neg_mask[(size_t)b * A + in[j]] = 1;
    if (t == 0) {
        counts[2 * b] = n_pos;
        counts[2 * b + 1] = n_neg;
    }
    if (!order) return;                                            // block-uniform
    int64_t *ob = order + (size_t)b * slots;
    for (int pass = 0; pass < 2; ++pass) {
        const int n_sel = pass ? n_neg : n_pos, first = pass ? n_pos : 0;
        const int64_t *src = pass ? in : ip;
        int n = 64;
        while (n < n_sel) n <<= 1;
        for (int j = t; j < n; j += 1024) w[j] = j < n_sel ? (unsigned)src[j] : 0xffffffffu;
        __syncthreads();
        sort_ascending(w, n);
        for (int j = t; j < n_sel; j += 1024)
            if (first + j < slots) ob[first + j] = (int64_t)w[j];
        __syncthreads();
    }
    for (int j = n_pos + n_neg + t; j < slots; j += 1024) ob[j] = 0;      // unused slots (masked by the caller)
}

}  // namespace

// segs [S][4] int64 (start, len, k, out) and chunk_tab [nchunks][2] int32 (segment, chunk inside it; the ceil(len / 4096)
// chunks of a segment consecutive and ascending) are DEVICE tables.  out_idx: position inside the segment.
extern "C" int htd_segmented_topk(const float *keys, const int64_t *segs, const int32_t *chunk_tab, int S, int64_t nchunks,
                                  int64_t *out_idx, float *out_val, void *workspace, void *stream)
{
    HTD_REQUIRE(S >= 0 && nchunks >= 0, "segmented_topk: bad sizes");
    if (S == 0) return HTD_OK;
    HTD_REQUIRE(keys && segs && out_idx && out_val && workspace && (nchunks == 0 || chunk_tab), "segmented_topk: null pointer");
    HTD_REQUIRE(nchunks < (1ll << 31), "segmented_topk: too many chunks");
    return run_topk(keys, segs, chunk_tab, S, nchunks, out_idx, out_val, workspace, (hipStream_t)stream);
}

extern "C" int64_t htd_random_sample_workspace_bytes(int B, int64_t A, int64_t nchunks)
{
    if (B <= 0 || A <= 0) return 0;
    // negated keys [2][B][A] f32 | ranked positions and keys [2B][2048] | top-k workspace
    return 2 * (int64_t)B * A * 4 + 64 + (int64_t)2 * B * KMAX * 12 + 64 + htd_segmented_topk_workspace_bytes(2 * B, nchunks);
}

// assigned [B][A] (AssignResult.gt_inds: > 0 positive, 0 negative, < 0 ignored), keys [B][A] i.i.d. in [0, 1): the drawn sample
// is the candidates with the smallest keys.  segs / chunk_tab: the tables of htd_segmented_topk for the 2B segments
// (b * A, A, kpos, b * kpos) and (B * A + b * A, A, kneg, B * kpos + b * kneg), kpos = min(max_pos, A), kneg = min(num, A).
// pos_mask / neg_mask [B][A] bytes (0 / 1), counts [B][2] = (positives, negatives) drawn, order [B][slots] (may be NULL).
extern "C" int htd_random_sample(const int64_t *assigned, const float *keys, int B, int64_t A, int num, int max_pos,
                                 float neg_pos_ub, const int64_t *segs, const int32_t *chunk_tab, int64_t nchunks,
                                 unsigned char *pos_mask, unsigned char *neg_mask, int64_t *counts, int64_t *order, int slots,
                                 void *workspace, void *stream)
{
    HTD_REQUIRE(B > 0 && A > 0 && num > 0 && num <= KMAX && max_pos >= 0 && max_pos <= num && slots >= 0,
                "random_sample: bad sizes (num <= %d)", KMAX);
    HTD_REQUIRE(assigned && keys && segs && chunk_tab && pos_mask && neg_mask && counts && workspace, "random_sample: null pointer");
    HTD_REQUIRE(A < (1ll << 31) && nchunks > 0 && nchunks < (1ll << 31), "random_sample: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int kpos = (int)(max_pos < A ? max_pos : A), kneg = (int)(num < A ? num : A);
    char *ws = (char *)workspace;
    float *neg_keys = (float *)ws;
    size_t off = ((size_t)2 * B * A * 4 + 63) / 64 * 64;
    int64_t *idx = (int64_t *)(ws + off);
    off += (size_t)2 * B * KMAX * 8;
    float *val = (float *)(ws + off);
    off += ((size_t)2 * B * KMAX * 4 + 63) / 64 * 64;
    if (hipMemsetAsync(pos_mask, 0, (size_t)B * A, s) != hipSuccess || hipMemsetAsync(neg_mask, 0, (size_t)B * A, s) != hipSuccess) {
        htd::set_error("random_sample: memset failed");
        return HTD_ERR_LAUNCH;
    }
    const int64_t n = (int64_t)B * A;
    hipLaunchKernelGGL(sample_keys_kernel, dim3((unsigned)htd::ceil_div(n, 256)), dim3(256), 0, s, assigned, keys, neg_keys, n);
    const int rc = run_topk(neg_keys, segs, chunk_tab, 2 * B, nchunks, idx, val, ws + off, s);
    if (rc != HTD_OK) return rc;
    hipLaunchKernelGGL(sample_finish_kernel, dim3((unsigned)B), dim3(1024), 0, s, idx, val, B, A, kpos, kneg, num, neg_pos_ub,
                       pos_mask, neg_mask, counts, order, slots);
    return htd::check_launch("random_sample");
}
