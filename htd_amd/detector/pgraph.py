"""PGraph: per (image, pyramid level) group of RoIs, local spatial aggregation followed by global
semantic interaction (HTDBBoxHead.forward, htd_bbox_head.py:195-219).

The reference runs a Python double loop over images and levels with boolean-mask gathers, `.any()`
host syncs and five tiny mm's per group.  Here ALL groups of the call are one padded batch
[G, n_pad, .] (n_pad = largest group rounded up to the 128-row MFMA tile): one stable sort puts each
group's RoIs next to each other, the three adjacency x feature contractions are three launches of the
batched split-bf16 MFMA GEMM (htd_bgemm_nt_counts: tiles and reduction ranges beyond a group's size are skipped on the
device), the four per-level Linear layers run on the same kernels, and the result is scattered back -- a fixed, small
number of launches; with rois_per_img given (the static train path, inference) no host read at all.
"""
import torch

from .. import capi, dense
from ..core.misc import arange_cached, const_tensor

_P, _S = capi.ptr, capi.current_stream_ptr
FUSED_MAX_NPAD = 1024          # one wavefront holds a row of the (n, n) matrices in registers


def local_adjacency(bx, counts):
    """bx (G, npad, 4) group-padded boxes, counts (G,) -> A_local = D^-1/2 M D^-1/2 (G, npad, npad), M = (IoU with unit
    diagonal) > 0 (htd_bbox_head.py:207-210): one launch, no gradient (a function of the boxes only)."""
    G, npad = bx.shape[:2]
    A = torch.empty(G, npad, npad, device=bx.device, dtype=torch.float32)
    dinv = torch.empty(G, npad, device=bx.device, dtype=torch.float32)
    capi.call('htd_pgraph_adjacency', _P(bx.contiguous()), _P(counts), _P(A), _P(dinv), G, npad, _S())
    return A


class _GlobalSoftmax(torch.autograd.Function):
    """A_glob = softmax_row((1 - M) * sim) over each group's valid columns (:211,214-215), M read as A_local > 0."""

    @staticmethod
    def forward(ctx, sim, A_local, counts):
        G, npad = sim.shape[:2]
        sim = sim.contiguous()
        A = torch.empty_like(sim)
        capi.call('htd_pgraph_softmax_fwd', _P(sim), _P(A_local), _P(counts), _P(A), G, npad, _S())
        ctx.save_for_backward(A, A_local, counts)
        return A

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gA):
        A, A_local, counts = ctx.saved_tensors
        G, npad = A.shape[:2]
        gsim = torch.empty_like(A)
        capi.call('htd_pgraph_softmax_bwd', _P(gA.contiguous()), _P(A), _P(A_local), _P(counts), _P(gsim), G, npad, _S())
        return gsim, None, None


class _GatherGroups(torch.autograd.Function):
    """x (N,F) -> the padded group slots: out[i] = valid[i] ? x[rows[i]] : 0 as (G*npad, Fo) rows (Fo >= F, zero tail) or, with
    transposed=True, as (G, F, npad) -- the K-major operand of the adjacency product, written directly (no index_select,
    mask multiply and transpose copy).  Backward is the adjoint scatter: every RoI occupies at most one valid slot, so it is
    plain stores into a zeroed map (htd_pgraph_gather / htd_pgraph_scatter), not an index_add of 8 192 rows."""

    @staticmethod
    def forward(ctx, x, rows, valid, G, Fo, transposed):
        x = x.contiguous()
        N, Fdim = x.shape
        n_out = rows.numel()
        npad = n_out // G
        out = torch.empty((G, Fdim, npad) if transposed else (n_out, Fo), device=x.device, dtype=x.dtype)
        capi.call('htd_pgraph_gather', _P(x), _P(rows), _P(valid), _P(out), n_out, Fdim, Fdim if transposed else Fo, G,
                  int(transposed), _S())
        ctx.save_for_backward(rows, valid)
        ctx.dims = (N, Fdim, Fo, G, bool(transposed))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        rows, valid = ctx.saved_tensors
        N, Fdim, Fo, G, transposed = ctx.dims
        gx = torch.empty(N, Fdim, device=g.device, dtype=g.dtype)
        capi.call('htd_pgraph_scatter', _P(g.contiguous()), _P(rows), _P(valid), _P(gx), rows.numel(), N, Fdim,
                  Fdim if transposed else Fo, G, int(transposed), _S())
        return gx, None, None, None, None, None


def group_layout(rois, target_lvls, num_levels, num_imgs=None, roi_valid=None):
    """Sort RoIs by (level, image): group g = level * B + image, so that the groups of one level -- the rows of that level's
    Linear layer -- are one contiguous slab of the padded batch.  -> perm (N,), counts (G,) device tensor, G = B*num_levels, B.
    With num_imgs given nothing here reads the device (no .item(), no bincount size probe)."""
    img = rois[:, 0].long()
    if num_imgs is None:
        B = int(img.max().item()) + 1 if rois.numel() else 0
    else:
        B = num_imgs
    key = target_lvls * B + img
    if roi_valid is not None:                      # unused sample slots: a group of their own past the real ones
        key = torch.where(roi_valid, key, torch.full_like(key, B * num_levels))
    perm = torch.sort(key, stable=True)[1]
    counts = (key[:, None] == arange_cached(B * num_levels, key.device)[None, :]).sum(0)
    return perm, counts, B


def pgraph_refine(x, sam, rois, target_lvls, graph_layers, rois_per_img=None, roi_valid=None):
    """x (N,F) fc features, sam (N,S) semantic embedding, -> refined (N,F):
         M       = (IoU(rois_g, rois_g) with unit diagonal) > 0
         A_local = D^-1/2 M D^-1/2,  D = rowsum(M)
         mixed   = A_local @ x_g
         A_glob  = softmax_row((1 - M) * (sam_g sam_g^T))
         refined_g = ReLU(Linear_level(A_glob @ mixed))          rows in empty groups stay 0.
    rois_per_img (host ints, optional): RoIs of each image; the padded group size is then bounded by the largest
    image instead of read back from the device, and the whole op runs without a host/device synchronisation."""
    N, Fdim = x.shape
    L = len(graph_layers)
    refined = x.new_zeros(N, Fdim)
    if N == 0:
        return refined
    if rois_per_img is not None:
        perm, counts, B = group_layout(rois, target_lvls, L, len(rois_per_img), roi_valid)
        nmax = max(rois_per_img)
    else:
        perm, counts, B = group_layout(rois, target_lvls, L)
        nmax = max(counts.tolist())                 # the one host read of this op
    G = B * L
    if nmax == 0:
        return refined
    # padded batch index: row r of group g  <-  sorted position start_g + r
    npad = (nmax + 127) // 128 * 128
    starts = torch.cumsum(counts, 0) - counts
    ar = arange_cached(npad, x.device)
    valid = ar[None, :] < counts[:, None]                                   # (G, npad)
    src = (starts[:, None] + ar[None, :]).clamp(max=N - 1)
    rows = perm[src]                                                        # (G, npad) original RoI rows
    vf = valid[..., None].to(x.dtype)
    flat_rows = rows.reshape(-1).contiguous()
    flat_valid = valid.reshape(-1).contiguous()
    S8 = (sam.size(1) + 7) // 8 * 8
    fused_gather = x.is_cuda and x.dtype == torch.float32 and sam.dtype == torch.float32
    if fused_gather:
        sg = _GatherGroups.apply(sam, flat_rows, flat_valid, G, S8, False).view(G, npad, S8)
        xgT = _GatherGroups.apply(x, flat_rows, flat_valid, G, Fdim, True)            # (G, F, npad)
    else:
        xg = torch.index_select(x, 0, flat_rows).view(G, npad, Fdim) * vf
        xgT = xg.transpose(1, 2).contiguous()
        sg = torch.nn.functional.pad(torch.index_select(sam, 0, flat_rows).view(G, npad, -1) * vf, (0, S8 - sam.size(1)))
    bx = torch.index_select(rois, 0, flat_rows).view(G, npad, -1)[..., 1:5]
    counts = counts.contiguous()
    if npad <= FUSED_MAX_NPAD:
        # IoU -> mask -> degree -> normalisation in one kernel; (1 - M) * sim -> row soft-max in another
        A_local = local_adjacency(bx, counts)
        # mixed^T[f][i] = sum_j x^T[f][j] * A_local[i][j]   (A_local @ x, kept transposed: it is the NT operand below)
        # counts + limit bits: the products skip tiles / reduction ranges beyond each group's size on the device (at B = 64 x 512
        # proposals the groups are padded to the largest image: 4.75x the real work otherwise), no host read of the sizes
        mixedT = dense.bgemm_nt(xgT, A_local, counts, 2 | 4)                # (G, F, npad): columns and reduction < count
        A_glob = _GlobalSoftmax.apply(dense.bgemm_nt(sg, sg, counts, 1 | 2), A_local, counts)
    else:       # groups beyond 1024 RoIs (no HTD config gets there): the same arithmetic as tensor expressions
        lt = torch.max(bx[:, :, None, :2], bx[:, None, :, :2])
        rb = torch.min(bx[:, :, None, 2:], bx[:, None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        area = (bx[..., 2] - bx[..., 0]) * (bx[..., 3] - bx[..., 1])
        union = torch.max(area[:, :, None] + area[:, None, :] - inter, const_tensor([1e-6], inter.device, inter.dtype))
        eye = torch.eye(npad, device=x.device, dtype=torch.bool)[None]
        pair = valid[:, :, None] & valid[:, None, :]
        Mloc = (((inter / union > 0) | eye) & pair).to(x.dtype)             # (G, npad, npad), symmetric
        dinv = Mloc.sum(-1).clamp(min=1.0).pow(-0.5)                        # padded rows: avoid 0^-1/2
        A_local = dinv[:, :, None] * Mloc * dinv[:, None, :]
        mixedT = dense.bgemm_nt(xgT, A_local)
        sim = dense.bgemm_nt(sg, sg)                                        # (G, npad, npad)
        logits = (1.0 - Mloc) * sim
        logits = torch.where(pair, logits, logits.new_full((1, ), float('-inf')))   # padded columns carry no mass
        logits = torch.where(valid[:, :, None], logits, torch.zeros_like(logits))   # padded rows: finite, unused
        A_glob = torch.softmax(logits, dim=-1) * vf
    lim = (counts, 1 | 4) if npad <= FUSED_MAX_NPAD else (None, 0)          # rows and reduction < count
    agg = dense.bgemm_nt(A_glob, mixedT, *lim).view(L, B * npad, Fdim)      # A_glob @ mixed; level-major groups
    out = torch.cat([dense.linear(agg[i], layer.weight, layer.bias, relu=True) for i, layer in enumerate(graph_layers)], 0)
    # scatter back; padded rows all land on one extra row that is dropped (no boolean-mask gather = no host sync)
    dst = torch.where(valid, rows, torch.full_like(rows, N)).reshape(-1)
    return x.new_zeros(N + 1, Fdim).index_copy(0, dst, out)[:N]
