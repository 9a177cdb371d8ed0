"""Differentiable graph ops of the training path on the HIP kernels of csrc/ag_train.hip — SURVEY.md §8f row n4.

`gather_rows` and `message_sum` replace the one-hot `Rr.bmm / Rs.bmm / Rr_t.bmm` of src/dynamics/gnn/model.py:220-295;
their backward passes are segment sums over a (pointer, permutation) view of the same edge list, so gradients are
reproducible bit for bit (the reference's dense bmm autograd is too; index_add-style atomics would not be).
"""
import torch

from . import _lib
from .graph import CSREdges, _require_gpu, _stream_ptr


class EdgeViews:
    """Receiver- and sender-sorted views of one batch's edges.  One host read (the edge count) per batch; the reference
    training loop keeps the edges of a batch fixed across its n_future unroll (train.py:90-108), so this is per batch."""

    def __init__(self, csr: CSREdges):
        self.M = csr.B * csr.N
        self.row_ptr = csr.row_ptr.contiguous()
        self.E = int(self.row_ptr[-1].item())
        self.recv = csr.edge_recv[: self.E].contiguous()
        self.send = csr.edge_send[: self.E].contiguous()
        send64 = self.send.long()
        self.send_perm = torch.sort(send64, stable=True).indices.to(torch.int32).contiguous()
        counts = torch.bincount(send64, minlength=self.M)
        self.col_ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32).contiguous()


def _segment_sum(vals, ptr, perm, n_seg):
    out = torch.empty((n_seg, vals.shape[1]), dtype=torch.float32, device=vals.device)
    with torch.cuda.device(vals.device):
        rc = _lib.lib().ag_segment_sum(vals.data_ptr(), ptr.data_ptr(), perm.data_ptr() if perm is not None else None,
                                       out.data_ptr(), n_seg, vals.shape[1], _stream_ptr(vals.device))
    _lib.check(rc, "ag_segment_sum")
    return out


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, ptr, perm):
        _require_gpu(x, "x")
        x = x.contiguous().float()
        out = torch.empty((idx.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.lib().ag_gather_rows(x.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.shape[0], x.shape[1], _stream_ptr(x.device))
        _lib.check(rc, "ag_gather_rows")
        ctx.views = (ptr, perm, x.shape[0])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ptr, perm, n = ctx.views
        return _segment_sum(grad_out.contiguous().float(), ptr, perm, n), None, None, None


class _MessageSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eterm, hr, hs, views):
        eterm, hr, hs = eterm.contiguous().float(), hr.contiguous().float(), hs.contiguous().float()
        _require_gpu(hr, "hr")
        agg = torch.empty_like(hr)
        with torch.cuda.device(hr.device):
            rc = _lib.lib().ag_message_forward(eterm.data_ptr(), hr.data_ptr(), hs.data_ptr(), views.row_ptr.data_ptr(),
                                               views.send.data_ptr(), agg.data_ptr(), hr.shape[0], hr.shape[1], _stream_ptr(hr.device))
        _lib.check(rc, "ag_message_forward")
        ctx.save_for_backward(eterm, hr, hs)
        ctx.views = views
        return agg

    @staticmethod
    def backward(ctx, grad_agg):
        eterm, hr, hs = ctx.saved_tensors
        v = ctx.views
        grad_agg = grad_agg.contiguous().float()
        g_e = torch.empty_like(eterm)
        g_hr = torch.empty_like(hr)
        with torch.cuda.device(hr.device):
            rc = _lib.lib().ag_message_backward(eterm.data_ptr(), hr.data_ptr(), hs.data_ptr(), v.row_ptr.data_ptr(), v.send.data_ptr(),
                                                grad_agg.data_ptr(), g_e.data_ptr(), g_hr.data_ptr(), hr.shape[0], hr.shape[1],
                                                _stream_ptr(hr.device))
        _lib.check(rc, "ag_message_backward")
        g_hs = _segment_sum(g_e, v.col_ptr, v.send_perm, hs.shape[0])
        return g_e, g_hr, g_hs, None


def gather_receivers(x, views):
    """x (M,D) -> (E,D) rows of each edge's receiver."""
    if views.E == 0:
        return x[:0] * 1.0
    return _GatherRows.apply(x, views.recv, views.row_ptr, None)


def gather_senders(x, views):
    """x (M,D) -> (E,D) rows of each edge's sender."""
    if views.E == 0:
        return x[:0] * 1.0
    return _GatherRows.apply(x, views.send, views.col_ptr, views.send_perm)


def message_sum(eterm, hr, hs, views):
    """(E,D), (M,D), (M,D) -> (M,D): sum over each receiver's edges of relu(eterm[e] + hr[recv] + hs[send])."""
    if views.E == 0:      # a batch without edges (the reference's dense bmm handles it: model.py:295 on an empty Rr)
        return hr.sum() * 0 + hs.sum() * 0 + torch.zeros_like(hr)
    return _MessageSum.apply(eterm, hr, hs, views)
