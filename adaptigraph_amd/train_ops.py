"""Differentiable graph ops of the training path on the HIP kernels of csrc/ag_train.hip — SURVEY.md §8f row n4.

`gather_rows` and `message_sum` replace the one-hot `Rr.bmm / Rs.bmm / Rr_t.bmm` of src/dynamics/gnn/model.py:220-295;
their backward passes are segment sums over a (pointer, permutation) view of the same edge list, so gradients are
reproducible bit for bit (the reference's dense bmm autograd is too; index_add-style atomics would not be).
"""
import contextlib
import ctypes
import weakref

import torch

from . import _lib
from .graph import CSREdges, _require_gpu, _stream_ptr, workspace


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


class _EdgeInputs(torch.autograd.Function):
    """rel_inputs (model.py:220-253) from the per-node table [attrs | group | state_norm] in one kernel; backward in two."""

    @staticmethod
    def forward(ctx, tab, views, attr_dim, group_dim):
        _require_gpu(tab, "tab")
        tab = tab.contiguous().float()
        D = tab.shape[1]
        out = torch.empty((views.E, 2 * attr_dim + 1 + (D - attr_dim - group_dim)), dtype=torch.float32, device=tab.device)
        with torch.cuda.device(tab.device):
            rc = _lib.lib().ag_edge_inputs_forward(tab.data_ptr(), D, attr_dim, group_dim, views.recv.data_ptr(), views.send.data_ptr(),
                                                   out.data_ptr(), views.E, _stream_ptr(tab.device))
        _lib.check(rc, "ag_edge_inputs_forward")
        ctx.save_for_backward(tab)
        ctx.meta = (views, attr_dim, group_dim)
        return out

    @staticmethod
    def backward(ctx, g):
        (tab,) = ctx.saved_tensors
        v, a, gdim = ctx.meta
        g = g.contiguous()
        scratch = torch.empty((2, max(v.E, 1), tab.shape[1]), dtype=torch.float32, device=tab.device)
        gtab = torch.empty_like(tab)
        with torch.cuda.device(tab.device):
            rc = _lib.lib().ag_edge_inputs_backward(tab.data_ptr(), tab.shape[1], a, gdim, v.recv.data_ptr(), v.send.data_ptr(), v.row_ptr.data_ptr(),
                                                    v.col_ptr.data_ptr(), v.send_perm.data_ptr(), g.data_ptr(), scratch[0].data_ptr(),
                                                    scratch[1].data_ptr(), gtab.data_ptr(), v.E, tab.shape[0], _stream_ptr(tab.device))
        _lib.check(rc, "ag_edge_inputs_backward")
        return gtab, None, None, None


def edge_inputs(tab, views, attr_dim, group_dim):
    """tab (M, A + G + S) = [attrs | group | state_norm] -> rel_inputs (E, 2A + 1 + S) as DynamicsPredictor.forward builds them."""
    return _EdgeInputs.apply(tab, views, attr_dim, group_dim)


def message_sum(eterm, hr, hs, views):
    """(E,D), (M,D), (M,D) -> (M,D): sum over each receiver's edges of relu(eterm[e] + hr[recv] + hs[send])."""
    if views.E == 0:      # a batch without edges (the reference's dense bmm handles it: model.py:295 on an empty Rr)
        return hr.sum() * 0 + hs.sum() * 0 + torch.zeros_like(hr)
    return _MessageSum.apply(eterm, hr, hs, views)


# ---------------------------------------------------------------------------------------------------------------------
# Dense stacks on the fused MFMA kernels (csrc/ag_mlp.hip: chain_forward_kernel / chain_backward_kernel)
# ---------------------------------------------------------------------------------------------------------------------
AG_FP, _ROW_TILE, _CHUNK = 160, 128, 5120
CHAIN_KINDS = {"edge": (0, 4), "node": (1, 3), "decoder": (2, 3)}      # name -> (AG_CHAIN_*, layers)
CHAIN_PRECISION = 1     # 1: split-bf16 MFMA (x = hi + lo, three bf16 products, fp32 accumulate; default); 0: exact fp32 MFMA
_PACK_CACHE = {}


def invalidate_packs():
    """Drop every cached weight pack.  Needed only after writes the version counter cannot see: `p.data` mutation (legacy
    optimisers, EMA swaps) or a HIP-graph replay that updated the parameters (bench_train --graph)."""
    _PACK_CACHE.clear()


def _pack_chain(kind, layers, dev):
    """Pack [(W, b)] into the forward stream and the transposed backward stream (device-side, one launch per layer), cached
    until a parameter is updated in place (optimiser step) or replaced.  A cache entry is valid only for the SAME tensor
    objects (weak references: a freed model's allocator blocks and version counters are routinely re-used by the next model,
    so (data_ptr, _version) alone would hand it the old pack) at the same versions; `.data` writes bypass the version counter
    and need invalidate_packs()."""
    prec = int(CHAIN_PRECISION)
    base = lambda t: t._base if t._base is not None else t          # views (column slices of W_rp) share their base's version counter
    key = (kind, dev.index, prec) + tuple((W.data_ptr(), W._version, b.data_ptr(), b._version) for W, b in layers)
    hit = _PACK_CACHE.get((kind, dev.index))
    if hit is not None and hit[0] == key and all(r() is base(x) for r, x in zip(hit[3], (x for Wb in layers for x in Wb))):
        return hit[1], hit[2], prec
    L = _lib.lib()
    narrow = kind != "decoder"
    n = len(layers)
    fwd = torch.empty(((1 if narrow else 5) + 5 * (n - 1)) * _CHUNK, dtype=torch.float32, device=dev)
    bwd = torch.empty((5 * (n - 1) + (1 if narrow else 5)) * _CHUNK, dtype=torch.float32, device=dev)
    st = _stream_ptr(dev)
    with torch.cuda.device(dev):
        off = 0
        for l, (W, b) in enumerate(layers):
            n_out, n_in = W.shape
            assert W.stride(1) == 1 and b.is_contiguous()
            compact = narrow and l == 0
            _lib.check(L.ag_train_pack(W.data_ptr(), b.data_ptr(), n_out, n_in, W.stride(0), 0, 0, int(compact), 1 if compact else 5, prec,
                                       fwd.data_ptr() + 4 * off, st), "ag_train_pack")
            off += _CHUNK * (1 if compact else 5)
        off = 0
        for l in range(n - 1, -1, -1):
            W, _ = layers[l]
            n_out, n_in = W.shape
            tiles = 1 if (narrow and l == 0) else 5
            _lib.check(L.ag_train_pack(W.data_ptr(), None, n_in, n_out, W.stride(0), 0, 1, 0, tiles, prec, bwd.data_ptr() + 4 * off, st), "ag_train_pack")
            off += _CHUNK * tiles
    _PACK_CACHE[(kind, dev.index)] = (key, fwd, bwd, [weakref.ref(base(x)) for Wb in layers for x in Wb])
    return fwd, bwd, prec


def _ptr_array(tensors):
    return (ctypes.c_void_p * 4)(*([t.data_ptr() for t in tensors] + [None] * (4 - len(tensors))))


_ZEROS = {}


def _zero_padded(tag, rows_pad, dev):
    """A (rows_pad, 160) fp32 buffer whose every element outside the block a caller overwrites is zero.  One per (tag, shape):
    callers overwrite the SAME live block each time and consume the buffer on the same stream before the next use."""
    key = (tag, rows_pad, dev.index)
    buf = _ZEROS.get(key)
    if buf is None:
        buf = torch.zeros((rows_pad, AG_FP), dtype=torch.float32, device=dev)
        _ZEROS[key] = buf
    return buf


class _FusedChain(torch.autograd.Function):
    """y = layer_{L-1}(... layer_0(x)) with the kind's fixed widths and ReLU pattern; saves every layer output, backward =
    one fused kernel for the input / pre-activation gradients + one library GEMM and one column sum per layer."""

    @staticmethod
    def forward(ctx, kind, x, *params):
        _require_gpu(x, "x")
        code, n = CHAIN_KINDS[kind]
        layers = [(params[2 * l], params[2 * l + 1]) for l in range(n)]
        dev = x.device
        rows = x.shape[0]
        rows_pad = max(_ROW_TILE, -(-rows // _ROW_TILE) * _ROW_TILE)
        fwd, bwd, prec = _pack_chain(kind, layers, dev)             # (keyed on the parameter objects themselves, see _pack_chain)
        if kind == "decoder":
            xin = x.new_zeros((rows_pad, AG_FP))                    # saved for the backward: a fresh table per call
            xin[:rows, : x.shape[1]] = x
        else:
            xin = x.contiguous().float()
        ys = [torch.empty((rows_pad, AG_FP), dtype=torch.float32, device=dev) for _ in range(n)]
        with torch.cuda.device(dev):
            rc = _lib.lib().ag_train_chain(code, 0, prec, xin.data_ptr(), fwd.data_ptr(), _ptr_array(ys), None, _ptr_array([]), None, rows,
                                           x.shape[1], _stream_ptr(dev))
        _lib.check(rc, "ag_train_chain(forward)")
        ctx.kind, ctx.rows, ctx.d_in, ctx.bwd, ctx.prec = kind, rows, x.shape[1], bwd, prec
        ctx.params = list(layers) if DIRECT_GRADS else None
        ctx.shapes = [tuple(W.shape) for W, _ in layers]
        ctx.save_for_backward(xin, *ys)
        return ys[-1][:rows, : layers[-1][0].shape[0]]

    @staticmethod
    def backward(ctx, grad_out):
        xin, *ys = ctx.saved_tensors
        code, n = CHAIN_KINDS[ctx.kind]
        dev, rows = xin.device, ctx.rows
        rows_pad = ys[0].shape[0]
        dy = _zero_padded("dy_" + ctx.kind, rows_pad, dev)                  # columns >= n_out and padding rows stay zero: only the live block is written
        dy[:rows, : grad_out.shape[1]] = grad_out
        dzs = [torch.empty((rows_pad, AG_FP), dtype=torch.float32, device=dev) for _ in range(n)]
        dx = torch.empty_like(xin)
        with torch.cuda.device(dev):
            rc = _lib.lib().ag_train_chain(code, 1, ctx.prec, xin.data_ptr(), ctx.bwd.data_ptr(), _ptr_array(ys), dy.data_ptr(), _ptr_array(dzs),
                                           dx.data_ptr(), rows, ctx.d_in, _stream_ptr(dev))
        _lib.check(rc, "ag_train_chain(backward)")
        # dW_l = dz_l^T y_{l-1}, db_l = column sums of dz_l: all layers in two launches (row-slab split-K on the fp32 MFMA,
        # fixed-order reduction) — a 150 x 150 output over 10^4..10^5 rows runs on 25 workgroups as a library GEMM
        out = weight_grads(dzs, [xin] + ys[:-1], [s[1] for s in ctx.shapes], rows, ctx.params)
        if out is None:
            return (None, dx[:rows, : ctx.d_in]) + (None,) * (2 * n)
        grads = []
        for l, (n_out, n_in) in enumerate(ctx.shapes):
            grads += [out[l, :n_out, :n_in], out[l, :n_out, n_in]]
        return (None, dx[:rows, : ctx.d_in]) + tuple(grads)


DIRECT_GRADS = False    # True: weight / bias gradients are ACCUMULATED straight into the leaf parameters' .grad by the gradient kernel
                        # and the autograd Functions return None for them (no per-parameter slice-backward / clone / add kernels:
                        # ~200 of the ~620 launches of a training step).  Set by train.train() and bench_train.py; code that asks
                        # autograd for parameter gradients (torch.autograd.grad) must leave it False.


@contextlib.contextmanager
def direct_grads(enabled=True):
    """Scope DIRECT_GRADS to one forward + backward: `with train_ops.direct_grads(): loss = ...; loss.backward()`.  The flag is
    read when a Function's forward runs and captured in its ctx, so backward passes started inside the block behave as built."""
    global DIRECT_GRADS
    prev, DIRECT_GRADS = DIRECT_GRADS, bool(enabled)
    try:
        yield
    finally:
        DIRECT_GRADS = prev


def _wants_grad(t):
    base = t._base if t._base is not None else t
    return base.requires_grad


def _grad_slot(t):
    """(pointer, row stride) of the .grad storage that corresponds to parameter tensor `t` — a leaf, or a column / row slice of
    one (e.g. relation_propagator.linear.weight[:, :nf]); allocates a zero .grad on first use."""
    base = t._base if t._base is not None else t
    assert base.is_leaf and base.requires_grad and base.is_contiguous(), "DIRECT_GRADS needs leaf parameters (or plain slices of them)"
    if base.grad is None:
        base.grad = torch.zeros_like(base)
    off = (t.data_ptr() - base.data_ptr()) // 4
    return base.grad.data_ptr() + 4 * off, (base.grad.stride(0) if base.dim() == 2 else 0)


def weight_grads(dzs, prevs, n_ins, rows, params=None):
    """[dz_l (rows+, ld <= 160)], [prev_l (rows+, ld)], [n_in_l] -> (n, 160, 160): out[l, o, k] = sum_rows dz_l[row, o] prev_l[row, k] for
    k < n_in_l and out[l, o, n_in_l] = sum_rows dz_l[row, o]: weight and bias gradients of up to 4 layers in two launches.
    With `params` = [(W_l, b_l or None)] (DIRECT_GRADS) the sums are accumulated into W_l.grad / b_l.grad instead and None is returned."""
    n, dev = len(dzs), dzs[0].device
    assert 1 <= n <= 4 and all(t.stride(1) == 1 for t in list(dzs) + list(prevs))
    L = _lib.lib()
    ws = workspace(dev, L.ag_train_weight_grads_workspace_bytes(rows, n))
    i32 = lambda v: (ctypes.c_int32 * 4)(*(list(v) + [0] * (4 - n)))
    if params is None:
        out = torch.empty((n, AG_FP, AG_FP), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = L.ag_train_weight_grads(n, _ptr_array(dzs), i32(t.stride(0) for t in dzs), _ptr_array(prevs), i32(t.stride(0) for t in prevs),
                                         i32(n_ins), rows, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        _lib.check(rc, "ag_train_weight_grads")
        return out
    keep = [l for l, (W, _) in enumerate(params) if _wants_grad(W)]          # frozen layers (requires_grad False) get no gradient
    if len(keep) < n:
        if keep:
            weight_grads([dzs[l] for l in keep], [prevs[l] for l in keep], [n_ins[l] for l in keep], rows, [params[l] for l in keep])
        return None
    slots = [(_grad_slot(W), _grad_slot(b)[0] if (b is not None and _wants_grad(b)) else None, W.shape[0]) for W, b in params]
    pv = lambda v: (ctypes.c_void_p * 4)(*(list(v) + [None] * (4 - n)))
    with torch.cuda.device(dev):
        rc = L.ag_train_weight_grads_into(n, _ptr_array(dzs), i32(t.stride(0) for t in dzs), _ptr_array(prevs), i32(t.stride(0) for t in prevs),
                                          i32(n_ins), rows, None, pv(s[0][0] for s in slots), i32(s[0][1] for s in slots), pv(s[1] for s in slots),
                                          i32(s[2] for s in slots), ws.data_ptr(), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "ag_train_weight_grads_into")
    return None


# Deferred weight gradients (DIRECT_GRADS only).  The node-level linears of the three propagation rounds each need a (150 x 150)
# gradient over the same ~10^4 rows; one call per layer is latency-bound (24-66 us for 5 us of MFMA time).  Nothing reads a
# parameter's .grad before backward() returns, so their (dz, input) pairs are queued and computed four layers per launch —
# when a group fills up and, for the remainder, from a callback the autograd engine runs at the end of the backward pass.
_PENDING, _PENDING_ARMED = [], [False]


def _flush_pending(final=True):
    by_rows = {}
    for item in _PENDING:
        by_rows.setdefault((item[3], item[0].device, item[5]), []).append(item)
    _PENDING.clear()
    for (rows, _, stream), items in by_rows.items():
        while items and (final or len(items) >= 8):
            # up to four layers per launch, each with its OWN destination: two layers of one launch accumulating into the same
            # parameter (the same weight in two propagation rounds) would race on its .grad
            grp, rest, seen = [], [], set()
            for it in items:
                key = it[4][0].data_ptr()
                if len(grp) < 4 and key not in seen:
                    grp.append(it)
                    seen.add(key)
                else:
                    rest.append(it)
            items = rest
            # on the stream the backward ran on when the pair was queued (the engine's final callback may run on a thread whose current
            # stream is another one; switching is skipped in the common case, the step is host-bound)
            same = stream == torch.cuda.current_stream(stream.device)
            with (contextlib.nullcontext() if same else torch.cuda.stream(stream)):
                weight_grads([i[0] for i in grp], [i[1] for i in grp], [i[2] for i in grp], rows, [i[4] for i in grp])
        _PENDING.extend(items)
    if final:
        _PENDING_ARMED[0] = False


def reset_pending():
    """Drop queued weight-gradient work (a backward pass that raised leaves its queue behind; unrolled_loss calls this)."""
    _PENDING.clear()
    _PENDING_ARMED[0] = False


def _defer_weight_grads(dz, prev, n_in, rows, params):
    """Queue layers [(dz, prev, n_in, (W, b))] sharing `rows`; call only from inside a backward pass."""
    stream = torch.cuda.current_stream(dz[0].device)
    for z, p, k, wb in zip(dz, prev, n_in, params):
        _PENDING.append((z, p, k, rows, wb, stream))
    if not _PENDING_ARMED[0]:
        _PENDING_ARMED[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_flush_pending)
    _flush_pending(final=False)


class _Linear(torch.autograd.Function):
    """y = x W^T (+ b) as a library GEMM, with the weight / bias gradient on the split-K MFMA kernel: for the node-level
    linears (150 x 150 outputs contracted over ~10^4 rows) hipBLASLt picks a 25-workgroup kernel (75 us per call)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.params = [(W, b)] if DIRECT_GRADS else None
        ctx.has_bias = b is not None
        return torch.nn.functional.linear(x, W, b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous()
        xs = x if x.stride(1) == 1 else x.contiguous()
        if ctx.params is not None:
            _defer_weight_grads([g], [xs], [W.shape[1]], x.shape[0], ctx.params)
            return g @ W, None, None
        out = weight_grads([g], [xs], [W.shape[1]], x.shape[0])
        return g @ W, out[0, : W.shape[0], : W.shape[1]], (out[0, : W.shape[0], W.shape[1]] if ctx.has_bias else None)


class _Linear2(torch.autograd.Function):
    """(x W1^T, x W2^T): the receiver / sender blocks of relation_propagator applied at node level.  One
    backward for both: their weight gradients in ONE split-K call, the input gradient as two accumulating library GEMMs."""

    @staticmethod
    def forward(ctx, x, W1, W2):
        ctx.save_for_backward(x, W1, W2)
        ctx.params = [(W1, None), (W2, None)] if DIRECT_GRADS else None
        return torch.nn.functional.linear(x, W1), torch.nn.functional.linear(x, W2)

    @staticmethod
    def backward(ctx, g1, g2):
        x, W1, W2 = ctx.saved_tensors
        g1, g2 = g1.contiguous(), g2.contiguous()
        xs = x if x.stride(1) == 1 else x.contiguous()
        dx = torch.addmm(g1 @ W1, g2, W2)
        if ctx.params is not None:
            _defer_weight_grads([g1, g2], [xs, xs], [W1.shape[1], W2.shape[1]], x.shape[0], ctx.params)
            return dx, None, None
        out = weight_grads([g1, g2], [xs, xs], [W1.shape[1], W2.shape[1]], x.shape[0])
        return dx, out[0, : W1.shape[0], : W1.shape[1]], out[1, : W2.shape[0], : W2.shape[1]]


def linear2(x, W1, W2):
    return _Linear2.apply(x, W1, W2)


class _Add3Relu(torch.autograd.Function):
    """relu(a + b + c) with one kernel each way (the three inputs share the gradient g * [y > 0])."""

    @staticmethod
    def forward(ctx, a, b, c):
        a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
        y = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.check(_lib.lib().ag_add3_relu(a.data_ptr(), b.data_ptr(), c.data_ptr(), y.data_ptr(), a.numel(), _stream_ptr(a.device)), "ag_add3_relu")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(y)
        with torch.cuda.device(y.device):
            _lib.check(_lib.lib().ag_relu_mask(g.data_ptr(), y.data_ptr(), out.data_ptr(), y.numel(), _stream_ptr(y.device)), "ag_relu_mask")
        return out, out, out


def add3_relu(a, b, c):
    """relu((a + b) + c) for same-shape float tensors whose element count is a multiple of 4."""
    return _Add3Relu.apply(a, b, c)


def linear(x, W, b=None):
    """F.linear for (rows, <= 150) x (<= 150, <= 150) with rows >> 150 (see _Linear)."""
    return _Linear.apply(x, W, b)


def fused_chain(kind, x, layers):
    """kind in CHAIN_KINDS; x (rows, d_in); layers = [(weight, bias)] in order (a weight may be a column slice of a larger
    parameter, e.g. relation_propagator.linear.weight[:, :nf]).  -> (rows, n_out of the last layer)."""
    flat = [t for wb in layers for t in wb]
    return _FusedChain.apply(kind, x, *flat)
