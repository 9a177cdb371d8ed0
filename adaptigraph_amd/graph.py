"""Radius-graph / top-k adjacency on the HIP engine.

Drop-in for src/dynamics/dataset/graph.py: `construct_edges_from_states(...)` (:38-89) and
`construct_edges_from_states_batch(...)` (:91-156) keep the reference's signatures and dense
one-hot `(Rr, Rs)` return value; `build_edges(...)` is the native fast path returning the CSR
adjacency the kernels consume (no O(E*N) one-hots, no host sync).
"""
import ctypes

import numpy as np
import torch

from . import _lib

_WS = {}


def workspace(device, nbytes):
    """Grow-only scratch buffer handed to the C ABI (the library never allocates scratch), one per (device, stream):
    calls enqueued on different streams may overlap, so they must not share scratch."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"adaptigraph_amd: `{name}` must live on an MI355X (got {t.device}); "
                           "the engine has no CPU path")


class CSREdges:
    """Receiver-sorted adjacency of a batch of graphs, as produced by ag_build_edges.

    row_ptr (B*N+1,) int32 over global rows b*N+i; edge_recv / edge_send (e_cap,) int32 global node ids;
    only the first row_ptr[-1] entries are valid.  Edge order == the reference's nonzero() order.
    """

    def __init__(self, row_ptr, edge_recv, edge_send, B, N, e_cap):
        self.row_ptr, self.edge_recv, self.edge_send = row_ptr, edge_recv, edge_send
        self.B, self.N, self.e_cap = B, N, e_cap

    def n_rel(self):
        """(B,) int64 edge count per sample (device tensor)."""
        rp = self.row_ptr.long()
        return rp[self.N::self.N][:self.B] - rp[0:self.B * self.N:self.N]

    def to_lists(self):
        """[(recv_local, send_local) numpy int32 arrays] per sample (host sync; tests / debugging)."""
        rp = self.row_ptr.cpu().numpy()
        r = self.edge_recv.cpu().numpy()
        s = self.edge_send.cpu().numpy()
        out = []
        for b in range(self.B):
            lo, hi = rp[b * self.N], rp[(b + 1) * self.N]
            out.append((r[lo:hi] - b * self.N, s[lo:hi] - b * self.N))
        return out

    def to_dense(self, dtype=torch.float32):
        """(Rr, Rs) one-hot (B, max n_rel, N) exactly as graph.py:146-155 lays them out (host sync)."""
        n = self.n_rel()
        e_max = int(n.max().item()) if self.B else 0
        total = int(self.row_ptr[-1].item())
        dev = self.row_ptr.device
        Rr = torch.zeros((self.B, e_max, self.N), dtype=dtype, device=dev)
        Rs = torch.zeros((self.B, e_max, self.N), dtype=dtype, device=dev)
        if total:
            r = self.edge_recv[:total].long()
            s = self.edge_send[:total].long()
            b = r // self.N
            start = self.row_ptr.long()[b * self.N]
            idx = torch.arange(total, device=dev) - start
            Rr[b, idx, r - b * self.N] = 1
            Rs[b, idx, s - b * self.N] = 1
        return Rr, Rs


def threshold_sq(adj_thresh, B, device, variant):
    """Squared radius per sample, rounded the way the chosen builder variant rounds it:
    single: Python double r*r, cast to fp32 by the tensor-scalar subtraction (graph.py:53,68);
    batch:  fp32 tensor r * r (graph.py:106-108)."""
    if torch.is_tensor(adj_thresh):
        t = adj_thresh.to(device=device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.repeat(B)
        return (t * t).contiguous()
    if isinstance(adj_thresh, (list, tuple, np.ndarray)):     # one host double per sample (e.g. radii drawn by a dataset)
        r = np.asarray(adj_thresh, np.float64).reshape(-1)
        assert r.size == B, f"{r.size} radii for {B} samples"
        v = (r * r).astype(np.float32) if variant == _lib.AG_VARIANT_SINGLE else r.astype(np.float32) * r.astype(np.float32)
        return torch.from_numpy(np.ascontiguousarray(v)).to(device)
    r = float(adj_thresh)
    if variant == _lib.AG_VARIANT_SINGLE:
        v = np.float32(r * r)
    else:
        v = np.float32(r) * np.float32(r)
    return torch.full((B,), float(v), dtype=torch.float32, device=device)


def build_edges(states, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False, variant="batch",
                max_tools=None):
    """states (B,N,3) fp32 cuda; mask, tool_mask (B,N) bool -> CSREdges.  No host synchronisation when
    `max_tools` is given (default: N, always sufficient)."""
    _require_gpu(states, "states")
    L = _lib.lib()
    B, N, _ = states.shape
    dev = states.device
    var = _lib.AG_VARIANT_SINGLE if variant == "single" else _lib.AG_VARIANT_BATCH
    states = states.contiguous().float()
    mask_u8 = mask.to(device=dev).contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.to(dev, torch.uint8).contiguous()
    tool_u8 = tool_mask.to(device=dev).contiguous().view(torch.uint8) if tool_mask.dtype == torch.bool else tool_mask.to(dev, torch.uint8).contiguous()
    thr = threshold_sq(adj_thresh, B, dev, var)
    if max_tools is None:
        max_tools = N
    connect = 1 if connect_tools_all else 0
    e_cap = int(L.ag_edge_capacity(B, N, int(topk), connect, int(max_tools)))
    row_ptr = torch.empty(B * N + 1, dtype=torch.int32, device=dev)
    edge_recv = torch.empty(max(e_cap, 1), dtype=torch.int32, device=dev)
    edge_send = torch.empty(max(e_cap, 1), dtype=torch.int32, device=dev)
    nbytes = L.ag_edges_workspace_bytes(B, N, int(topk), connect, int(max_tools))
    ws = workspace(dev, nbytes)
    with torch.cuda.device(dev):
        rc = L.ag_build_edges(states.data_ptr(), mask_u8.data_ptr(), tool_u8.data_ptr(), thr.data_ptr(), int(topk),
                              connect, var, B, N, int(max_tools), row_ptr.data_ptr(), edge_recv.data_ptr(),
                              edge_send.data_ptr(), e_cap, ws.data_ptr(), ws.numel(), _stream_ptr(dev))
    _lib.check(rc, "ag_build_edges")
    return CSREdges(row_ptr, edge_recv, edge_send, B, N, e_cap)


def construct_edges_from_states(states, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False):
    """Drop-in for graph.py:38-89: states (N,3) -> (Rr, Rs) of shape (n_rel, N)."""
    n_tools = int(tool_mask.sum().item())
    csr = build_edges(states[None], adj_thresh, mask[None], tool_mask[None], topk, connect_tools_all, "single",
                      max_tools=n_tools)
    Rr, Rs = csr.to_dense(states.dtype)
    return Rr[0], Rs[0]


def construct_edges_from_states_batch(states, adj_thresh, mask, tool_mask, topk=10, connect_tools_all=False):
    """Drop-in for graph.py:91-156: states (B,N,3) -> (Rr, Rs) of shape (B, max n_rel, N)."""
    n_tools = int(tool_mask.sum(1).max().item())
    csr = build_edges(states, adj_thresh, mask, tool_mask, topk, connect_tools_all, "batch", max_tools=n_tools)
    return csr.to_dense(states.dtype)


def csr_from_dense(Rr, Rs):
    """One-hot (B,E,N) pair -> CSREdges (compat path of DynamicsPredictor.forward; host sync).
    All-zero (padded) rows are dropped, which is exact (truncate_graph / pad_torch vanish);
    edges are stably re-sorted by receiver because the kernels reduce over CSR rows."""
    B, E, N = Rr.shape
    dev = Rr.device
    valid = (Rr.sum(-1) > 0) & (Rs.sum(-1) > 0)
    b_idx = torch.arange(B, device=dev)[:, None].expand(B, E)
    recv = (Rr.argmax(-1) + b_idx * N)[valid]
    send = (Rs.argmax(-1) + b_idx * N)[valid]
    order = torch.sort(recv, stable=True).indices
    recv, send = recv[order], send[order]
    counts = torch.bincount(recv, minlength=B * N)
    row_ptr = torch.zeros(B * N + 1, dtype=torch.int32, device=dev)
    row_ptr[1:] = torch.cumsum(counts, 0).int()
    total = int(recv.numel())
    return CSREdges(row_ptr, recv.int().contiguous() if total else torch.zeros(1, dtype=torch.int32, device=dev),
                    send.int().contiguous() if total else torch.zeros(1, dtype=torch.int32, device=dev), B, N,
                    max(total, 1) if total else 0)
