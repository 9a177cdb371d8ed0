"""Dense one-hot `bmm` restatement of the reference forward in plain torch — baseline / checker infrastructure, NOT product code
(only tests/, bench.py's cpu_baseline leg and bench_train.py's --dense-baseline may import it).

Same algorithmic shape as src/dynamics/gnn/model.py:129-313: adjacency as two dense one-hot float matrices Rr, Rs (B, E, N),
every gather / scatter a `bmm` with them (2*E*N*F FLOP each), the 450- and 300-wide propagator concatenations materialised.
`module` is anything holding the reference's sub-modules (`particle_encoder.model`, `relation_encoder.model`,
`relation_propagator.linear`, `particle_propagator.linear`, `non_rigid_predictor.linear_{0,1,2}`)."""
import torch
import torch.nn.functional as F


def dense_forward(module, state, attrs, Rr, Rs, p_instance, action, phys, pstep=3, clamp=100.0):
    B, N = attrs.shape[:2]
    n_p = p_instance.shape[1]
    Rr_t = Rr.transpose(1, 2)
    sn = torch.cat([state[:, 1:] - state[:, :-1], state[:, -1:]], 1).transpose(1, 2).reshape(B, N, -1)      # model.py:155-165
    ph = torch.cat([phys[:, None].expand(B, n_p, -1), phys.new_zeros(B, N - n_p, phys.shape[1])], 1)
    p_in = torch.cat([attrs, ph, action], 2)                                                                 # :168-195
    g = torch.cat([p_instance, p_instance.new_zeros(B, N - n_p, p_instance.shape[2])], 1)
    rel = torch.cat([Rr.bmm(attrs), Rs.bmm(attrs), (Rr.bmm(g) - Rs.bmm(g)).abs().sum(2, keepdim=True), Rr.bmm(sn) - Rs.bmm(sn)], 2)   # :220-253
    mlp = lambda blk, x: F.relu(blk.model[4](F.relu(blk.model[2](F.relu(blk.model[0](x))))))
    enc_n, enc_e = mlp(module.particle_encoder, p_in), mlp(module.relation_encoder, rel)                    # :268, :274
    h = enc_n
    for _ in range(pstep):                                                                                   # :283-301
        eff = F.relu(module.relation_propagator.linear(torch.cat([enc_e, Rr.bmm(h), Rs.bmm(h)], 2)))
        h = F.relu(module.particle_propagator.linear(torch.cat([enc_n, Rr_t.bmm(eff)], 2)) + h)
    d = module.non_rigid_predictor
    m = d.linear_2(F.relu(d.linear_1(F.relu(d.linear_0(h[:, :n_p])))))                                      # :306
    return state[:, -1, :n_p] + m.clamp(-clamp, clamp), m                                                    # :309


def one_hots(n_rel, recv, send, N, dtype=torch.float32):
    """Per-sample edge lists (oracle / golden layout) -> dense Rr, Rs (B, max n_rel, N), zero-padded like pad_torch (utils.py:37-46)."""
    B, E = len(n_rel), int(max(int(n) for n in n_rel)) if len(n_rel) else 0
    Rr, Rs = torch.zeros(B, max(E, 1), N, dtype=dtype), torch.zeros(B, max(E, 1), N, dtype=dtype)
    for b in range(B):
        n = int(n_rel[b])
        if n:
            idx = torch.arange(n)
            Rr[b, idx, torch.as_tensor(recv[b, :n].astype("int64"))] = 1
            Rs[b, idx, torch.as_tensor(send[b, :n].astype("int64"))] = 1
    return Rr, Rs
