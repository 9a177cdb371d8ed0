"""Differentiable DynamicsPredictor for training on MI355X — SURVEY.md §8f row n4 (src/dynamics/gnn/model.py:129-313,
src/dynamics/train/train.py:66-130).

Same parameters, state_dict and call signature as the inference `DynamicsPredictor` (a checkpoint trained here loads into
the fused engine unchanged), but `forward` is built from autograd-capable pieces: the three Linear(+ReLU) stacks
(relation_encoder + the edge block of relation_propagator, particle_encoder, non_rigid_predictor: > 95 % of the dense FLOPs)
run forward AND backward as fused MFMA chain kernels — split-bf16 arithmetic by default (`train_ops.CHAIN_PRECISION` = 1: every
fp32 operand as hi + lo bf16, three products, fp32 accumulate, gradients held to 1e-4 of fp64; 0 selects exact fp32 MFMA, held
to 2e-5) — (`train_ops.fused_chain`: activations stay in registers
between layers, weight gradients are one library GEMM per layer over the saved tables), the per-round node-level linears are
library GEMMs (torch / hipBLASLt), and the graph part is the HIP gather / segment-reduce kernels of `train_ops` on the CSR
adjacency — the reference's one-hot `bmm`s never exist.  The relation propagator runs in its column-split form
(`W_rp = [W_e | W_r | W_s]`), which is the same function of the same 22 tensors, so autograd reaches the
original `relation_propagator.linear.weight` through its three slices.
"""
import torch
import torch.nn.functional as F

from .graph import CSREdges, csr_from_dense
from .model import DynamicsPredictor
from .train_ops import EdgeViews, add3_relu, edge_inputs, fused_chain, gather_receivers, gather_senders, linear, linear2, message_sum, reset_pending


def _mlp3(block, x):
    for i in (0, 2, 4):
        x = F.relu(F.linear(x, block.model[i].weight, block.model[i].bias))
    return x


class TrainableDynamicsPredictor(DynamicsPredictor):
    fused_dense = True     # encoder / decoder stacks on the fused MFMA chain kernels (False: library GEMMs, the r01 path)

    def forward(self, state, attrs, Rr, Rs, p_instance, action=None, particle_den=None, obj_mask=None, **kwargs):
        dev = state.device
        B, N = attrs.size(0), attrs.size(1)
        n_p, n_inst = p_instance.size(1), p_instance.size(2)
        physics_keys = [k for k in kwargs.keys() if k.endswith("_physics_param")]
        assert len(physics_keys) == 1 and action is not None                      # model.py:184-185, 193-194
        phys = kwargs[physics_keys[0]].to(dev, torch.float32)
        views = kwargs.get("edge_views")
        if views is None:
            views = EdgeViews(Rr if isinstance(Rr, CSREdges) else csr_from_dense(Rr, Rs))
        M, nf = B * N, self.nf_effect

        # node inputs [attrs | physics (object slots only) | action], 12-vector [v0, v1, v2, x_cur] per node  (:155-195)
        state_norm = torch.cat([state[:, 1:] - state[:, :-1], state[:, -1:]], 1).transpose(1, 2).reshape(M, -1)
        phys_n = torch.cat([phys[:, None, :].expand(B, n_p, -1), phys.new_zeros(B, N - n_p, phys.shape[1])], 1)
        p_inputs = torch.cat([attrs, phys_n, action], 2).reshape(M, -1)
        group = torch.cat([p_instance, p_instance.new_zeros(B, N - n_p, n_inst)], 1).reshape(M, n_inst)

        # edge inputs [attrs_r | attrs_s | sum|g_r - g_s| | state_norm_r - state_norm_s]  (:220-253): one gather per side
        node_tab = torch.cat([attrs.reshape(M, -1), group, state_norm], 1)
        a, g = attrs.shape[2], n_inst
        if self.fused_dense:
            rel_inputs = edge_inputs(node_tab, views, a, g)
        else:
            tab_r, tab_s = gather_receivers(node_tab, views), gather_senders(node_tab, views)
            rel_inputs = torch.cat([tab_r[:, :a], tab_s[:, :a], (tab_r[:, a:a + g] - tab_s[:, a:a + g]).abs().sum(1, keepdim=True),
                                    tab_r[:, a + g:] - tab_s[:, a + g:]], 1)

        w_rp, b_rp = self.relation_propagator.linear.weight, self.relation_propagator.linear.bias
        w_pp, b_pp = self.particle_propagator.linear.weight, self.particle_propagator.linear.bias
        if self.fused_dense:      # the three Linear(+ReLU) stacks on the fused MFMA chain kernels, forward and backward
            pe, re = self.particle_encoder.model, self.relation_encoder.model
            enc_n = fused_chain("node", p_inputs, [(pe[i].weight, pe[i].bias) for i in (0, 2, 4)])            # :268
            eterm = fused_chain("edge", rel_inputs, [(re[i].weight, re[i].bias) for i in (0, 2, 4)] + [(w_rp[:, :nf], b_rp)]) \
                if views.E else rel_inputs.new_zeros((0, nf))                                                 # :274, :289 first block
        else:
            enc_n = _mlp3(self.particle_encoder, p_inputs)                               # :268
            enc_e = _mlp3(self.relation_encoder, rel_inputs)                             # :274
            eterm = F.linear(enc_e, w_rp[:, :nf], b_rp)                                  # round-invariant edge term
        h = enc_n
        lin = linear if self.fused_dense else F.linear
        pn = lin(enc_n, w_pp[:, :nf], b_pp)                                          # round-invariant block of particle_propagator (:300)
        for _ in range(self.model_config["pstep"]):                                  # :283-301
            if self.fused_dense and (M * nf) % 4 == 0:
                hr, hs = linear2(h, w_rp[:, nf:2 * nf], w_rp[:, 2 * nf:])
                h = add3_relu(lin(message_sum(eterm, hr, hs, views), w_pp[:, nf:]), pn, h)
            else:
                agg = message_sum(eterm, lin(h, w_rp[:, nf:2 * nf]), lin(h, w_rp[:, 2 * nf:]), views)
                h = F.relu(lin(agg, w_pp[:, nf:]) + pn + h)
        d = self.non_rigid_predictor
        x = h.reshape(B, N, nf)[:, :n_p].reshape(B * n_p, nf)
        if self.fused_dense:
            pred_motion = fused_chain("decoder", x, [(d.linear_0.weight, d.linear_0.bias), (d.linear_1.weight, d.linear_1.bias),
                                                     (d.linear_2.weight, d.linear_2.bias)]).reshape(B, n_p, 3)
        else:
            x = F.relu(F.linear(x, d.linear_0.weight, d.linear_0.bias))
            x = F.relu(F.linear(x, d.linear_1.weight, d.linear_1.bias))
            pred_motion = F.linear(x, d.linear_2.weight, d.linear_2.bias).reshape(B, n_p, 3)
        pred_pos = state[:, -1, :n_p] + torch.clamp(pred_motion, max=self.motion_clamp, min=-self.motion_clamp)   # :309
        return pred_pos, pred_motion


def unrolled_loss(model, data, n_future, loss_funcs=None):
    """The n_future-step training objective of train.py:84-108: predict, score against the recorded future state, feed the
    prediction back as the newest history frame (tool slots from `eef_future`, motion from `action_future`), edges fixed.
    `data` is the collated batch dict; it is not modified.  -> scalar loss."""
    if loss_funcs is None:
        loss_funcs = [(F.mse_loss, 1)]
    reset_pending()
    data = dict(data)
    if "edge_views" not in data:
        Rr, Rs = data["Rr"], data.get("Rs")
        data["edge_views"] = EdgeViews(Rr if isinstance(Rr, CSREdges) else csr_from_dense(Rr, Rs))
    loss_sum = 0
    for fi in range(n_future):
        gt_state = data["state_future"][:, fi]
        pred_state, _ = model(**data)
        pred_state_p = pred_state[:, :gt_state.shape[1], :3]
        loss_sum = loss_sum + sum(w * f(pred_state_p, gt_state) for f, w in loss_funcs)
        if fi < n_future - 1:
            nxt = data["eef_future"][:, fi].clone()
            nxt[:, :pred_state_p.shape[1]] = pred_state_p
            data["state"] = torch.cat([data["state"][:, 1:], nxt[:, None]], 1)
            data["action"] = data["action_future"][:, fi]
    return loss_sum
