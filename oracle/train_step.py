"""Oracle (test-only): one full optimisation step on the CPU, composed exactly like
Model._train_on_batch of the reference (/root/reference/models/scene_flow_motion_field.py:152-227):

    depth_1, depth_2 = net_depth(img_1), net_depth(img_2)            (:233-234)
    pred = geometry + Euler-integrated scene-flow MLP                 (:243-264, oracle.losses.predict_train)
    loss = masked flow / disparity losses                             (:285-324)
    loss.backward(retain_graph=True); acc-reg second backward         (:192-195,326-344)
    Adam step on both nets (betas 0.5/0.9)                            (:212-213)

The depth network is injected as a torch module living on the CPU (the networks are
plain ATen graphs in the reference too); everything else is this package.  Used by the
`cpu_baseline` leg of bench.py and by tests; never by the product path.
"""
import time

import torch

from . import losses as L


def mlp_state_from_module(net_sceneflow):
    return {k: v.detach().cpu().clone() for k, v in net_sceneflow.state_dict().items()}


def train_step(opt, depth_net, sd_mlp, batch, warm, lr_depth, lr_mlp, betas=(0.5, 0.9), adam_state=None, return_grads=False):
    """Runs one step IN PLACE on `depth_net` (CPU module) and `sd_mlp` (dict of leaf tensors).
    Returns (batch_log, timings); with `return_grads` the timings dict also carries the gradients the Adam steps
    consumed: 'mlp_grads' {key: tensor} and 'depth_grad_norms' {parameter name: float64 norm}."""
    t0 = time.time()
    depth_net.eval()
    for p in depth_net.parameters():
        p.requires_grad_(not warm)
        p.grad = None
    leaves = {k: v.detach().requires_grad_(True) for k, v in sd_mlp.items()}
    ctx = torch.no_grad() if warm else torch.enable_grad()
    with ctx:
        d1 = depth_net(batch['img_1'])
        d2 = depth_net(batch['img_2'])
    pred = L.predict_train(opt, leaves, batch, d1, d2)
    loss, parts, _ = L.train_losses(opt, warm, batch, pred)
    logged_loss = float(loss.detach())        # `**loss_data` overwrites 'loss' with the unweighted value (:226)
    if opt.weight_steps:
        loss = loss * pred['_steps']
    do_reg = opt.interp_steps > 0 and (not warm or opt.warm_reg) and opt.acc_mul > 0
    if do_reg:
        loss.backward(retain_graph=True)
        reg = L.acceleration_reg(opt, leaves, batch, pred['global_p1'])
        reg.backward()
        acc = float(reg.detach())
    else:
        loss.backward()
        acc = 0.0
    t_grad = time.time()
    grads = {}
    if return_grads:
        grads['mlp_grads'] = {k: v.grad.detach().clone() for k, v in leaves.items()}
        grads['depth_grad_norms'] = {k: float(p.grad.double().norm()) for k, p in depth_net.named_parameters()
                                     if p.grad is not None}
    state = adam_state if adam_state is not None else {}
    groups = []
    if not warm:
        groups.append(('depth', [p for p in depth_net.parameters()], lr_depth))
    groups.append(('mlp', [leaves[k] for k in sorted(leaves)], lr_mlp))
    for name, params, lr in groups:
        if name not in state:
            state[name] = torch.optim.Adam(params, lr=lr, betas=betas)
        else:
            for g in state[name].param_groups:
                g['params'] = params
        state[name].step()
    for k in sd_mlp:
        sd_mlp[k] = leaves[k].detach()
    log = {'loss': logged_loss, 'flow_loss_1_2': float(parts['flow_loss_1_2'].detach()),
           'disp_loss_1_2': float(parts['disp_loss_1_2'].detach()), 'sf_loss': float(parts['sf_loss'].detach()),
           'acc_reg': acc}
    return log, dict(grads, total_s=time.time() - t0, fwd_bwd_s=t_grad - t0)
