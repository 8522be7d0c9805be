"""autograd.Function wrappers: forward = HIP operator, backward = HIP operator.

The reference gets its gradients from stock autograd through mm / div / cat /
relu; here each operator carries its own backward kernel
(include/sage355.h: sage_linear_act_backward, sage_gather_mean_backward).
"""
import torch

from . import native, ops


class _GatherMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, nbr, cnt, any_nonempty, slot_rows, self_row):
        out = ops.gather_mean(table, nbr, cnt, slot_rows=slot_rows, self_row=self_row, any_nonempty=any_nonempty)
        ctx.save_for_backward(nbr, cnt, slot_rows, self_row)
        ctx.table_shape = tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        nbr, cnt, slot_rows, self_row = ctx.saved_tensors
        rows, dim = ctx.table_shape
        grad_out = grad_out.contiguous()
        grad_table = torch.zeros((rows, dim), dtype=torch.float32, device=grad_out.device)
        n, k = nbr.shape
        rc = native.lib().sage_gather_mean_backward(
            native.ptr(grad_out), grad_out.stride(0), dim, native.ptr(nbr), native.ptr(cnt), k, n, None,
            native.ptr(slot_rows), native.ptr(self_row), native.ptr(grad_table), rows, grad_table.stride(0),
            native.stream_handle())
        native.check(rc, "gather_mean_backward")
        return grad_table, None, None, None, None, None


class _LinearAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, agg, weight, self_tab, self_index, act):
        out = ops.linear_act(agg, weight, act=act, self_tab=self_tab, self_index=self_index)
        ctx.save_for_backward(agg, weight, self_tab, self_index, out)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, grad_out):
        agg, weight, self_tab, self_index, out = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        n, dim = agg.shape
        out_dim, kw = weight.shape
        need_x = ctx.needs_input_grad[0] or (self_tab is not None and ctx.needs_input_grad[2])
        grad_w = torch.zeros_like(weight) if ctx.needs_input_grad[1] else None
        grad_x = torch.empty((n, kw), dtype=torch.float32, device=agg.device) if need_x else None
        # the reproducible form: partial tiles in a workspace, added in a fixed order (no fp32 atomics)
        lib = native.lib()
        ws = torch.empty(max(256, lib.sage_linear_act_backward_workspace_bytes(n, dim, int(self_tab is not None), out_dim)),
                         dtype=torch.uint8, device=agg.device) if grad_w is not None else None
        rc = lib.sage_linear_act_backward_ws(
            native.ptr(self_tab), self_tab.stride(0) if self_tab is not None else 0, native.ptr(self_index),
            native.ptr(agg), agg.stride(0), dim, native.ptr(weight), weight.stride(0), out_dim, int(ctx.act),
            native.ptr(out), out.stride(0), native.ptr(grad_out), grad_out.stride(0), n, None,
            native.ptr(grad_w), grad_w.stride(0) if grad_w is not None else 0,
            native.ptr(grad_x), grad_x.stride(0) if grad_x is not None else 0, None,
            native.ptr(ws), ws.numel() if ws is not None else 0, native.stream_handle())
        native.check(rc, "linear_act_backward")
        grad_agg = grad_self = None
        if grad_x is not None:
            ds = kw - dim
            if ctx.needs_input_grad[0]:
                grad_agg = grad_x[:, ds:]
            if self_tab is not None and ctx.needs_input_grad[2]:
                gs = grad_x[:, :ds]
                if self_index is None:
                    grad_self = torch.zeros_like(self_tab)
                    grad_self[:n] = gs
                else:
                    grad_self = torch.zeros_like(self_tab).index_add_(0, self_index.long(), gs)
        return grad_agg, grad_w, grad_self, None, None


class _TwoHop(torch.autograd.Function):
    """Two stacked Encoders (model.py:219-222) as ONE differentiable operator: forward = TwoHopEngine.forward (sage_forward2),
    backward = TwoHopEngine.backward_weights on the intermediates that forward left in the engine's workspace.  Inputs are the
    two `weight` Parameters (on any device: their gradients go back to where they live); the raw feature table is frozen
    (model.py:214-215).  Output [B, h2] on the GPU."""

    @staticmethod
    def forward(ctx, w1, w2, engine, ids, key):
        out = engine.forward(ids, seed=key)
        ctx.engine, ctx.ids, ctx.key, ctx.generation = engine, ids, key, engine.generation
        ctx.devices = (w1.device, w2.device)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        e = ctx.engine
        (out,) = ctx.saved_tensors
        if e.generation != ctx.generation:
            # another forward has used the workspace since (an evaluation between loss and backward): run this one again --
            # the device sampler is a pure function of (key, node id), so the same sets, rows and h1 come back
            out = e.forward(ctx.ids, seed=ctx.key)
        g1, g2 = e.backward_weights(out, grad_out.to(out.device, torch.float32), need_w1=ctx.needs_input_grad[0])
        g1 = g1.to(ctx.devices[0]) if (g1 is not None and ctx.needs_input_grad[0]) else None
        g2 = g2.to(ctx.devices[1]) if ctx.needs_input_grad[1] else None
        return g1, g2, None, None, None


def two_hop(w1, w2, engine, ids, key):
    return _TwoHop.apply(w1, w2, engine, ids, key)


def gather_mean(table, nbr, cnt, any_nonempty=None, slot_rows=None, self_row=None):
    if torch.is_grad_enabled() and table.requires_grad:
        return _GatherMean.apply(table, nbr, cnt, any_nonempty, slot_rows, self_row)
    return ops.gather_mean(table.detach(), nbr, cnt, slot_rows=slot_rows, self_row=self_row, any_nonempty=any_nonempty)


def linear_act(agg, weight, act, self_tab=None, self_index=None):
    needs = torch.is_grad_enabled() and (agg.requires_grad or weight.requires_grad
                                         or (self_tab is not None and self_tab.requires_grad))
    if needs:
        return _LinearAct.apply(agg, weight, self_tab, self_index, act)
    return ops.linear_act(agg.detach(), weight.detach(), act=act,
                          self_tab=None if self_tab is None else self_tab.detach(), self_index=self_index)
