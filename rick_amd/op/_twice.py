"""Second-order route of the fused (single-launch) ops.

The reference's ops can be differentiated twice with no caller-side switch: its backward passes are themselves
autograd Functions (op/fused_act.py:19-48, op/upfirdn2d.py:19-85), and its layers are compositions of torch ops.  The
fused forms here (a convolution with its activation tail in the epilogue, the shared-weight modulated convolution, the
blur with its tail, the one-launch ToRGB, minibatch-stddev, the modulation bank) have hand-written first-order backward
kernels.  When such a backward runs with grad mode ENABLED — exactly the ``create_graph=True`` case of R1 and the
path-length term (train_dynamic_update_prune.py:89-96, 104-118) — it instead rebuilds the op's output from the primitive
family that is closed under differentiation (conv / convT / wgrad, chan_scale / hw_dot, thin products, the activation's
L / L*, upfirdn2d) and differentiates that graph, so every higher order stays on HIP kernels and the caller needs no
``with op.second_order():`` (which remains as a hint that skips the fused forward and this recomputation)."""
import torch

from .conv import skip_param_grad


def second_order_backward(compose, inputs, needs, grad_outputs, param_like=None):
    """LOCAL gradients of ``compose(*aliases)`` (a tensor or tuple rebuilt from the op's inputs by twice-differentiable
    ops) w.r.t. the inputs flagged in `needs`, with the graph kept.

    The inputs may depend on each other upstream (the demodulation coefficients d are a function of the style scales s):
    differentiating a graph built on the input tensors themselves would let the gradient w.r.t. s leak through d's
    history and count that path twice.  The graph is therefore built on ALIASES (``t.view_as(t)``): a partial derivative
    w.r.t. an alias follows only the op's own use of that input, while the result stays a differentiable function of the
    original tensors (the alias is a view of them) — which is what the outer, second differentiation needs.
    `param_like[i]`: input i is a leaf parameter whose gradient an enclosing ``op.no_param_grads()`` block does not want."""
    with torch.enable_grad():
        alias = [t.view_as(t) if (torch.is_tensor(t) and t.requires_grad) else t for t in inputs]
        out = compose(*alias)
        outs = list(out) if isinstance(out, (tuple, list)) else [out]
        gouts = list(grad_outputs) if isinstance(grad_outputs, (tuple, list)) else [grad_outputs]
        pairs = [(o, g) for o, g in zip(outs, gouts) if g is not None and o.requires_grad]
        want = [i for i, (t, n) in enumerate(zip(inputs, needs))
                if n and torch.is_tensor(t) and t.requires_grad
                and not (param_like is not None and skip_param_grad(param_like[i]))]
        res = [None] * len(inputs)
        if pairs and want:
            grads = torch.autograd.grad([o for o, _ in pairs], [alias[i] for i in want], [g for _, g in pairs],
                                        create_graph=True, allow_unused=True)
            for i, g in zip(want, grads):
                res[i] = g
    return res
