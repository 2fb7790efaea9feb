"""Drop-in replacements for the reference's ``op`` package (op/__init__.py:1-2) plus the
additional HIP ops the MI355X engine is built from."""
import contextlib
import threading

from .fused_act import FusedLeakyReLU, FusedLeakyReLU_kml, deferred_sums, fused_leaky_relu, fused_noise_bias_act
from .upfirdn2d import upfirdn2d, upfirdn2d_noise_bias_act
from .conv import (bump_weights_epoch, conv2d, conv2d_bias_act, conv_transpose2d, get_precision, grad_sink,
                   no_param_grads, register_pack_group, set_precision, wgrad_overlap)
from .misc import add_scale, chan_scale, equal_linear, hw_dot, minibatch_stddev, thin_bwdx, thin_fwd, torgb, torgb_fork
from . import linear as _linear      # (module first: the next line rebinds the package attribute `linear` to the function)
from .linear import linear
from . import modconv

# per-thread mode switches: the reference ops are called from nn.DataParallel worker threads
# (train_dynamic_update_prune.py:941-944), so nothing a caller toggles may leak into another thread
_tls = threading.local()


def second_order_enabled():
    return getattr(_tls, 'second_order', False)


@contextlib.contextmanager
def second_order(enabled=True):
    """Inside this context every op builds a graph that can be differentiated again
    (R1: train_dynamic_update_prune.py:89-96; path length: :104-118).  Outside it the modulated
    convolutions and minibatch-stddev use fused single-kernel paths whose backward is
    first-order only (it raises if differentiated twice)."""
    prev = second_order_enabled()
    _tls.second_order = enabled
    try:
        yield
    finally:
        _tls.second_order = prev


__all__ = ['FusedLeakyReLU', 'FusedLeakyReLU_kml', 'fused_leaky_relu', 'fused_noise_bias_act', 'upfirdn2d', 'upfirdn2d_noise_bias_act',
           'conv2d', 'conv2d_bias_act', 'conv_transpose2d', 'set_precision', 'get_precision', 'bump_weights_epoch', 'register_pack_group', 'grad_sink', 'deferred_sums', 'no_param_grads', 'wgrad_overlap',
           'add_scale', 'chan_scale', 'equal_linear', 'hw_dot', 'minibatch_stddev', 'thin_fwd', 'thin_bwdx', 'torgb', 'torgb_fork',
           'second_order', 'second_order_enabled', 'modconv', 'linear']
