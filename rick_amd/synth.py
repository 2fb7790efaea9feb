"""Deterministic synthetic weights for parity tests and benchmarks.

The FFHQ checkpoint the reference loads (train_dynamic_update_prune.py:871-879) is a
download that is not available offline, so tests and bench fill a state dict in closed
form from the key names: the same call gives bit-identical tensors in the build
container (where goldens are generated from the reference modules) and on the GPU box.
"""
import zlib

import torch


def synth_tensor(key, shape, dtype=torch.float32):
    """Standard-normal tensor seeded by crc32(key); independent of call order."""
    gen = torch.Generator(device='cpu')
    gen.manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=gen, dtype=torch.float32).to(dtype)


def synth_state_dict(shapes, lr_mlp=0.01, dtype=torch.float32):
    """shapes: {key: shape} (e.g. from module.state_dict()).  Returns {key: tensor}.

    Scales follow the reference's default initialisation so activations stay O(1)
    (model_probe_tune.py:107-109,145,229-233,305), but biases / noise strengths are made
    non-zero so every term of the forward is exercised.
    """
    out = {}
    for key, shape in shapes.items():
        if key.endswith('.kernel'):          # FIR taps are constants of the architecture
            continue
        t = synth_tensor(key, shape, dtype)
        if key.startswith('style.') and key.endswith('.weight'):
            t = t / lr_mlp                                    # EqualLinear(lr_mul) init
        elif key.startswith('style.') and key.endswith('.bias'):
            t = t * 0.1 / lr_mlp
        elif key.endswith('modulation.bias'):
            t = 1.0 + 0.1 * t                                 # bias_init=1
        elif key.endswith('noise.weight'):
            t = 0.1 * t
        elif key.endswith('.bias'):
            t = 0.1 * t
        out[key] = t
    return out


def synth_latents(n, dim=512, seed=0):
    return synth_tensor(f'latent/{seed}', (n, dim))


def synth_reals(n, size=256, seed=0):
    """U(-1,1) images, the range of the reference's Normalize(0.5,0.5) pipeline
    (train_dynamic_update_prune.py:795-796)."""
    gen = torch.Generator(device='cpu')
    gen.manual_seed(1000 + seed)
    return torch.rand((n, 3, size, size), generator=gen) * 2 - 1
