"""CPU oracle for the RICK StyleGAN2 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``rick_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / reported CPU baseline.

Contents
--------
ops_ref.py     pure-PyTorch (CPU, fp32/fp64) restatement of the reference's two
               custom ops (op/upfirdn2d.py:159-200, op/fused_bias_act_kernel.cu:28-47)
model_ref.py   functional restatement of Generator / Discriminator
               (gan_training/models/model_probe_tune.py) driven by a state_dict
train_ref.py   losses, R1 / path-length penalties, Fisher estimate and the
               percentile freeze/fine-tune/prune decisions
               (train_dynamic_update_prune.py:82-118,214-393)
csrc/          plain-C scalar restatement of upfirdn2d / bias-act index math
               (op/upfirdn2d_kernel.cu:49-105, op/fused_bias_act_kernel.cu:18-49),
               built into oracle/_build/liboracle.so by oracle/Makefile

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference's own Python modules imported
on CPU in the build container (tools/make_golden.py -> tests/golden/*.npz;
tests/test_oracle_vs_golden.py).  The reference's CUDA kernels cannot be built
here (no nvcc), so parity with the original cuDNN arithmetic is unpinned; the
enforceable pin is the reference's CPU path.
"""
