"""Are the library GEMMs torch issues on the trainer's path (EqualLinear of D's final layers through torch.addmm, the einsums of
ToRGB's parameter gradients) bit-reproducible run to run?  Every shape is evaluated REPS times on the same inputs — alone and
with a second process loading the GPU (two ranks share the device in tests/test_gpu_dp.py) — and compared with the first result.

    python tools/diag_gemm_determinism.py [reps]"""
import os
import subprocess
import sys
import time

import torch


def shapes():
    for B in (2, 4, 8):
        yield f'final_linear.0 fwd  B={B}', (B, 8192), (512, 8192), 'addmm'
        yield f'final_linear.0 dgrad B={B}', (B, 512), (8192, 512), 'mm_t'
        yield f'final_linear.0 wgrad B={B}', (512, B), (8192, B), 'mm_t'
        yield f'final_linear.1 fwd  B={B}', (B, 512), (1, 512), 'addmm'
        yield f'mapping fwd B={B}', (B, 512), (512, 512), 'addmm'
    yield 'torgb gw einsum', (4, 3, 128), (4, 128), 'einsum_w'
    yield 'torgb gs einsum', (4, 3, 128), (3, 128), 'einsum_s'


def main(reps):
    torch.manual_seed(0)
    dev = 'cuda'
    bad = 0
    for name, sa, sb, kind in shapes():
        a, b = torch.randn(*sa, device=dev), torch.randn(*sb, device=dev)
        bias = torch.randn(sb[0], device=dev)

        def run():
            if kind == 'addmm':
                return torch.addmm(bias, a, b.t(), beta=1, alpha=0.0110485)
            if kind == 'mm_t':
                return a @ b.t()
            if kind == 'einsum_w':
                return torch.einsum('njc,nc->jc', a, b)
            return torch.einsum('njc,jc->nc', a, b)
        ref = run()
        diff = 0
        for _ in range(reps):
            diff += int(not torch.equal(run(), ref))
        torch.cuda.synchronize()
        bad += diff
        print(f'{name:32s} {diff:5d} / {reps} runs differ from the first', flush=True)
    return bad


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'load':
        x = torch.randn(4096, 4096, device='cuda')
        t0 = time.time()
        while time.time() - t0 < float(sys.argv[2]):
            for _ in range(50):
                x = (x @ x).clamp_(-1, 1)
            torch.cuda.synchronize()
        sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    # (the child is started before this process touches the GPU)
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), 'load', '100'])
    time.sleep(20)
    print('--- with a second process on the same GPU')
    n = main(reps)
    p.wait()
    print('--- alone')
    n += main(reps)
    print('TOTAL differing runs:', n)
