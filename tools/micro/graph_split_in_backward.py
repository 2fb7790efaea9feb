"""Feasibility probe (round 6, item 6c): can a hipGraph capture be ENDED and a new one BEGUN from inside an autograd hook
(the engine's device thread) so that a backward pass is cut into several graphs at gradient-bucket boundaries?
Needs capture_error_mode='relaxed' (end-capture from another thread)."""
import threading
import torch

dev = 'cuda'
torch.manual_seed(0)
lin = [torch.nn.Linear(256, 256).to(dev) for _ in range(4)]
x = torch.randn(64, 256, device=dev)


def fwd():
    h = x
    for l in lin:
        h = torch.relu(l(h))
    return h.square().mean()


for _ in range(3):      # warm-up (allocator, cuBLAS handles)
    for l in lin:
        l.zero_grad(set_to_none=False)
    fwd().backward()
torch.cuda.synchronize()
ref = [l.weight.grad.clone() for l in lin]

graphs = [torch.cuda.CUDAGraph()]
cut_log = []


def hook(_p):
    # end the running capture, start the next graph in the same pool — on the autograd thread
    cut_log.append(threading.current_thread().name)
    graphs[-1].capture_end()
    g = torch.cuda.CUDAGraph()
    g.capture_begin(pool=graphs[0].pool(), capture_error_mode='relaxed')
    graphs.append(g)


h = lin[2].weight.register_post_accumulate_grad_hook(hook)
for l in lin:
    l.weight.grad.zero_()
    l.bias.grad.zero_()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    graphs[0].capture_begin(capture_error_mode='relaxed')
    loss = fwd()
    loss.backward()
    graphs[-1].capture_end()
torch.cuda.current_stream().wait_stream(s)
h.remove()
print('main thread', threading.current_thread().name, 'cut on', cut_log, 'graphs', len(graphs))
for l in lin:
    l.weight.grad.zero_()
    l.bias.grad.zero_()
for g in graphs:
    g.replay()
torch.cuda.synchronize()
print('equal to eager:', [bool(torch.equal(l.weight.grad, r)) for l, r in zip(lin, ref)])
for l in lin:
    l.weight.grad.zero_()
    l.bias.grad.zero_()
graphs[0].replay()
torch.cuda.synchronize()
print('after first segment only (late layers done, early not):', [float(l.weight.grad.abs().sum()) > 0 for l in lin])
