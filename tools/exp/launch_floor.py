"""Cost of a kernel boundary inside a HIP graph: N dependent trivial kernels, replayed."""
import torch
x = torch.zeros(64, device="cuda:0")
for n in (1, 100, 400):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(n): x.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"graph of {n:4d} dependent 64-element kernels: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per replay, {e0.elapsed_time(e1) / 10 / n * 1e3:6.2f} us per kernel")
