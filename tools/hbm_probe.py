"""Achievable HBM rates of plain streaming kernels on this box (PyTorch elementwise kernels): the practical roof for the
HBM-bound layers (fill = write only, copy = 1R+1W, add = 2R+1W, sum = read only)."""
import torch
dev = torch.device("cuda:0")
n = 822 * 1024 * 1024  # floats: 3.29 GB, the conv2_x 256-channel activation at 1024 frames
a = torch.empty(n, device=dev); b = torch.rand(n, device=dev); c = torch.rand(n, device=dev)
def t(fn, nbytes, name, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print("%-22s %.3f ms  %.2f TB/s" % (name, ms, nbytes / ms / 1e9))
t(lambda: a.fill_(1.0), n * 4, "fill (W)")
t(lambda: a.copy_(b), 2 * n * 4, "copy (R+W)")
t(lambda: torch.add(b, c, out=a), 3 * n * 4, "add (2R+W)")
t(lambda: b.sum(), n * 4, "sum (R)")
t(lambda: torch.relu_(b), 2 * n * 4, "relu_ in place (R+W)")
