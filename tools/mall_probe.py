"""Does a producer->consumer pair run faster when the intermediate fits the 256 MB Infinity Cache (MALL)?"""
import torch
dev = torch.device("cuda:0")
for mb in (16, 32, 64, 128, 192, 256, 512, 1024, 3072):
    n = mb * 1024 * 1024 // 4
    x = torch.rand(n, device=dev); a = torch.empty_like(x); c = torch.empty_like(x)
    def pair():
        torch.mul(x, 2.0, out=a)      # read x, write a
        torch.add(a, 1.0, out=c)      # read a (just written), write c
    reps = max(3, 3072 // mb)
    for _ in range(2): pair()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): pair()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%5d MB: pair %.3f ms -> %.2f TB/s over the 4 streams" % (mb, ms, 4 * n * 4 / ms / 1e9))
