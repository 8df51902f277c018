"""How fast is the vendor fp32 GEMM (torch.matmul -> rocBLAS/hipBLASLt, highest precision = no TF32/xf32) on the shapes of
the path?  Context for the engine's 110-137 TFLOP/s; cross-check only, nothing in the product calls a BLAS."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
torch.set_float32_matmul_precision("highest")
dev = torch.device("cuda:0")
for (M, K, N) in ((200704, 1024, 256), (200704, 256, 1024), (802816, 128, 512), (3211264, 64, 256), (50176, 2048, 512), (4096, 4096, 4096), (8192, 8192, 8192)):
    a = torch.rand(M, K, device=dev) - 0.5; b = torch.rand(K, N, device=dev) - 0.5
    for _ in range(3): c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("M=%d K=%d N=%d: %.3f ms  %.1f TFLOP/s" % (M, K, N, ms, 2.0 * M * K * N / ms / 1e9))
