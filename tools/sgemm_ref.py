#!/usr/bin/env python3
"""Context for the roofline numbers (GPU box): the vendor fp32 GEMM (torch.mm -> hipBLASLt/rocBLAS)
on the plain GEMMs that yolov4's biggest conv layers reduce to (im2col already done, no gather, no
epilogue).  Not part of the product path."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("L123 256->512 38x38 b16 (3x3)", 512, 2304, 23104), ("L106 512->1024 19x19 b16 (3x3)", 1024, 4608, 5776),
          ("L29 128->128 76x76 b16 (3x3)", 128, 1152, 92416), ("256->256 38x38 b16 (1x1)", 256, 256, 23104)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(K, N, device="cuda")
    for _ in range(3): c = a @ b
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it = 20
    for _ in range(it): c = a @ b
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    print("%-34s M=%d K=%d N=%d  %.3f ms  %.1f TFLOP/s" % (name, M, K, N, dt * 1e3, 2.0 * M * K * N / dt / 1e12))
