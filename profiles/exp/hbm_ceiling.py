"""Practical HBM ceiling on this box for the traffic mix of the depthwise-gate kernel (read 2, write 1) and
for a plain copy, with torch elementwise kernels on tensors of the benchmark's size."""
import torch
n = 16 * 510 * 65536 // 2           # elements of one half of the gate input (B=16, 255 channels, 256x256)
a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda"); c = torch.empty_like(a)
def timeit(f, bytes_, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return bytes_ / ms / 1e6
print("copy      GB/s", round(timeit(lambda: c.copy_(a), 8 * n)))
print("mul 2r1w  GB/s", round(timeit(lambda: torch.mul(a, b, out=c), 12 * n)))
print("read-only GB/s", round(timeit(lambda: a.sum(), 4 * n)))
# write-only and write-heavy mixes (the LayerNorm GEMM of the benchmark reads 96 and writes 510 channels)
d = torch.empty(5 * n // 2, device="cuda")
print("write-only GB/s", round(timeit(lambda: d.fill_(1.0), 4 * d.numel())))
small = torch.randn(n // 5, device="cuda")
big = torch.empty(5, n // 5, device="cuda")
print("1r5w      GB/s", round(timeit(lambda: big.copy_(small[None, :].expand(5, -1)), 4 * (small.numel() + big.numel()))))
