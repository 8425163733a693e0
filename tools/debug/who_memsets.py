"""Which torch operation of the step issues hipMemsetAsync (a memset NODE once captured)?  Run under
rocprofv3 --kernel-trace: every candidate is bracketed by marker kernels (torch.arange of a telling length)."""
import torch
dev = torch.device("cuda:0")
M = 65536
def mark(n): torch.arange(n, device=dev); torch.cuda.synchronize()
x = torch.randn(M, 256, device=dev).bfloat16()
mark(11); z = torch.zeros(8, 8192, 120, device=dev); torch.cuda.synchronize()
mark(12); z.zero_(); torch.cuda.synchronize()
mark(13); z2 = torch.zeros(8, 8192, 10, device=dev, dtype=torch.bfloat16); torch.cuda.synchronize()
for i, (N, K, bias) in enumerate([(1024, 256, True), (512, 256, False), (256, 512, True), (256, 832, True), (64, 256, True), (3, 256, True), (128, 272, True)]):
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev).bfloat16() if bias else None
    dy = torch.randn(M, N, device=dev).bfloat16()
    torch.nn.functional.linear(a, w, b); dy @ w; torch.cuda.synchronize()
    mark(100 + 2 * i); torch.nn.functional.linear(a, w, b); torch.cuda.synchronize()
    mark(101 + 2 * i); dy @ w; torch.cuda.synchronize()
mark(99)
