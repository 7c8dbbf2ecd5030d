"""Developer probe: which hipBLASLt kernels (macro tile, wave tiling, split) torch.matmul picks on the product's GEMM shapes.
Run under rocprofv3 --kernel-trace --stats; the kernel names carry the solution parameters."""
import torch
dev = torch.device("cuda:0")
for m in (2050, 16400, 12290, 1025):
    for n, k in ((12288, 1536), (4608, 1536), (1536, 1536), (1536, 6144)):
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = torch.randn(n, k, device=dev).to(torch.bfloat16)
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        torch.cuda.nvtx.range_push(f"M{m}N{n}K{k}") if False else None
        for _ in range(3):
            torch.matmul(a, w.t(), out=out)
        torch.cuda.synchronize()
        print("done", m, n, k, flush=True)
