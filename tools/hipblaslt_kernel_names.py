"""Runs torch.matmul (hipBLASLt) on the 7B GEMM shapes so that `rocprofv3 --kernel-trace --stats` shows which library kernel serves them
(round 1: a stream-K custom kernel, ..._SK3_..._MT256x256x64_MI16x16x1_...). Usage: rocprofv3 --kernel-trace --stats -- python tools/hipblaslt_kernel_names.py"""
import torch
T=17536
dev=torch.device("cuda:0")
for (M,N,K) in ((T,12288,4096),(T,22016,4096),(12288,4096,T),(4096,4096,T),(T,4096,11008)):
    a=torch.randn(M,K,device=dev).to(torch.bfloat16); b=torch.randn(N,K,device=dev).to(torch.bfloat16)
    for _ in range(3): c=a@b.t()
    torch.cuda.synchronize()
