"""A few launches of one NT GEMM shape through gemm256 and through hipBLASLt (torch.matmul) so that rocprofv3 --pmc can attribute
fabric-side counters per kernel:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o p --output-format csv -- python tools/pmc_gemm.py [M N K]
    python tools/pmc_summary.py out/p_counter_collection.csv "gemm256_kernel|Cijk_\\w{0,40}"
Round 1 (KB per launch): M=16384 N=4096 K=16384: hipBLASLt 2.26e6 / gemm256 2.50e6; M=17536 N=4096 K=22016: 5.27e6 / 2.77e6."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 4096, 16384)
a = torch.randn((M, K), device=dev).to(torch.bfloat16)
b = torch.randn((N, K), device=dev).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
FG = [int(v) for v in os.environ.get("PMC_GEMM_FG", "0").split(",")]      # force_generic values to run (4 / 5: assembly kernels, experiment build)
for _ in range(4):
    for fg in FG:
        hip.gemm(a, b, out=out, force_generic=fg)
    torch.matmul(a, b.t(), out=out)
torch.cuda.synchronize()
