"""Energy shares of the gemm256 main loop under the power cap. Needs the ablation build:
   MLA_EXTRA_FLAGS=-DMLA_GEMM256_ABLATION mla_amd/csrc/build.sh build_abl;  MLA_HIP_LIB=build_abl/libmla_hip.so python tools/exp_power_abl.py
debug bits: 4 = fragment reads only for the first two K-tiles, 8 = global->LDS loads only for the first two K-tiles, 16/32 = no barriers.
Results are wrong on purpose (timing only); operands stay real, so power tracks the data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit

dev = torch.device("cuda:0")
T, H = 17536, 4096
for name, M, N, K in (("o wgrad", H, H, T), ("qkv dgrad", T, H, 3 * H)):
    for kind in ("randn", "zeros"):
        a = (torch.randn(M, K, device=dev) if kind == "randn" else torch.zeros(M, K, device=dev)).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) if kind == "randn" else torch.zeros(N, K, device=dev)).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K
        for dbg, what in ((0, "full"), (64, "loads always hit L2"), (8, "no global loads"), (4, "no LDS reads"), (12, "no loads, no reads"), (60, "MFMA only (no barriers)")):
            ms = timeit(lambda: hip.gemm(a, b, out=out, force_generic=dbg << 4), iters=20)
            print(f"{name:10s} {kind:6s} {what:26s} {fl/ms/1e9:7.1f} TF/s", flush=True)
