"""assembly GEMM experiment: correctness vs fp32 matmul + speed vs gemm256. Usage: python tools/exp_asm.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip
from tools.bench_gemm import timeit  # noqa

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in ((256, 256, 128), (512, 768, 256), (1000, 520, 1024), (264, 4096, 384)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    out = hip.gemm(a, b, out_dtype=torch.float32, force_generic=4)
    torch.cuda.synchronize()
    err = float((out - ref).norm() / ref.norm())
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out2 = hip.gemm(a, b, bias=bias, residual=res, force_generic=4, alpha=0.5)
    ref2 = 0.5 * ref + bias.float() + res.float()
    err2 = float((out2.float() - ref2).norm() / ref2.norm())
    print(f"check M={M} N={N} K={K}: rel err fp32-out {err:.2e}  bf16+bias+res {err2:.2e}", flush=True)
T = 17536
if len(sys.argv) > 1 and sys.argv[1] == "ablate":
    names = {0: "full", 1: "no glds", 2: "no ds_read", 3: "no glds, no ds_read", 4: "no barrier", 5: "no setprio", 6: "MFMA only (no glds/reads/barrier)", 7: "reads before loads in k-step 1", 8: "with L2 prefetch 2 tiles ahead", 9: "never waiting for loads (timing only)"}
    for (M, N, K) in ((4096, 4096, T), (T, 12288, 4096)):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for v, nm in names.items():
            ms = timeit(lambda: hip.gemm(a, b, out=out, force_generic=4 + 16 * v))
            print(f"M={M} N={N} K={K} variant {v} ({nm}): {ms:.3f} ms {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
    sys.exit(0)
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for name, M, N, K in (("qkv fwd", T, 12288, 4096), ("o fwd", T, 4096, 4096), ("gu fwd", T, 22016, 4096), ("down fwd", T, 4096, 11008),
                      ("qkv dgrad", T, 4096, 12288), ("gu dgrad", T, 4096, 22016), ("down dgrad", T, 11008, 4096),
                      ("qkv wgrad", 12288, 4096, T), ("o wgrad", 4096, 4096, T), ("gu wgrad", 22016, 4096, T), ("down wgrad", 4096, 11008, T)):
    if K % 128:
        continue
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ms4 = timeit(lambda: hip.gemm(a, b, out=out, force_generic=4 + 16 * VAR))
    ms8 = timeit(lambda: hip.gemm(a, b, out=out))
    fl = 2.0 * M * N * K
    print(f"{name:10s} M={M:6d} N={N:6d} K={K:6d}  asm(v{VAR}) {ms4:7.3f} ms {fl/ms4/1e9:7.1f} TF/s | gemm256 {ms8:7.3f} ms {fl/ms8/1e9:7.1f} TF/s  ratio {ms8/ms4:5.3f}", flush=True)
