"""gemm256 on reduction-major operands (ds_read_b64_tr_b16 fragments, untracked LDS-DMA staging) against the all-NT formulation it
would replace (k-contiguous operands + the explicit transposes that make them). Decoder-layer backward shapes, random data.
Usage: python tools/bench_gemm_modes.py [tokens]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mla_amd import hip

T = int(sys.argv[1]) if len(sys.argv) > 1 else 17536
dev = torch.device("cuda:0")
H, I = 4096, 11008


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


print("dgrad  dX[T, Kin] = dY[T, Nout] W[Nout, Kin]:  NT needs W^T (transpose pass), NN reads W as stored (b_mode 1)")
for name, nout, kin in (("qkv", 3 * H, H), ("o", H, H), ("gate|up", 2 * I, H), ("down", H, I)):
    dy, w = rnd(T, nout), rnd(nout, kin, scale=0.02)
    out_nt, out_nn = torch.empty(T, kin, dtype=torch.bfloat16, device=dev), torch.empty(T, kin, dtype=torch.bfloat16, device=dev)
    wt = hip.transpose(w)
    t_tr = timeit(lambda: hip.transpose(w))
    t_nt = timeit(lambda: hip.gemm(dy, wt, out=out_nt))
    t_nn = timeit(lambda: hip.gemm(dy, w, out=out_nn, b_mode=1))
    err = float((out_nn.float() - out_nt.float()).abs().max() / out_nt.float().abs().max())
    fl = 2.0 * T * nout * kin
    print(f"  {name:8s} NT {t_nt:7.3f} ms ({fl / t_nt / 1e9:6.0f} TF/s) + W^T {t_tr:6.3f} ms | NN {t_nn:7.3f} ms ({fl / t_nn / 1e9:6.0f} TF/s) | "
          f"NN - (NT + W^T) = {(t_nn - t_nt - t_tr) * 1e3:7.1f} us   max rel diff {err:.1e}", flush=True)

print("wgrad  dW[Nout, Kin] = dY[T, Nout]^T X[T, Kin]:  NT needs dY^T and X^T, TN reads both as stored (a_mode 1, b_mode 1)")
for name, nout, kin in (("qkv", 3 * H, H), ("o", H, H), ("gate|up", 2 * I, H), ("down", H, I)):
    dy, x = rnd(T, nout), rnd(T, kin)
    dyt, xt = hip.transpose(dy), hip.transpose(x)
    out_nt, out_tn = torch.empty(nout, kin, dtype=torch.float32, device=dev), torch.empty(nout, kin, dtype=torch.float32, device=dev)
    t_tr = timeit(lambda: (hip.transpose(dy), hip.transpose(x)))
    t_nt = timeit(lambda: hip.gemm(dyt, xt, out=out_nt))
    t_tn = timeit(lambda: hip.gemm(dy, x, out=out_tn, a_mode=1, b_mode=1))
    err = float((out_tn - out_nt).abs().max() / out_nt.abs().max())
    fl = 2.0 * T * nout * kin
    print(f"  {name:8s} NT {t_nt:7.3f} ms ({fl / t_nt / 1e9:6.0f} TF/s) + dY^T, X^T {t_tr:6.3f} ms | TN {t_tn:7.3f} ms ({fl / t_tn / 1e9:6.0f} TF/s) | "
          f"TN - NT = {(t_tn - t_nt) * 1e3:7.1f} us   max rel diff {err:.1e}", flush=True)

print("reduction-major operands: gemm256 (force_generic 3) vs gemm128 (force_generic 2)")
for name, M, N, K, am, bm in (("qkv wgrad TN", 3 * H, H, T, 1, 1), ("down dgrad NN", T, I, H, 0, 1), ("o dgrad NN", T, H, H, 0, 1),
                              ("heads 8192 NN", 4096, 8192, 4096, 0, 1), ("heads TN", 8192, 4096, 4096, 1, 1)):
    a = rnd(*((M, K) if am == 0 else (K, M)))
    b = rnd(*((N, K) if bm == 0 else (K, N)))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t3 = timeit(lambda: hip.gemm(a, b, out=out, a_mode=am, b_mode=bm, force_generic=3))
    t2 = timeit(lambda: hip.gemm(a, b, out=out, a_mode=am, b_mode=bm, force_generic=2))
    fl = 2.0 * M * N * K
    print(f"  {name:14s} gemm256 {t3:7.3f} ms ({fl / t3 / 1e9:6.0f} TF/s) | gemm128 {t2:7.3f} ms ({fl / t2 / 1e9:6.0f} TF/s)", flush=True)

print("wgrad with ONE reduction-major operand (the other one k-contiguous, as the producers' fused transposed outputs provide it)")
for name, nout, kin in (("qkv", 3 * H, H), ("o", H, H), ("gate|up", 2 * I, H), ("down", H, I)):
    dy, x = rnd(T, nout), rnd(T, kin)
    dyt, xt = hip.transpose(dy), hip.transpose(x)
    o0 = torch.empty(nout, kin, dtype=torch.float32, device=dev)
    t00 = timeit(lambda: hip.gemm(dyt, xt, out=o0))
    t10 = timeit(lambda: hip.gemm(dy, xt, out=o0, a_mode=1))
    t01 = timeit(lambda: hip.gemm(dyt, x, out=o0, b_mode=1))
    print(f"  {name:8s} NT {t00:7.3f} ms | A as stored (a_mode 1) {t10:7.3f} ms ({(t10 - t00) * 1e3:+7.1f} us) | "
          f"B as stored (b_mode 1) {t01:7.3f} ms ({(t01 - t00) * 1e3:+7.1f} us)", flush=True)
