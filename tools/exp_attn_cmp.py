"""Compare two dumps of tools/exp_attn_bits.py at tolerance level: per tensor max abs diff and Frobenius-relative error."""
import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
worst = 0.0
for k in a:
    x, y = a[k].float(), b[k].float()
    fin = torch.isfinite(x) & torch.isfinite(y)
    same_inf = bool((torch.isfinite(x) == torch.isfinite(y)).all())
    d = (x[fin] - y[fin])
    rel = float(d.norm() / (x[fin].norm() + 1e-30))
    worst = max(worst, rel)
    if k.endswith("_o") or k.endswith("_lse"):
        print(f"{k:28s} rel {rel:.2e} max|d| {float(d.abs().max()) if d.numel() else 0:.2e} nan/inf pattern equal: {same_inf}")
print("worst rel", worst)
