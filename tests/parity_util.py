"""Strict per-parameter gradient parity against the reference goldens (round 6).

Every end-to-end golden (tests/golden/mla_tiny_e2e*.npz, written by oracle/capture_golden*.py from the imported reference) carries, for
EVERY parameter that receives a gradient, a sample of the reference's gradient in mode A (fp32) and mode C (the reference in its own GPU
arithmetic: model.to(bf16) + bf16 autocast): `A_gs::<name>` / `C_gs::<name>` = oracle.recipe.grad_slice(grad) -- the whole tensor up to
4 096 elements, else its fixed 64 x 64 corner. The rule is the yardstick alone, per tensor:

    err(hip, A) <= 2 x err(C, A)          err(x, A) = ||x - A||_F / ||A||_F  on the sample

Where the reference's fp32 gradient sample is identically zero (a parameter whose gradient is analytically zero: both reference runs
hold nothing or only rounding noise) err is undefined and the rule becomes ||hip|| <= 2 ||C|| on the sample.
The scalar-norm percentile rule (test_model_gpu.gradnorm_yardstick) stays for the gradient NORMS only."""
import numpy as np

from oracle import recipe


def _err(a, ref):
    return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))


def grad_sample_rows(grads, gold):
    """One row per parameter of gold['grad_names']: dict(name, normA, hip, C, ratio, zero). `hip` / `C` = err(., A) on the sample (or
    the sample's norm when A's is zero); ratio = hip / C (inf when C == 0 < hip; 0 when both are 0)."""
    rows = []
    for n in [str(x) for x in gold["grad_names"]]:
        A, C = gold["A_gs::" + n].astype(np.float64), gold["C_gs::" + n].astype(np.float64)
        g = grads.get(n)
        hip = np.zeros_like(A) if g is None else recipe.grad_slice(g.detach().float().cpu()).numpy().astype(np.float64)
        assert hip.shape == A.shape, (n, hip.shape, A.shape)
        nA = float(np.linalg.norm(A))
        zero = nA == 0.0
        eh, ec = (float(np.linalg.norm(hip)), float(np.linalg.norm(C))) if zero else (_err(hip, A), _err(C, A))
        ratio = 0.0 if eh == 0.0 else (float("inf") if ec == 0.0 else eh / ec)
        rows.append(dict(name=n, normA=nA, hip=eh, C=ec, ratio=ratio, zero=zero, missing=g is None))
    return rows


def strict_violations(rows, exceptions=()):
    """Rows that break err(hip, A) <= 2 x err(C, A) and are not named in `exceptions` (a dict name -> reason, or a set)."""
    return [(r["name"], r["hip"], r["C"]) for r in rows if r["ratio"] > 2.0 and r["name"] not in exceptions]


def format_rows(rows, title):
    out = [f"== {title}: {len(rows)} parameters, per-tensor gradient sample  err(hip, A) | err(C, A) | ratio   (rule: ratio <= 2)"]
    for r in sorted(rows, key=lambda r: -r["ratio"]):
        flag = "  <-- > 2 x C" if r["ratio"] > 2.0 else ""
        z = " [A == 0: norms]" if r["zero"] else ""
        out.append(f"  {r['name']:<105} |A| {r['normA']:.3e}  {r['hip']:.3e} | {r['C']:.3e} | {r['ratio']:.2f}{z}{flag}")
    over = [r for r in rows if r["ratio"] > 2.0]
    out.append(f"  -> {len(rows) - len(over)} / {len(rows)} tensors within 2 x mode C; median ratio {np.median([r['ratio'] for r in rows]):.2f}; "
               f"over: {[r['name'] for r in over]}")
    return "\n".join(out)
