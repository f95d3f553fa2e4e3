"""Build-time check of gemm256's assembly-loop instantiations (gemm256_kernel<0, 0, EPI, true>).

The accumulators of the hand-scheduled main loop live in a0..a127 across TWO inline-asm statements when the output is fp32 (the loop's
tail writes accumulator rows 0..3 to the LDS image, gemm256_kloop_half1.inc writes rows 4..7 after the C++ side has consumed the first
half). The compiler does not know that, so this script proves from the device assembly that it never touches an accumulation register
(or spills) outside the inline-asm blocks of those kernels. Usage: check_kloop_asm.py <gemm256 device assembly .s>; exit status 1 on a finding."""
import re
import sys


def main(path):
    fn, bad, seen = None, [], 0
    agpr = re.compile(r"(?<![\w.])a(\[\d+:\d+\]|\d+)\b|accvgpr|scratch_")
    for n, ln in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w*gemm256_kernel\w*):", ln)
        if m:
            fn = m.group(1) if "ELb1EEE" in m.group(1) else None
            seen += fn is not None
            continue
        if fn is None:
            continue
        if ln.startswith(".Lfunc_end") or ln.lstrip().startswith(".end_amdhsa_kernel"):
            fn = None
            continue
        s = ln.strip()
        # hipcc prints its own instructions behind a tab and the text of an inline-asm statement verbatim, i.e. from column 0
        if ln.startswith("\t") and s and not s.startswith((";", ".")) and agpr.search(s.split(";")[0]):
            bad.append((fn, n, s))
    for fn, n, s in bad:
        print(f"{path}:{n}: {fn}: accumulator register / scratch access outside the inline assembly: {s}")
    if not seen:
        print(f"{path}: no assembly-loop instantiation of gemm256_kernel found")
        return 1
    print(f"check_kloop_asm: {seen} assembly-loop kernels, {len(bad)} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
