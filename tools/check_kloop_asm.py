"""Build-time check of gemm256's assembly-loop instantiations (gemm256_kernel<0, 0, EPI, true>).

The accumulators of the hand-scheduled main loop live in a0..a127. In the plain kernel (EPI 0) they cross from the main inline-asm statement
to a second one when the output needs the fp32 tile image (the loop's tail writes accumulator rows 0..3, gemm256_kloop_half1.inc rows 4..7
after the C++ side has consumed the first half). The compiler does not know that, so this script proves from the device assembly that
  * in a kernel with MORE than one inline-asm statement, no compiler-generated instruction after the first statement touches an
    accumulation register (the fused-epilogue kernels have a single statement that consumes every accumulator itself: there the
    compiler may use the registers afterwards), and
  * no assembly-loop kernel spills (scratch_ access) anywhere.
hipcc prints its own instructions behind a tab and the text of an inline-asm statement verbatim from column 0; that is how the two are told
apart. Usage: check_kloop_asm.py <gemm256 device assembly .s>; exit status 1 on a finding."""
import re
import sys


def main(path):
    agpr = re.compile(r"(?<![\w.])a(\[\d+:\d+\]|\d+)\b|accvgpr")
    bad, seen = [], 0
    fn, cand, statements, prev_asm, mfma_in_asm = None, [], 0, False, 0
    # the generated main loop issues 64 MFMAs per K-tile and is unrolled over two K-tiles plus the peeled last pair: a detected asm
    # region with fewer than this many v_mfma lines means the column-0 / tab heuristic below no longer tells the two apart
    MIN_MFMA = 128

    def close():
        if fn is None:
            return
        if statements >= 2:
            bad.extend(cand)
        # advisor (round 3): if a future hipcc indents inline asm, `statements` stays 0 and every check above passes vacuously
        need = 2 if "Li0ELb1EEE" in fn else 1          # the plain kernel (EPI 0) has the second statement for the fp32 half
        if statements < need:
            bad.append((fn, 0, f"only {statements} inline-asm statement(s) detected, expected >= {need}: the assembly listing format changed?"))
        if mfma_in_asm < MIN_MFMA:
            bad.append((fn, 0, f"only {mfma_in_asm} v_mfma lines inside the detected inline-asm regions, expected >= {MIN_MFMA}"))

    for n, ln in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w*gemm256_kernel\w*):", ln)
        if m:
            close()
            fn = m.group(1) if "ELb1EEE" in m.group(1) else None
            seen += fn is not None
            cand, statements, prev_asm, mfma_in_asm = [], 0, False, 0
            continue
        if fn is None:
            continue
        if ln.startswith(".Lfunc_end"):
            close()
            fn = None
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        if ln.startswith(("\t", " ")):       # compiler-generated instruction
            prev_asm = False
            code = s.split(";")[0]
            if "scratch_" in code:
                bad.append((fn, n, s))
            elif statements >= 1 and agpr.search(code):
                cand.append((fn, n, s))
        else:                                 # a line of an inline-asm statement
            if not prev_asm:
                statements += 1
            prev_asm = True
            mfma_in_asm += s.startswith("v_mfma")
    close()
    for f, n, s in bad:
        print(f"{path}:{n}: {f}: accumulator register / scratch access outside the inline assembly: {s}")
    if not seen:
        print(f"{path}: no assembly-loop instantiation of gemm256_kernel found")
        return 1
    print(f"check_kloop_asm: {seen} assembly-loop kernels, {len(bad)} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
