#!/usr/bin/env python
"""Audit of the cross-tile pipelined attention forward (attn_fwd32p_kernel, mla_amd/csrc/attention.hip) from the device assembly:
between its tile statements the scores of the next tile live in physical VGPRs v[64:127] and the output accumulators / Q in AGPRs,
which only the generated assembly (tools/gen_attn_asm.py) may touch (v40..63 are statement-local temporaries: free in between). Checks, for the region from the first to the last tile statement
(the key-tile loop): no compiler instruction names a VGPR >= LIMIT or any AGPR, scratch accesses in the loop are reported. Fails (exit 1) when the
kernel or its tile statements cannot be found -- a check that sees nothing must not pass.
Usage: check_attn_asm.py <attention.s> [kernel-name-fragment] [first clobbered VGPR]"""
import re
import sys


def audit(text, name="attn_fwd32p_kernel", limit=64):
    m = re.search(r"^(_Z\S*%s\S*):" % re.escape(name), text, re.M)
    if not m:
        return [f"kernel {name} not found"], {}
    end = text.index(".end_amdhsa_kernel", m.start())
    lines = text[m.start():end].split("\n")
    blocks, cur = [], None              # (first line, last line, has MFMA, touches AGPRs) of every inline-asm statement
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            cur = [i, i, False, False]
        elif "#ASMEND" in ln and cur:
            cur[1] = i
            blocks.append(tuple(cur))
            cur = None
        elif cur and "v_mfma" in ln:
            cur[2] = True
        elif cur and "v_accvgpr" in ln:
            cur[3] = True
    tiles = [b for b in blocks if b[2]]
    if len(tiles) < 2:
        return [f"{name}: expected two tile statements (parity 0 / 1), found {len(tiles)}"], {}
    lo, hi = tiles[0][0], tiles[-1][1]
    inside = set()
    for a, b, _, _ in blocks:
        inside.update(range(a, b + 1))
    # AGPR lifetime (advisor, round 4): Q (a[96:127]) and the zeroed accumulators are live from the first statement that writes an
    # AGPR (QW / Z4) up to the last one that reads one (EXPORT16), i.e. also across the compiler's staging set-up before the first tile
    # statement and the epilogue arithmetic after the last -- a compiler that used AGPRs as spill space there would corrupt Q or O
    acc = [b for b in blocks if b[3] or b[2]]
    alo, ahi = acc[0][0], acc[-1][1]
    findings, n, scratch = [], 0, 0
    for i in list(range(alo, lo)) + list(range(hi + 1, ahi + 1)):
        if i in inside:
            continue
        t = lines[i].split(";")[0].strip()
        if t and not t.startswith(".") and not t.endswith(":") and re.search(r"\ba\d+\b|\ba\[\d+", t):
            findings.append(f"line {i}: compiler code names an AGPR while Q / O live in AGPRs (outside the tile loop): {t}")
    for i in range(lo, hi + 1):
        if i in inside:
            continue
        t = lines[i].split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        n += 1
        regs = [int(r) for r in re.findall(r"\bv(\d+)\b", t)] + [int(b) for _, b in re.findall(r"\bv\[(\d+):(\d+)\]", t)]
        if any(r >= limit for r in regs):
            findings.append(f"line {i}: compiler code names a VGPR >= v{limit} inside the tile loop: {t}")
        if re.search(r"\ba\d+\b|\ba\[\d+", t):
            findings.append(f"line {i}: compiler code names an AGPR inside the tile loop: {t}")
        if "scratch_" in t:
            scratch += 1
    return findings, dict(statements=len(tiles), loop_instructions=n, scratch_accesses_in_loop=scratch, agpr_live_region_lines=ahi - alo + 1)


def main():
    text = open(sys.argv[1]).read()
    name = sys.argv[2] if len(sys.argv) > 2 else "attn_fwd32p_kernel"
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    findings, info = audit(text, name, limit)
    for f in findings[:40]:
        print("check_attn_asm:", f)
    print(f"check_attn_asm: {name}: {info}, {len(findings)} findings")
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
