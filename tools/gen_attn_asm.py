#!/usr/bin/env python
"""Generates the hand-scheduled iteration bodies of the 32-rows-per-wave attention forward (attn_fwd32p_kernel, mla_amd/csrc/attention.hip,
opt-in via MLA_ATTN_FWD=1): ONE inline-asm statement per (wave, iteration) on v_mfma_f32_32x32x16_bf16 with the output accumulators in
a[0:63] (owned by the assembly for the whole kernel), the operand fragments and Q in a[64:127] and the two score buffers in v[64:127].
Why assembly: HISTORY.md "Round 4" -- at 32 rows per wave the compiler keeps two copies of the 64 accumulators, serialises
ds_read -> s_waitcnt -> v_mfma with one fragment in flight and spills as soon as fragments are batched by hand.
Outputs: attn_fwd32p_tile{0,1}.inc (generic body, score-buffer parity 0 / 1: every case by flag tests), attn_fwd32p_tile{0,1}c.inc (the
same body with the flag tests resolved for the common case: no scalar branches), attn_fwd32p_clobbers.inc.
Hazards padded by hand (cdna_hip_programming.md 5.7 item 2): MFMA D -> VALU 12+ states, VALU -> MFMA operand 2, VALU -> permlane 2,
v_exp_f32 -> non-trans VALU 1.
GEN_ATTN_ABL=nosm,nomfma,nok,nov,nold strips an instruction class for TIMING-ONLY builds (tools/build_attn_abl.sh).
Usage: python tools/gen_attn_asm.py   (tests/test_abi.py checks the committed .inc files are in sync)"""
import os

CSRC = os.environ.get("GEN_OUT_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mla_amd", "csrc")
# With AGPRs in use at two waves per SIMD hipcc splits the unified file 128 arch VGPRs + 128 AGPRs (arch registers >= v128 are
# reserved), so everything only the matrix pipe and the LDS touch lives in AGPRs: the output accumulators, both fragment buffers
# (ds_read_* can target AGPRs, MFMA A / B operands may be AGPRs) and the Q fragments.
AO, FA, FB, AQ = 0, 64, 80, 96                        # AGPRs: O^T accumulators, fragment buffers A / B, Q fragments
NST = 4                                               # LDS-DMA copies per wave and operand (4 waves x 4 x 1 KiB = one 16 KiB tile)
THR = "0x41000000"          # 8.0: the running max moves only when a tile's max exceeds it by more than 2^8


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=1):
    return f"a{base}" if n == 1 else f"a[{base}:{base + n - 1}]"


# ================================================================================================ cross-tile pipelined body (round 4, variant 3)
# One statement per (wave, iteration k): softmax + P V of tile k (scores already in registers, computed by the previous statement) and
# Q K^T of tile k + 1, with the softmax VALU of tile k in the MFMA gaps of Q K^T (k + 1) / P V (k). What hides where follows
# tools/micro/mfma_fillers.hip: a 32x32x16 MFMA occupies the matrix pipe for 32 cycles; ~4 plain VALU or 1 v_exp_f32 + 2-3 plain
# VALU issue under it for free, more stretch the gap.
#   region A: 16 (8) Q K^T MFMAs of tile k+1 -> S_next   | fillers: key block 0 of tile k: fma, exp2, row-sum, pack (56 VALU)
#   region B: 16 (8) P V MFMAs of tile k                 | fillers: key block 1 of tile k (56 VALU, must precede P V steps 2 / 3),
#                                                        |          then the running row maximum of S_next (20 VALU)
# The two score buffers swap roles every iteration (two generated bodies, parity 0 / 1); packed P^T overwrites the first 16 registers
# of the current score buffer in place. The running-max decision (deferred maximum, threshold 2^8) is taken at the top of the statement
# from the maximum the previous statement left in %2.
# Operands: %0 m  %1 l (this lane's half of the row sum)  %2 mx (row max of the scores in S_cur, x sc2)  %3 kaddr (tile k+1)
#   %4 vaddr (tile k)  %5 sc2  %6 thr of tile k+1  %7 flags (SGPR): bit 0 tile k+1 touches the diagonal, bit 1 tile k+1 has two key
#   blocks for this wave, bit 2 issue the staging copies (K of tile k+2, V of tile k+1) from here, bit 3 tile k+1 exists for this wave,
#   bit 4 tile k exists (false in the prologue iteration k = -1), bit 5 tile k has two key blocks
#   %8..%11 / %12..%15 byte offsets of this wave's K / V copies, %16 / %17 SGPR-pair bases of K tile k+2 / V tile k+1,
#   %18 / %19 LDS destinations of this wave's first K / V copy
# Registers: v40..47 scalars, v48..55 K addresses, v56..59 V^T addresses, v[64:95] / v[96:127] the two score buffers;
#   a[0:63] O^T, a[64:79] ring of four K fragments / V^T fragments of odd key steps, a[80:95] V^T fragments of even key steps, a[96:127] Q.
PB_MX, PB_T1, PB_MSAFE, PB_ALPHA, PB_LS0, PB_LS1, PB_NINF, PB_TMP = 40, 41, 42, 43, 44, 45, 46, 47
PB_KA, PB_VA, PB_SC2P, PB_MSP = 48, 56, 60, 62      # K / V^T addresses; (sc2, sc2) and (msafe, msafe) pairs of the packed-fp32 instructions
PB_S = (64, 96)
PO_M, PO_L, PO_MX, PO_KADDR, PO_VADDR, PO_SC2, PO_THR, PO_FLAGS, PO_KOFF, PO_VOFF, PO_KBASE, PO_VBASE, PO_LDSK, PO_LDSV = 0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 17, 18, 19
PF_NDIAG, PF_NTWO, PF_STAGE, PF_NEXT, PF_CUR, PF_CTWO = 0, 1, 2, 3, 4, 5


def spread(groups, fillers, first=0, last=None):
    """groups: list of instruction lists (each starts with its MFMA); fillers spread evenly over the gaps behind groups first..last"""
    last = len(groups) - 1 if last is None else last
    n = last - first + 1
    per = -(-len(fillers) // n) if fillers else 0
    out, pos = [], 0
    for g, grp in enumerate(groups):
        out.extend(grp)
        if first <= g <= last:
            out.extend(fillers[pos:pos + per])
            pos += per
    assert pos >= len(fillers)
    return out


def gen_pipe(par):
    SC, SN = PB_S[par], PB_S[1 - par]
    L = []
    A = L.append
    lab = [100 + 100 * par]

    def label():
        lab[0] += 1
        return str(lab[0])

    def bit(b, target):            # branch to target when flag bit b is clear
        A(f"s_bitcmp1_b32 %{PO_FLAGS}, {b}")
        A(f"s_cbranch_scc0 {target}f")

    def vread(buf, t):
        out = []
        for db in range(4):
            out.append(f"ds_read_b64_tr_b16 {ar(buf + 4 * db, 2)}, v{PB_VA + db} offset:{t * 4096}")
            out.append(f"ds_read_b64_tr_b16 {ar(buf + 4 * db + 2, 2)}, v{PB_VA + db} offset:{t * 4096 + 2048}")
        return out

    def stage(i):
        isv, q = divmod(i, NST)
        off = f"%{(PO_VOFF if isv else PO_KOFF) + q}"
        base = f"%{PO_VBASE if isv else PO_KBASE}"
        dst = f"%{PO_LDSV if isv else PO_LDSK}"
        lb = label()
        return [f"s_bitcmp1_b32 %{PO_FLAGS}, {PF_STAGE}", f"s_cbranch_scc0 {lb}f", f"s_add_u32 m0, {dst}, {q * 1024}", "s_nop 0",
                f"global_load_lds_dwordx4 {off}, {base}", f"{lb}:"]

    def softmax_fillers(e0, e1):
        """scale-and-subtract / exp2 / row-sum / pack of score registers e0..e1-1 of the current tile, two registers per packed-fp32
        instruction where one exists (v_pk_fma_f32, v_pk_add_f32: 80 instead of 112 VALU instructions per 32 scores), as a software
        pipeline: no result is consumed by the next instruction (and a v_exp_f32 result needs one wait state before a non-trans VALU
        reads it)"""
        out = []
        pairs = list(range(e0, e1, 2))
        for t in range(len(pairs) + 3):
            if t < len(pairs):
                e = pairs[t]
                out.append(f"v_pk_fma_f32 v[{SC + e}:{SC + e + 1}], v[{SC + e}:{SC + e + 1}], v[{PB_SC2P}:{PB_SC2P + 1}], v[{PB_MSP}:{PB_MSP + 1}] neg_lo:[0,0,1] neg_hi:[0,0,1]")
            if 0 <= t - 1 < len(pairs):
                e = pairs[t - 1]
                out.append(f"v_exp_f32 v{SC + e}, v{SC + e}")
                out.append(f"v_exp_f32 v{SC + e + 1}, v{SC + e + 1}")
            if 0 <= t - 2 < len(pairs):
                e = pairs[t - 2]
                out.append(f"v_pk_add_f32 v[{PB_LS0}:{PB_LS1}], v[{PB_LS0}:{PB_LS1}], v[{SC + e}:{SC + e + 1}]")
            if 0 <= t - 3 < len(pairs):
                e = pairs[t - 3]
                out.append(f"v_cvt_pk_bf16_f32 v{SC + e // 2}, v{SC + e}, v{SC + e + 1}")
        return out

    def max_fillers():
        out = [f"v_max3_f32 v{PB_MX}, v{SN}, v{SN + 1}, v{SN + 2}"]
        i = 3
        while i < 32:
            b = i + 1 if i + 1 < 32 else i
            out.append(f"v_max3_f32 v{PB_MX}, v{PB_MX}, v{SN + i}, v{SN + b}")
            i += 2
        return out

    def max_finish():
        return [f"v_mov_b32 v{PB_T1}, v{PB_MX}", "s_nop 1", f"v_permlane32_swap_b32 v{PB_MX}, v{PB_T1}", f"v_max_f32 v{PB_MX}, v{PB_MX}, v{PB_T1}",
                f"v_mul_f32 %{PO_MX}, v{PB_MX}, %{PO_SC2}"]

    def region_a(nblk, fill):
        """Q K^T of tile k+1 into S_next: fragments through a ring of four slots in a[64:79] (the read of fragment f + 4 goes out right behind
        MFMA f, which fetched slot f % 4 in its first cycles; LDS data return in order, so lgkmcnt counts the reads still wanted)"""
        frs = [(kb, kd) for kb in range(nblk) for kd in range(8)]
        n = len(frs)

        def rd(f):
            kb, kd = frs[f]
            return f"ds_read_b128 {ar(FA + 4 * (f % 4), 4)}, v{PB_KA + kd} offset:{kb * 8192}"
        out = [rd(f) for f in range(4)]
        groups = []
        for f, (kb, kd) in enumerate(frs):
            g = [f"s_waitcnt lgkmcnt({min(3, n - 1 - f)})"]
            c = "0" if kd == 0 else vr(SN + 16 * kb, 16)
            g.append(f"v_mfma_f32_32x32x16_bf16 {vr(SN + 16 * kb, 16)}, {ar(FA + 4 * (f % 4), 4)}, {ar(AQ + 4 * kd, 4)}, {c}")
            if f + 4 < n:
                g.append(rd(f + 4))
            si = f - 1 if n == 16 else f
            if 0 <= si < 8:
                g.extend(stage(si))
            groups.append(g)
        out.extend(spread(groups, softmax_fillers(0, 16) if fill else []))
        if nblk == 1:
            out.extend(f"v_mov_b32 v{SN + 16 + r}, v{PB_NINF}" for r in range(16))
        return out

    def region_b(T, with_max):
        """P V of tile k: V^T fragments of key step 0 are in a[80:95] (requested at the top of the statement), of step 1 in a[64:79]
        (requested behind the last Q K^T MFMA); steps 2 / 3 are requested behind the MFMAs of steps 0 / 1."""
        groups = []
        for t in range(T):
            buf = FB if t % 2 == 0 else FA
            for db in range(4):
                g = []
                if db == 0:
                    g.append(f"s_waitcnt lgkmcnt({8 if t + 1 < T else 0})")
                g.append(f"v_mfma_f32_32x32x16_bf16 {ar(AO + 16 * db, 16)}, {ar(buf + 4 * db, 4)}, {vr(SC + 4 * t, 4)}, {ar(AO + 16 * db, 16)}")
                if db == 3 and t + 2 < T:
                    if t == 1:
                        g.append("s_waitcnt lgkmcnt(7)")          # never 16 outstanding (4-bit counter)
                    g.extend(vread(buf, t + 2))
                groups.append(g)
        if T == 4:
            f1 = softmax_fillers(16, 32)
            per = 5
            seq = spread(groups[:10], f1 + [""] * (per * 10 - len(f1)))
            seq = [x for x in seq if x]
            # every packed register must be written (and two wait states old) before the P V MFMA that reads it as its B operand
            written = {}
            for i, ln in enumerate(seq):
                if ln.startswith("v_cvt_pk"):
                    written[int(ln.split()[1].strip(",")[1:])] = i
                if ln.startswith("v_mfma"):
                    t = (int(ln.split("v[")[1].split(":")[0]) - SC) // 4
                    for r in range(SC + 4 * t, SC + 4 * t + 4):
                        assert r < SC + 8 or (r in written and written[r] < i - 2), (r, i, written)
            rest = groups[10:]
            tail = spread(rest, max_fillers() if with_max else [])
            out = seq + tail
        else:
            out = spread(groups, max_fillers() if with_max else [], first=2)   # S_next: >= 2 MFMAs (64 cycles) behind the last Q K^T MFMA
        out.append(f"v_add_f32 v{PB_LS0}, v{PB_LS0}, v{PB_LS1}")
        out.append(f"v_add_f32 %{PO_L}, %{PO_L}, v{PB_LS0}")
        if with_max:
            out.extend(max_finish())
        return out

    # ---------------------------------------------------------------- top: running-max decision for tile k
    A(f"v_mov_b32 v{PB_NINF}, 0xff800000")
    nodec, nomove = label(), label()
    bit(PF_CUR, nodec)
    A(f"v_sub_f32 v{PB_T1}, %{PO_MX}, %{PO_M}")                 # -inf - -inf = NaN compares false: no update
    A(f"v_cmp_lt_f32_e32 vcc, {THR}, v{PB_T1}")
    A(f"s_cbranch_vccz {nomove}f")
    A(f"v_max_f32 v{PB_T1}, %{PO_M}, %{PO_MX}")                 # new running max (lanes below the threshold move too: harmless)
    A(f"v_cmp_eq_f32_e32 vcc, v{PB_NINF}, v{PB_T1}")
    A(f"v_cndmask_b32_e64 v{PB_MSAFE}, v{PB_T1}, 0, vcc")
    A(f"v_sub_f32 v{PB_ALPHA}, %{PO_M}, v{PB_MSAFE}")
    A(f"v_exp_f32 v{PB_ALPHA}, v{PB_ALPHA}")
    A(f"v_cmp_neq_f32_e32 vcc, v{PB_NINF}, %{PO_M}")            # any lane with a finite old max? (none at the first tile: O is still zero)
    A(f"v_mov_b32 %{PO_M}, v{PB_T1}")
    A(f"v_mul_f32 %{PO_L}, %{PO_L}, v{PB_ALPHA}")
    A(f"s_cbranch_vccz {nomove}f")
    for base in range(0, 64, 8):
        tmp = [PB_KA + k for k in range(8)]
        for k in range(8):
            A(f"v_accvgpr_read_b32 v{tmp[k]}, a{base + k}")
        A("s_nop 0")
        for k in range(8):
            A(f"v_mul_f32 v{tmp[k]}, v{tmp[k]}, v{PB_ALPHA}")
        for k in range(8):
            A(f"v_accvgpr_write_b32 a{base + k}, v{tmp[k]}")
    A(f"{nomove}:")
    A(f"v_cmp_eq_f32_e32 vcc, v{PB_NINF}, %{PO_M}")
    A(f"v_cndmask_b32_e64 v{PB_MSAFE}, %{PO_M}, 0, vcc")        # a row with nothing unmasked yet: subtract 0, not -inf
    A(f"v_mov_b32 v{PB_LS0}, 0")
    A(f"v_mov_b32 v{PB_LS1}, 0")
    A(f"v_mov_b32 v{PB_MSP}, v{PB_MSAFE}")
    A(f"v_mov_b32 v{PB_MSP + 1}, v{PB_MSAFE}")
    A(f"v_mov_b32 v{PB_SC2P}, %{PO_SC2}")
    A(f"v_mov_b32 v{PB_SC2P + 1}, %{PO_SC2}")
    A(f"{nodec}:")
    # ---------------------------------------------------------------- addresses; V^T fragments of key step 0 of tile k
    A(f"v_mov_b32 v{PB_KA}, %{PO_KADDR}")
    for kd in range(1, 8):
        A(f"v_xor_b32 v{PB_KA + kd}, {kd << 5}, %{PO_KADDR}")
    A(f"v_mov_b32 v{PB_VA}, %{PO_VADDR}")
    for db in range(1, 4):
        A(f"v_xor_b32 v{PB_VA + db}, {db << 6}, %{PO_VADDR}")
    A("s_nop 1")
    nov0 = label()
    bit(PF_CUR, nov0)
    L.extend(vread(FB, 0))
    A(f"{nov0}:")
    # ---------------------------------------------------------------- region A
    a3, a2, a1one, a2one, aend = label(), label(), label(), label(), label()
    bit(PF_NEXT, a3)
    bit(PF_CUR, a2)
    bit(PF_NTWO, a1one)
    L.extend(region_a(2, True))
    A(f"s_branch {aend}f")
    A(f"{a1one}:")
    L.extend(region_a(1, True))
    A(f"s_branch {aend}f")
    A(f"{a2}:")
    bit(PF_NTWO, a2one)
    L.extend(region_a(2, False))
    A(f"s_branch {aend}f")
    A(f"{a2one}:")
    L.extend(region_a(1, False))
    A(f"s_branch {aend}f")
    A(f"{a3}:")
    L.extend(softmax_fillers(0, 16))
    A(f"{aend}:")
    # ---------------------------------------------------------------- V^T fragments of key step 1 (the K ring is free); causal mask of S_next
    nov1 = label()
    bit(PF_CUR, nov1)
    L.extend(vread(FA, 1))
    A(f"{nov1}:")
    nomask = label()
    bit(PF_NDIAG, nomask)
    A("s_nop 15")                                               # MFMA D -> VALU
    for i in range(32):
        kb, r = divmod(i, 16)
        c = kb * 32 + (r >> 2) * 8 + (r & 3)
        A(f"v_cmp_gt_i32_e32 vcc, {c}, %{PO_THR}")
        A(f"v_cndmask_b32_e32 v{SN + i}, v{SN + i}, v{PB_NINF}, vcc")
    A(f"{nomask}:")
    # ---------------------------------------------------------------- region B
    bnocur, bone, b2n, b1n, bend = label(), label(), label(), label(), label()
    bit(PF_CUR, bnocur)
    bit(PF_CTWO, bone)
    bit(PF_NEXT, b2n)
    L.extend(region_b(4, True))
    A(f"s_branch {bend}f")
    A(f"{b2n}:")
    L.extend(region_b(4, False))
    A(f"s_branch {bend}f")
    A(f"{bone}:")
    bit(PF_NEXT, b1n)
    L.extend(region_b(2, True))
    A(f"s_branch {bend}f")
    A(f"{b1n}:")
    L.extend(region_b(2, False))
    A(f"s_branch {bend}f")
    A(f"{bnocur}:")
    A("s_nop 15")
    L.extend(max_fillers())
    L.extend(max_finish())
    A(f"{bend}:")
    return L


def specialise(lines, flags):
    """The generic body with every flag test resolved for one value of %7: s_bitcmp1 / s_cbranch_scc0 / s_branch disappear (a taken
    or not-taken scalar branch costs a wave tens of cycles, and the generic body executes 18 of them per tile); the data-dependent
    s_cbranch_vccz of the running-max decision stays."""
    import re
    where = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\d+):$", ln)
        if m:
            where.setdefault(m.group(1), []).append(i)
    out, pc, scc = [], 0, 0
    while pc < len(lines):
        ln = lines[pc]
        op = ln.split()[0]
        if op == "s_bitcmp1_b32":
            assert ln.split()[1] == f"%{PO_FLAGS},", ln
            scc = (flags >> int(ln.split(",")[1])) & 1
        elif (op == "s_cbranch_scc0" and scc == 0) or op == "s_branch":
            t = ln.split()[1][:-1]
            pc = min(x for x in where[t] if x > pc)
            continue
        elif op == "s_cbranch_scc0":
            pass
        else:
            out.append(ln)
        pc += 1
    return out


PIPE_COMMON = (1 << PF_NTWO) | (1 << PF_STAGE) | (1 << PF_NEXT) | (1 << PF_CUR) | (1 << PF_CTWO)    # a tile below the diagonal, next one too


def write_inc(name, lines):
    with open(os.path.join(CSRC, name), "w") as f:
        f.write("// generated by tools/gen_attn_asm.py -- do not edit\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')


def main():
    abl = set(os.environ.get("GEN_ATTN_ABL", "").split(","))     # TIMING-ONLY ablations (wrong results)

    def keep(ln):
        if "nov" in abl and ln.startswith("ds_read_b64_tr"):
            return False
        if "nok" in abl and ln.startswith("ds_read_b128"):
            return False
        if "nold" in abl and ln.startswith("global_load_lds"):
            return False
        if "nomfma" in abl and ln.startswith("v_mfma"):
            return False
        if "nosm" in abl and ln.startswith(("v_exp_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_add_f32", "v_max3", "v_cvt_pk")):
            return False
        return True
    for par in (0, 1):
        pl = [ln for ln in gen_pipe(par) if keep(ln)]
        write_inc(f"attn_fwd32p_tile{par}.inc", pl)
        print(f"attn_fwd32p_tile{par}.inc: {len(pl)} lines, {sum(ln.startswith('v_mfma') for ln in pl)} MFMAs (all paths)")
        sp = specialise(pl, PIPE_COMMON)
        write_inc(f"attn_fwd32p_tile{par}c.inc", sp)
        print(f"attn_fwd32p_tile{par}c.inc: {len(sp)} lines, {sum(ln.startswith('v_mfma') for ln in sp)} MFMAs (flags == {PIPE_COMMON})")
    clob = [f'"v{i}"' for i in range(40, 128)] + [f'"a{i}"' for i in range(128)] + ['"vcc"', '"scc"', '"memory"']
    with open(os.path.join(CSRC, "attn_fwd32p_clobbers.inc"), "w") as f:
        f.write("// generated by tools/gen_attn_asm.py -- do not edit\n")
        f.write(", ".join(clob) + "\n")


if __name__ == "__main__":
    main()
