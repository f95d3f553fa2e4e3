#!/usr/bin/env python
"""Generates the hand-scheduled key-tile body of the 32-rows-per-wave attention forward (mla_amd/csrc/attn_fwd32_tile.inc): ONE inline-asm
statement per (wave, 64-key tile) on v_mfma_f32_32x32x16_bf16, with the output accumulators in a[0:63] (owned by the assembly for the
whole kernel), the operand fragments and Q in a[64:127], and 64 physical VGPRs v[64:127] as statement-local temporaries. Why assembly: HISTORY.md "Round 4" -- at 32 rows per wave
the compiler keeps two copies of the 64 accumulators, serialises ds_read -> s_waitcnt -> v_mfma with one fragment in flight and spills
as soon as fragments are batched by hand; the layout below is the one verified by the round-4 compiler version.

Per statement (operands, see attention.hip):
  %0 m (running max, log2 domain)  %1 lpart (this lane's half of the row sum)   [Q fragments: a[96 + 4 kd : 99 + 4 kd], written once]
  %2 kaddr  = LDS byte address of this lane's K row (lane & 31) + ((kh ^ (lane & 15)) << 4); d-step kd is at kaddr ^ (kd << 5),
               32-key block kb at + 8192
  %3 vaddr  = LDS byte address of this lane's V^T gather for d-block 0; d-block db is at vaddr ^ (db << 6), 16-key step t at
               + 4096 t, its second key quad at + 2048
  %4 sc2 = scale * log2(e)   %5 thr = myq - 64 kt - 4 kh (causal threshold of this lane relative to the tile)
  %6 flags (SGPR): bit 0 = the tile touches the diagonal (apply the mask), bit 1 = the upper 32 keys hold unmasked pairs,
     bit 2 = stage the NEXT tile from inside this statement: eight LDS-DMA copies spread between the QK^T MFMAs
  %7..%10 / %11..%14 byte offsets of this wave's four K / V copies, %15 / %16 SGPR-pair bases of the next K / V tile, %17 LDS
     destination of this wave's first K copy (V copies at + 16 KiB)
Register map: S0 = v[64:79], S1 = v[80:95] scores of key block 0 / 1 (register r <-> key kb*32 + (r >> 2)*8 + kh*4 + (r & 3));
FA = a[64:79], FB = a[80:95] two batches of four operand fragments (K rows by ds_read_b128, V^T by ds_read_b64_tr_b16);
PF = v[96:111] P^T packed to bf16 (fragment t = registers 4t..4t+3 = the B operand of key step t); v[112:119] addresses;
v120.. scalars of the softmax. a[16 db : 16 db + 15] = O^T accumulator of d-block db.
Schedule: K fragments in batches of four, two batches in flight (counted lgkmcnt), the next tile's eight LDS-DMA copies behind QK^T
MFMAs, the V^T fragments of key steps 0 / 1 requested right behind QK^T and landing during the softmax, steps 2 / 3 requested under the
P V MFMAs of steps 0 / 1. Measured and NOT kept (HISTORY.md "Round 4"): a sub-tile pipeline with the softmax VALU of one key block in
the MFMA gaps of the other -- 4 cheap VALU per 32-cycle MFMA gap are free (tools/micro/mfma_fillers.hip), 8 cost 17 cycles, a v_exp_f32
~10: the tile's ~140 VALU only hide under MFMAs of ANOTHER tile (cross-tile software pipeline: the next step).
Hazards padded by hand (cdna_hip_programming.md 5.7 item 2): MFMA D -> VALU 12+ states, VALU -> MFMA operand 2, VALU -> permlane 2.
Usage: python tools/gen_attn_asm.py   (writes the .inc; tests/test_abi.py checks it is in sync)"""
import os

CSRC = os.environ.get("GEN_OUT_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mla_amd", "csrc")
# With AGPRs in use at two waves per SIMD hipcc splits the unified file 128 arch VGPRs + 128 AGPRs (arch registers >= v128 are
# reserved), so everything only the matrix pipe and the LDS touch lives in AGPRs: the output accumulators, both fragment buffers
# (ds_read_* can target AGPRs, MFMA A / B operands may be AGPRs) and the Q fragments.
S0, S1, PF, TA = 64, 80, 96, 112                     # arch VGPRs: scores, packed P^T, K addresses (8)
MX, T1, MSAFE, ALPHA, LS0, LS1, NINF, TMP = 120, 121, 122, 123, 124, 125, 126, 127
AO, FA, FB, AQ = 0, 64, 80, 96                        # AGPRs: O^T accumulators, fragment batches A / B, Q fragments
OP_M, OP_L, OP_KADDR, OP_VADDR, OP_SC2, OP_THR, OP_FLAGS = 0, 1, 2, 3, 4, 5, 6
OP_KOFF, OP_VOFF, OP_KBASE, OP_VBASE, OP_LDSDST = 7, 11, 15, 16, 17       # staging: 4 + 4 VGPR byte offsets, two SGPR-pair bases, LDS destination
NST = 4                                                                    # LDS-DMA copies per wave and operand (4 waves x 4 x 1 KiB = one 16 KiB tile)
THR = "0x41000000"          # 8.0: the running max moves only when a tile's max exceeds it by more than 2^8


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=1):
    return f"a{base}" if n == 1 else f"a[{base}:{base + n - 1}]"


def s_reg(i):          # score register i of 32 (S0 ++ S1)
    return S0 + i


def gen():
    L = []
    A = L.append

    def kread(buf, kb, kd0):
        return [f"ds_read_b128 {ar(buf + 4 * i, 4)}, v{TA + kd0 + i} offset:{kb * 8192}" for i in range(4)]

    def vread(buf, t):      # V^T fragments of key step t, d-blocks 0..3 (addresses v[112:115] by then)
        out = []
        for db in range(4):
            out.append(f"ds_read_b64_tr_b16 {ar(buf + 4 * db, 2)}, v{TA + db} offset:{t * 4096}")
            out.append(f"ds_read_b64_tr_b16 {ar(buf + 4 * db + 2, 2)}, v{TA + db} offset:{t * 4096 + 2048}")
        return out

    def pv(buf, t):
        return [f"v_mfma_f32_32x32x16_bf16 {ar(AO + 16 * db, 16)}, {ar(buf + 4 * db, 4)}, {vr(PF + 4 * t, 4)}, {ar(AO + 16 * db, 16)}"
                for db in range(4)]

    def stage(i):
        """i-th staging copy of the NEXT tile (0..3 K, 4..7 V), issued behind an MFMA when flags bit 2 is set: ~60 cycles of issue
        each that would otherwise sit in a burst at the top of the iteration (all waves of the CU at once)"""
        isv, q = divmod(i, NST)
        off = f"%{(OP_VOFF if isv else OP_KOFF) + q}"
        base = f"%{OP_VBASE if isv else OP_KBASE}"
        lab = f"8{i}"
        return [f"s_bitcmp1_b32 %{OP_FLAGS}, 2", f"s_cbranch_scc0 {lab}f", f"s_add_u32 m0, %{OP_LDSDST}, {isv * 16384 + q * 1024}", "s_nop 0",
                f"global_load_lds_dwordx4 {off}, {base}", f"{lab}:"]

    A("; ---- addresses: K d-step kd -> v[112 + kd]")
    A(f"v_mov_b32 v{TA}, %{OP_KADDR}")
    for kd in range(1, 8):
        A(f"v_xor_b32 v{TA + kd}, {kd << 5}, %{OP_KADDR}")
    A(f"v_mov_b32 v{NINF}, 0xff800000")
    A(f"s_bitcmp1_b32 %{OP_FLAGS}, 1")
    A("s_cbranch_scc0 10f")
    # ---------------- two key blocks; fragments in batches of four, two batches in flight; the next tile's eight copies behind MFMAs
    A("; ---- QK^T, key blocks 0 and 1")
    order = [(kb, kd) for kd in range(8) for kb in (0, 1)]                # fragment f -> (kb, kd)

    def kread2(buf, b):
        return [f"ds_read_b128 {ar(buf + 4 * i, 4)}, v{TA + order[4 * b + i][1]} offset:{order[4 * b + i][0] * 8192}" for i in range(4)]

    def qk2(buf, b):
        out = []
        for i in range(4):
            kb, kd = order[4 * b + i]
            sreg = S0 if kb == 0 else S1
            c = "0" if kd == 0 else vr(sreg, 16)
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(sreg, 16)}, {ar(buf + 4 * i, 4)}, {ar(AQ + 4 * kd, 4)}, {c}")
        return out
    L.extend(kread2(FA, 0))
    L.extend(kread2(FB, 1))
    for b in range(4):
        buf = FA if b % 2 == 0 else FB
        A("s_waitcnt lgkmcnt(4)" if b < 3 else "s_waitcnt lgkmcnt(0)")
        m = qk2(buf, b)
        L.extend([m[0]] + stage(2 * b) + [m[1], m[2]] + stage(2 * b + 1) + [m[3]])
        if b + 2 < 4:
            L.extend(kread2(buf, b + 2))
    A("s_branch 11f")
    A("10:")
    # ---------------- upper 32 keys fully masked for this wave: one key block; scores of block 1 = -inf
    L.extend(kread(FA, 0, 0))
    L.extend(kread(FB, 0, 4))
    A("s_waitcnt lgkmcnt(4)")

    def qk1(buf, kd0):
        return [f"v_mfma_f32_32x32x16_bf16 {vr(S0, 16)}, {ar(buf + 4 * i, 4)}, {ar(AQ + 4 * (kd0 + i), 4)}, {'0' if kd0 + i == 0 else vr(S0, 16)}"
                for i in range(4)]
    m = qk1(FA, 0)
    L.extend([m[0]] + stage(0) + [m[1]] + stage(1) + [m[2]] + stage(2) + [m[3]] + stage(3))
    A("s_waitcnt lgkmcnt(0)")
    m = qk1(FB, 4)
    L.extend([m[0]] + stage(4) + [m[1]] + stage(5) + [m[2]] + stage(6) + [m[3]] + stage(7))
    for r in range(16):
        A(f"v_mov_b32 v{S1 + r}, v{NINF}")
    A("11:")
    # V^T addresses (the K addresses are dead) and the fragments of key steps 0 / 1: in flight during the softmax.
    # FA was last read >= 4 MFMAs ago; FB by the four MFMAs issued last -- by the time these reads can return (LDS latency + the
    # eight reads in front of them) those MFMAs have fetched their operands. lgkmcnt is a 4-bit counter: never 16 outstanding.
    A(f"v_mov_b32 v{TA}, %{OP_VADDR}")
    for db in range(1, 4):
        A(f"v_xor_b32 v{TA + db}, {db << 6}, %{OP_VADDR}")
    A("s_nop 1")
    L.extend(vread(FA, 0))
    r1 = vread(FB, 1)
    L.extend(r1[:7])
    A("s_waitcnt lgkmcnt(14)")
    L.extend(r1[7:])
    A("; ---- softmax over 32 scores per lane (MFMA D -> VALU: 8-pass XDL needs 12 wait states; 16 given)")
    A("s_nop 15")
    A(f"s_bitcmp1_b32 %{OP_FLAGS}, 0")
    A("s_cbranch_scc0 20f")
    for i in range(32):
        kb, r = divmod(i, 16)
        c = kb * 32 + (r >> 2) * 8 + (r & 3)
        A(f"v_cmp_gt_i32_e32 vcc, {c}, %{OP_THR}")
        A(f"v_cndmask_b32_e32 v{s_reg(i)}, v{s_reg(i)}, v{NINF}, vcc")
    A("20:")
    A(f"v_max3_f32 v{MX}, v{s_reg(0)}, v{s_reg(1)}, v{s_reg(2)}")
    i = 3
    while i < 32:
        b = i + 1 if i + 1 < 32 else i
        A(f"v_max3_f32 v{MX}, v{MX}, v{s_reg(i)}, v{s_reg(b)}")
        i += 2
    A(f"v_mov_b32 v{T1}, v{MX}")
    A("s_nop 1")
    A(f"v_permlane32_swap_b32 v{MX}, v{T1}")
    A(f"v_max_f32 v{MX}, v{MX}, v{T1}")
    A(f"v_mul_f32 v{MX}, v{MX}, %{OP_SC2}")                    # max of the raw scores x positive scale
    # Deferred maximum (cdna_hip_programming.md T13): the running max moves -- and the 64 accumulators are rescaled, 192 instructions
    # through v_accvgpr_read / write -- only when some lane's tile maximum exceeds its running max by more than THR (log2 units);
    # otherwise the old max stays and p = exp2(s - m_old) <= 2^THR (bf16 keeps its 8 bits of precision at any magnitude; sums in
    # fp32). -inf - -inf = NaN compares false: no update. (Measured: the step time does not depend on it -- the tile is not VALU-bound.)
    A(f"v_sub_f32 v{T1}, v{MX}, %{OP_M}")
    A(f"v_mov_b32 v{ALPHA}, 1.0")
    A(f"v_cmp_lt_f32_e32 vcc, {THR}, v{T1}")
    A("s_cbranch_vccz 31f")
    A(f"v_max_f32 v{T1}, %{OP_M}, v{MX}")                      # new running max (lanes below the threshold move too: harmless)
    A(f"v_cmp_eq_f32_e32 vcc, v{NINF}, v{T1}")
    A(f"v_cndmask_b32_e64 v{MSAFE}, v{T1}, 0, vcc")
    A(f"v_sub_f32 v{ALPHA}, %{OP_M}, v{MSAFE}")
    A(f"v_exp_f32 v{ALPHA}, v{ALPHA}")
    A(f"v_mov_b32 %{OP_M}, v{T1}")
    for base in range(0, 64, 8):      # eight registers per group through v[116:123]... the K-address registers v[116:119] + v[124:127]
        tmp = [TA + 4, TA + 5, TA + 6, TA + 7, LS0, LS1, TMP, MSAFE]
        for k in range(8):
            A(f"v_accvgpr_read_b32 v{tmp[k]}, a{base + k}")
        A("s_nop 0")
        for k in range(8):
            A(f"v_mul_f32 v{tmp[k]}, v{tmp[k]}, v{ALPHA}")
        for k in range(8):
            A(f"v_accvgpr_write_b32 a{base + k}, v{tmp[k]}")
    A("31:")
    A(f"v_cmp_eq_f32_e32 vcc, v{NINF}, %{OP_M}")
    A(f"v_cndmask_b32_e64 v{MSAFE}, %{OP_M}, 0, vcc")          # a row with nothing unmasked yet: subtract 0, not -inf
    for i in range(32):
        A(f"v_fma_f32 v{s_reg(i)}, v{s_reg(i)}, %{OP_SC2}, -v{MSAFE}")
    for i in range(32):
        A(f"v_exp_f32 v{s_reg(i)}, v{s_reg(i)}")
    A(f"v_add_f32 v{LS0}, v{s_reg(0)}, v{s_reg(1)}")
    A(f"v_add_f32 v{LS1}, v{s_reg(2)}, v{s_reg(3)}")
    for i in range(4, 32, 2):
        A(f"v_add_f32 v{LS0}, v{LS0}, v{s_reg(i)}")
        A(f"v_add_f32 v{LS1}, v{LS1}, v{s_reg(i + 1)}")
    A(f"v_add_f32 v{LS0}, v{LS0}, v{LS1}")
    A(f"v_fma_f32 %{OP_L}, %{OP_L}, v{ALPHA}, v{LS0}")
    for j in range(16):
        A(f"v_cvt_pk_bf16_f32 v{PF + j}, v{s_reg(2 * j)}, v{s_reg(2 * j + 1)}")
    A("; ---- P V")
    A("s_waitcnt lgkmcnt(8)")
    A("s_nop 1")
    L.extend(pv(FA, 0))
    A(f"s_bitcmp1_b32 %{OP_FLAGS}, 1")
    A("s_cbranch_scc0 40f")
    L.extend(vread(FA, 2))
    A("s_waitcnt lgkmcnt(8)")
    L.extend(pv(FB, 1))
    L.extend(vread(FB, 3))
    A("s_waitcnt lgkmcnt(8)")
    L.extend(pv(FA, 2))
    A("s_waitcnt lgkmcnt(0)")
    L.extend(pv(FB, 3))
    A("s_branch 41f")
    A("40:")
    A("s_waitcnt lgkmcnt(0)")
    L.extend(pv(FB, 1))
    A("41:")
    return L


def main():
    lines = gen()
    abl = set(os.environ.get("GEN_ATTN_ABL", "").split(","))     # TIMING-ONLY ablations (wrong results): nov, nok, nosm, nomfma
    def keep(ln):
        if "nov" in abl and ln.startswith("ds_read_b64_tr"):
            return False
        if "nok" in abl and ln.startswith("ds_read_b128"):
            return False
        if "nomfma" in abl and ln.startswith("v_mfma"):
            return False
        if "nosm" in abl and ln.startswith(("v_exp_f32", "v_fma_f32 v", "v_add_f32", "v_max3", "v_cvt_pk")):
            return False
        return True
    lines = [ln for ln in lines if keep(ln)]
    with open(os.path.join(CSRC, "attn_fwd32_tile.inc"), "w") as f:
        f.write("// generated by tools/gen_attn_asm.py -- do not edit\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    clob = [f'"v{i}"' for i in range(64, 128)] + [f'"a{i}"' for i in range(128)] + ['"vcc"', '"scc"', '"memory"']
    with open(os.path.join(CSRC, "attn_fwd32_clobbers.inc"), "w") as f:
        f.write("// generated by tools/gen_attn_asm.py -- do not edit\n")
        f.write(", ".join(clob) + "\n")
    n_mfma = sum(ln.startswith("v_mfma") for ln in lines)
    print(f"attn_fwd32_tile.inc: {len(lines)} lines, {n_mfma} MFMAs (both paths)")


if __name__ == "__main__":
    main()
