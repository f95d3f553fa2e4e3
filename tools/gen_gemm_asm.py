#!/usr/bin/env python
"""Generates the hand-scheduled main loops of the assembly GEMM kernels (mla_amd/csrc/gemm_asm.hip) as inline-asm bodies with
physical registers:  gemm_asm_8w_loop.inc (+ ablation variants) and gemm_asm_8w_clobbers.inc.

Kernel shape "8w": 256x256 block tile, 8 waves = 2 (M) x 4 (N), 128x64 per wave = 8x4 v_mfma_f32_16x16x32_bf16 fragments in
a[0:127]; k-step 32; LDS ring of 4 k-step slots of 32 KiB (A 256 rows x 64 B | B 256 rows x 64 B, 16-B chunks XOR-swizzled).
Per k-step a wave issues 32 MFMAs on fragment buffer P, 12 ds_read_b128 of the next slot into buffer Q and 4
global_load_lds_dwordx4 (its 32 rows of A and of B, three steps ahead); ONE counted vmcnt wait + ONE barrier per k-step.
Waves w and w+4 share a SIMD: they run two differently ordered copies of the loop (X: LDS reads first, loads late; Y: loads
first, reads late) so that one wave's LDS-DMA issue stalls (~60 cycles each) fall into its partner's MFMA stream.

(The 4-wave 128x128-per-wave variant was measured first -- tools/gen_gemm4w.py: MFMA-only 1.80 PFLOP/s, but with ONE wave per
SIMD every global_load_lds issue stalls the only MFMA issuer: 1.03 PFLOP/s with loads, 1.61 without.)

Register map (per lane), 8w:  a[0:127] accumulators, frag (i, j) at a[(i*4+j)*4 ..+3]
  buffer b in {0,1}: A frags v[48b + 4i ..], B frags v[48b + 32 + 4j ..]           (v[0:95])
  v96/v97 A read base (+0 / +65536)   v98/v99 B read base   v100,v101 A load voffsets   v102,v103 B load voffsets
  s[40:41] A tile base  s[42:43] B tile base  s44 next k byte offset  s45 last valid k byte offset  s46 loop counter
  s47 this wave's LDS store base  s[48:49]/s[50:51] current A/B source  s52 scratch
Usage: python tools/gen_gemm_asm.py   (re-run after editing; the .inc files are committed; tests/test_abi.py checks they are in sync)

Round 3. The class that matters now is CfgK64 (64-k tiles with 128-B rows, two 64 KiB tile sets) and its schedule `tile_hbl`
(flag "hbl"): three barriers per K-tile, never more than two non-MFMA instructions between two MFMAs, no vmcnt(0), and the loads
of a tile spread over the K-tile ("ls<S>": one load per S MFMA gaps) -- the waves of a workgroup run in lockstep, so a wave's burst
of loads is the whole CU's burst and stalls every MFMA issuer. main() emits
  * gemm256_kloop.inc / _half1.inc / _clobbers.inc: the PRODUCT main loop of gemm256's k-contiguous instantiations (8 waves,
    gemm256's sub-tile map "map256", peeled last iteration "peel", barrier 3 at gap 8 "wb8", LDS tile-image tail "tail256");
    GEN256_FLAGS overrides the schedule flags for A/B builds;
  * gemm_asm_8w_loop.inc, gemm_asm_4w_loop.inc (+ clobbers, + the timing variants v1..v9): the stand-alone experiment kernels of
    gemm_asm.hip (MLA_EXPERIMENTAL=1 builds); GEN8W_FLAGS / GEN4W_FLAGS select schedules.
Other flags: "wb<N>" gap of barrier 3, "rowmaj" / "colmaj" MFMA order, "noprio", "align" / "align4" loop-head placement, "spread" /
"loadsfirst" / "midbarrier" / "pgr2" the older schedules, and the TIMING-ONLY ablations (wrong results!) "noglds", "noreads",
"nobarrier", "nb1".."nb3", "novmwait", "vgprload", "vgprload_w", "l2pf". GEN_OUT_DIR redirects the output (tests).
"""
import os

CSRC = os.environ.get("GEN_OUT_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mla_amd", "csrc")   # GEN_OUT_DIR: tests
NSLOT, SLOT_BYTES = 4, 32768
AHEAD = NSLOT - 1


class Cfg:
    def __init__(self, name, fi, fj, nload, flags=()):
        self.name, self.FI, self.FJ, self.NLOAD, self.flags = name, fi, fj, nload, set(flags)   # NLOAD: glds per operand per step
        self.fbuf = (fi + fj) * 4                      # VGPRs per fragment buffer
        self.vbase = 2 * self.fbuf                     # first address register
        self.nvgpr = self.vbase + 4 + 2 * nload
        self.nacc = fi * fj * 4

    def acc(self, i, j):
        b = (i * self.FJ + j) * 4
        return f"a[{b}:{b + 3}]"

    def afrag(self, buf, i):
        b = buf * self.fbuf + i * 4
        return f"v[{b}:{b + 3}]"

    def bfrag(self, buf, j):
        b = buf * self.fbuf + self.FI * 4 + j * 4
        return f"v[{b}:{b + 3}]"

    def ds_read(self, dst, is_b, slot, idx):
        reg = self.vbase + (2 if is_b else 0) + (1 if slot >= 2 else 0)
        return f"ds_read_b128 {dst}, v{reg} offset:{(slot % 2) * SLOT_BYTES + idx * 1024}"

    def reads_for(self, slot, buf):
        a = [self.ds_read(self.afrag(buf, x), False, slot, x) for x in range(self.FI)]
        b = [self.ds_read(self.bfrag(buf, x), True, slot, x) for x in range(self.FJ)]
        out = []
        while a or b:                                   # interleave so the operands of the first MFMAs arrive first
            if a:
                out.append(a.pop(0))
            if b:
                out.append(b.pop(0))
            if a and len(a) > len(b):
                out.append(a.pop(0))
        if "noreads" in self.flags:
            out = ["s_nop 0"] * len(out)
        return out

    def load_group(self, slot):
        ptr = ["s_min_u32 s52, s44, s45", "s_add_u32 s48, s40, s52", "s_addc_u32 s49, s41, 0", "s_add_u32 s50, s42, s52",
               "s_addc_u32 s51, s43, 0", "s_add_u32 s44, s44, 64"]
        bundles = [ptr]
        va, vb = self.vbase + 4, self.vbase + 4 + self.NLOAD
        for q in range(self.NLOAD):
            bundles.append([f"s_add_i32 m0, s47, {slot * SLOT_BYTES + q * 1024}", "s_nop 0", f"global_load_lds_dwordx4 v{va + q}, s[48:49]"])
        for q in range(self.NLOAD):
            bundles.append([f"s_add_i32 m0, s47, {slot * SLOT_BYTES + 16384 + q * 1024}", "s_nop 0",
                            f"global_load_lds_dwordx4 v{vb + q}, s[50:51]"])
        if "noglds" in self.flags:
            bundles = [bundles[0]] + [b[:2] for b in bundles[1:]]
        return bundles

    def mfma_order(self):
        if "rowmaj" in self.flags:      # src0 fixed for FI consecutive MFMAs (the order hipBLASLt's hand-written kernel uses)
            return [(i, j) for j in range(self.FJ) for i in range(self.FI)]
        if "colmaj" in self.flags:      # src1 fixed for FJ consecutive MFMAs
            return [(i, j) for i in range(self.FI) for j in range(self.FJ)]
        order = []
        for ib in range(0, self.FI, 2):
            for jb in range(0, self.FJ, 2):
                for i in (ib, ib + 1):
                    for j in (jb, jb + 1):
                        order.append((i, j))
        return order

    def step(self, u, sched):
        cur, nxt = u & 1, (u & 1) ^ 1
        read_slot, load_slot = (u + 1) % NSLOT, (u + AHEAD) % NSLOT
        nload = 2 * self.NLOAD
        lines = [f"; ---- k-step {u} ({sched}): MFMA on buf{cur}, read slot {read_slot} -> buf{nxt}, load slot {load_slot}",
                 f"s_waitcnt vmcnt({nload * (AHEAD - 2)})", "s_waitcnt lgkmcnt(0)"]
        if "nobarrier" not in self.flags:
            lines.append("s_barrier")
        reads, loads = self.reads_for(read_slot, nxt), self.load_group(load_slot)
        nm = self.FI * self.FJ
        aux = {}
        if sched == "X":            # reads first (one per MFMA), loads in the second half
            for n, r in enumerate(reads):
                aux.setdefault(n, []).append(r)
            first = max(len(reads), nm // 2)
            aux.setdefault(first, []).extend(loads[0])
            gap = max(1, (nm - first - 2) // nload)
            for q in range(nload):
                aux.setdefault(first + 1 + q * gap, []).extend(loads[1 + q])
        else:                       # Y: loads first, reads in the second half
            aux.setdefault(0, []).extend(loads[0])
            gap = max(1, (nm // 2 - 2) // nload)
            for q in range(nload):
                aux.setdefault(1 + q * gap, []).extend(loads[1 + q])
            first = nm - len(reads) - 8
            for n, r in enumerate(reads):
                aux.setdefault(first + n, []).append(r)
        assert max(aux) < nm, (max(aux), nm)
        if "noprio" not in self.flags:
            lines.append("s_setprio 1")
        for n, (i, j) in enumerate(self.mfma_order()):
            lines.append(f"v_mfma_f32_16x16x32_bf16 {self.acc(i, j)}, {self.bfrag(cur, j)}, {self.afrag(cur, i)}, {self.acc(i, j)}")
            lines.extend(aux.get(n, []))
        if "noprio" not in self.flags:
            lines.append("s_setprio 0")
        return lines

    def prologue(self):
        vb = self.vbase
        lines = ["; ---- prologue: fixed registers, zero the accumulators, fill the first AHEAD slots, read k-step 0",
                 "s_mov_b64 s[40:41], %[pA]", "s_mov_b64 s[42:43], %[pB]", "s_mov_b32 s44, 0", "s_mov_b32 s45, %[kmax]",
                 "s_mov_b32 s46, %[nit]", "s_mov_b32 s47, %[ldsw]",
                 f"v_mov_b32 v{vb}, %[vA]", f"v_add_u32 v{vb + 1}, 0x10000, v{vb}", f"v_mov_b32 v{vb + 2}, %[vB]",
                 f"v_add_u32 v{vb + 3}, 0x10000, v{vb + 2}"]
        for q in range(self.NLOAD):
            lines.append(f"v_mov_b32 v{vb + 4 + q}, %[oA{q}]")
            lines.append(f"v_mov_b32 v{vb + 4 + self.NLOAD + q}, %[oB{q}]")
        for r in range(self.nacc):
            lines.append(f"v_accvgpr_write_b32 a{r}, 0")
        for g in range(AHEAD):
            for b in self.load_group(g):
                lines.extend(b)
        lines += [f"s_waitcnt vmcnt({2 * self.NLOAD * (AHEAD - 1)})", "s_barrier"]
        lines += self.reads_for(0, 0)
        return lines

    def body(self, two_schedules):
        out = self.prologue()
        tail = ["s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 7", "s_nop 7"]
        if two_schedules:
            out += ["s_cmp_lg_u32 %[sel], 0", "s_cbranch_scc1 2f", "1:"]
            for u in range(4):
                out += self.step(u, "X")
            out += ["s_sub_u32 s46, s46, 1", "s_cmp_lg_u32 s46, 0", "s_cbranch_scc1 1b", "s_branch 3f", "2:"]
            for u in range(4):
                out += self.step(u, "Y")
            out += ["s_sub_u32 s46, s46, 1", "s_cmp_lg_u32 s46, 0", "s_cbranch_scc1 2b", "3:"]
        else:
            out += ["1:"]
            for u in range(4):
                out += self.step(u, "X")
            out += ["s_sub_u32 s46, s46, 1", "s_cmp_lg_u32 s46, 0", "s_cbranch_scc1 1b"]
        return out + tail

    def emit(self, fname, two_schedules=True):
        body = self.body(two_schedules)
        with open(os.path.join(CSRC, fname), "w") as f:
            f.write("// generated by tools/gen_gemm_asm.py -- do not edit\n")
            for ln in body:
                f.write('"' + ln + '\\n"\n')
        return len(body)

    def emit_clobbers(self, fname, macro):
        regs = [f"v{i}" for i in range(self.nvgpr + (3 if "l2pf" in self.flags else 0))] + [f"a{i}" for i in range(self.nacc)]
        with open(os.path.join(CSRC, fname), "w") as f:
            f.write(f"// generated by tools/gen_gemm_asm.py -- do not edit\n#define {macro} \\\n")
            for k in range(0, len(regs), 16):
                last = k + 16 >= len(regs)
                f.write("  " + ", ".join(f'"{r}"' for r in regs[k:k + 16]) + ("\n" if last else ", \\\n"))


class CfgK64(Cfg):
    """Same tile / wave layout, but LDS holds 64-k tiles with 128-B rows (chunk ^= row & 7, the gemm256 layout): every
    global_load_lds instruction then fetches 8 full 128-B lines instead of 16 half lines. Ring of two 64 KiB tile sets
    (A 256 x 128 B | B 256 x 128 B); ONE barrier per 64 k.
      step (t, ks0): lgkmcnt(0);                     MFMA buf0; read (t, ks1) -> buf1
      step (t, ks1): vmcnt(0) lgkmcnt(0) s_barrier;  MFMA buf1; load tile t+2 -> set t&1; read (t+1, ks0) -> buf0 (set (t+1)&1)
    address registers: vbase + 2*ks + set for A, vbase + 4 + 2*ks + set for B; voffsets behind them."""

    PFD = 2

    def __init__(self, name, fi, fj, nload, flags=()):
        super().__init__(name, fi, fj, nload, flags)
        self.nvgpr = self.vbase + 8 + 2 * nload
        self.npf = 2 if "l2pf" in self.flags else 0            # extra VMEM instructions per tile (vmcnt bookkeeping)

    def ds_read(self, dst, is_b, tset, ks, idx):
        reg = self.vbase + (4 if is_b else 0) + 2 * ks + tset
        off = idx * 2048
        if "map256" in self.flags:      # gemm256's wave -> sub-tile map: A fragments at rows (i>>2)*128 + wr*64 + (i&3)*16, B at (j>>1)*128 + wc*32 + (j&1)*16
            off = ((idx >> 1) * 16384 + (idx & 1) * 2048) if is_b else ((idx >> 2) * 16384 + (idx & 3) * 2048)
        return f"ds_read_b128 {dst}, v{reg} offset:{off}"

    def reads_for(self, tset, ks, buf):
        a = [self.ds_read(self.afrag(buf, x), False, tset, ks, x) for x in range(self.FI)]
        b = [self.ds_read(self.bfrag(buf, x), True, tset, ks, x) for x in range(self.FJ)]
        out = []
        while a or b:
            if a:
                out.append(a.pop(0))
            if b:
                out.append(b.pop(0))
            if a and len(a) > len(b):
                out.append(a.pop(0))
        if "noreads" in self.flags:
            out = ["s_nop 0"] * len(out)
        return out

    def load_group(self, tset):
        ptr = ["s_min_u32 s52, s44, s45", "s_add_u32 s48, s40, s52", "s_addc_u32 s49, s41, 0", "s_add_u32 s50, s42, s52",
               "s_addc_u32 s51, s43, 0", "s_add_u32 s44, s44, 128"]
        bundles = [ptr]
        va, vb = self.vbase + 8, self.vbase + 8 + self.NLOAD
        for q in range(self.NLOAD):
            bundles.append([f"s_add_i32 m0, s47, {tset * 65536 + q * 1024}", "s_nop 0", f"global_load_lds_dwordx4 v{va + q}, s[48:49]"])
        for q in range(self.NLOAD):
            bundles.append([f"s_add_i32 m0, s47, {tset * 65536 + 32768 + q * 1024}", "s_nop 0",
                            f"global_load_lds_dwordx4 v{vb + q}, s[50:51]"])
        if "l2pf" in self.flags:
            # L2 prefetch: one dword per 128-B line of the rows this wave stages, PFD tiles beyond the tile being loaded, so the
            # LDS-DMA loads issued later hit L2 instead of exposing HBM latency (the ring only allows one tile of distance)
            pf = self.nvgpr
            bundles[0] = bundles[0][:-1] + [f"s_add_u32 s53, s44, {self.PFD * 128}", "s_min_u32 s53, s53, s45",
                                            "s_add_u32 s54, s40, s53", "s_addc_u32 s55, s41, 0", "s_add_u32 s56, s42, s53",
                                            "s_addc_u32 s57, s43, 0"] + bundles[0][-1:]
            bundles.append([f"global_load_dword v{pf + 2}, v{pf}, s[54:55]"])
            bundles.append([f"global_load_dword v{pf + 2}, v{pf + 1}, s[56:57]"])
        if "noglds" in self.flags:
            bundles = [bundles[0]] + [b[:2] for b in bundles[1:]]
        if "vgprload" in self.flags:      # timing only: plain loads into scratch VGPRs instead of LDS-DMA (no LDS write at all)
            nb = [bundles[0]]
            for k, b in enumerate(bundles[1:]):
                voff, base = b[2].split()[1].rstrip(","), b[2].split()[2]
                nb.append([f"global_load_dwordx4 v[{self.nvgpr}:{self.nvgpr + 3}], {voff}, {base}"])
            bundles = nb
        if "vgprload_w" in self.flags:    # timing only: plain loads + a ds_write_b128 of (stale) registers per load
            nb = [bundles[0]]
            for k, b in enumerate(bundles[1:]):
                voff, base = b[2].split()[1].rstrip(","), b[2].split()[2]
                r = self.nvgpr
                nb.append([f"global_load_dwordx4 v[{r}:{r + 3}], {voff}, {base}", f"ds_write_b128 v{self.vbase + 8}, v[0:3] offset:{k * 1024}"])
            bundles = nb
        return bundles

    def mfmas(self, cur, aux):
        lines = []
        if "noprio" not in self.flags:
            lines.append("s_setprio 1")
        for n, (i, j) in enumerate(self.mfma_order()):
            lines.append(f"v_mfma_f32_16x16x32_bf16 {self.acc(i, j)}, {self.bfrag(cur, j)}, {self.afrag(cur, i)}, {self.acc(i, j)}")
            lines.extend(aux.get(n, []))
        if "noprio" not in self.flags:
            lines.append("s_setprio 0")
        return lines

    def tile(self, tset):
        nm = self.FI * self.FJ
        nload = 2 * self.NLOAD
        # ---- ks0
        lines = [f"; ---- tile set {tset}, k-step 0: MFMA buf0, read ks1 -> buf1", "s_waitcnt lgkmcnt(0)"]
        aux = {}
        rd0 = self.reads_for(tset, 1, 1)
        for n, r in enumerate(rd0):
            aux.setdefault((n * (nm - 2)) // len(rd0) if "spread" in self.flags else n, []).append(r)
        lines += self.mfmas(0, aux)
        # ---- ks1
        lines += [f"; ---- tile set {tset}, k-step 1: MFMA buf1, load tile t+2 -> set {tset}, read next tile ks0 -> buf0",
                  f"s_waitcnt vmcnt({self.npf})", "s_waitcnt lgkmcnt(0)"]
        if "nobarrier" not in self.flags:
            lines.append("s_barrier")
        aux = {}
        loads = self.load_group(tset)
        reads = self.reads_for(tset ^ 1, 0, 0)
        if "novmwait" in self.flags:      # timing only: never wait for the loads
            lines = [ln for ln in lines if not ln.startswith("s_waitcnt vmcnt")]
        if "midbarrier" in self.flags:
            # top-of-step barrier only orders "tile t fully read" -> loads of t+2; the landing of tile t+1 is awaited mid-step
            lines = [ln for ln in lines if ln != "s_waitcnt vmcnt(0)"]
            aux.setdefault(0, []).extend(loads[0])
            for q in range(nload):
                aux.setdefault(1 + q * 2, []).extend(loads[1 + q])
            mid = 2 * nload + 1
            aux.setdefault(mid, []).extend([f"s_waitcnt vmcnt({nload})"] + ([] if "nobarrier" in self.flags else ["s_barrier"]))
            for n, r in enumerate(reads):
                aux.setdefault(mid + 1 + n, []).append(r)
        elif "spread" in self.flags:
            # one-wave-per-SIMD kernels: an LDS-DMA load costs its issuer 3-4 cycles when it sits BETWEEN MFMAs and stalls the only
            # MFMA issuer when loads come in clusters (tools/micro/dma_asm.hip) -> loads and reads alternate, evenly spaced
            aux.setdefault(0, []).extend(loads[0])
            items = []
            ld, rd = list(loads[1:]), list(reads)
            while ld or rd:
                if rd:
                    items.append([rd.pop(0)])
                if ld:
                    items.append(ld.pop(0))
            span = nm - 3
            for k, it in enumerate(items):
                aux.setdefault(1 + (k * span) // len(items), []).extend(it)
        elif "loadsfirst" in self.flags:
            aux.setdefault(0, []).extend(loads[0])
            for q in range(len(loads) - 1):
                aux.setdefault(1 + q * 2, []).extend(loads[1 + q])
            first = nm - len(reads) - 2
            for n, r in enumerate(reads):
                aux.setdefault(first + n, []).append(r)
        else:
            for n, r in enumerate(reads):
                aux.setdefault(n, []).append(r)
            first = len(reads)
            aux.setdefault(first, []).extend(loads[0])
            gap = max(1, (nm - first - 2) // nload)
            for q in range(nload):
                aux.setdefault(first + 1 + q * gap, []).extend(loads[1 + q])
        assert max(aux) < nm, (max(aux), nm)
        lines += self.mfmas(1, aux)
        return lines

    def tile_pgr2(self, tset):
        """One K-tile of the 4-wave loop with loads 1.5 tiles ahead (flag "pgr2"). The LDS set is handed back operand by operand:
             k-step 0 (MFMA buf0):  read A(t,ks1) -> buf1 | lgkmcnt(0) BARRIER 1: A region of this set is free | load A(t+2) | read
                                    B(t,ks1) -> buf1 | lgkmcnt(0) BARRIER 2: B region free | load B(t+2)
             k-step 1 (MFMA buf1):  vmcnt(24) BARRIER 3: A(t+1) landed | read A(t+1,ks0) -> buf0 | vmcnt(16) BARRIER 4: B(t+1) landed |
                                    read B(t+1,ks0) -> buf0
           vmcnt is in-order: behind A(t+1) sit B(t+1), A(t+2), B(t+2) = 24 loads; behind B(t+1) 16. Every load is issued >= 1.3 tiles
           before the barrier that waits for it (the one-barrier schedule: <= 1 tile)."""
        nm = self.FI * self.FJ
        NL = self.NLOAD
        loads = self.load_group(tset)                       # [pointer bundle, A x NL, B x NL]
        ptr, la, lb = loads[0], loads[1:1 + NL], loads[1 + NL:1 + 2 * NL]
        ra1 = [self.ds_read(self.afrag(1, x), False, tset, 1, x) for x in range(self.FI)]
        rb1 = [self.ds_read(self.bfrag(1, x), True, tset, 1, x) for x in range(self.FJ)]
        ra0 = [self.ds_read(self.afrag(0, x), False, tset ^ 1, 0, x) for x in range(self.FI)]
        rb0 = [self.ds_read(self.bfrag(0, x), True, tset ^ 1, 0, x) for x in range(self.FJ)]
        bar = [] if "nobarrier" in self.flags else ["s_barrier"]
        # ---- k-step 0
        aux = {}
        for k, r in enumerate(ra1):
            aux.setdefault(2 * k, []).append(r)
        n = 2 * len(ra1) + 2
        aux.setdefault(n, []).extend(["s_waitcnt lgkmcnt(0)"] + bar)
        aux.setdefault(n + 1, []).extend(ptr)
        for k in range(NL):                                 # A loads and B(ks1) reads alternate, one instruction per MFMA
            aux.setdefault(n + 2 + 2 * k, []).extend(la[k])
            aux.setdefault(n + 3 + 2 * k, []).append(rb1[k] if k < len(rb1) else "s_nop 0")
        n2 = n + 2 + 2 * NL + 2
        aux.setdefault(n2, []).extend(["s_waitcnt lgkmcnt(0)"] + bar)
        gap = max(1, (nm - n2 - 2) // NL)
        for k in range(NL):
            aux.setdefault(n2 + 1 + k * gap, []).extend(lb[k])
        assert max(aux) < nm, (max(aux), nm)
        lines = [f"; ---- tile set {tset}, k-step 0 (pgr2)", "s_waitcnt lgkmcnt(0)"] + self.mfmas(0, aux)
        # ---- k-step 1
        aux = {}
        aux.setdefault(1, []).extend([f"s_waitcnt vmcnt({3 * NL})"] + bar)
        for k, r in enumerate(ra0):
            aux.setdefault(3 + 2 * k, []).append(r)
        n3 = 3 + 2 * len(ra0) + 4
        aux.setdefault(n3, []).extend([f"s_waitcnt vmcnt({2 * NL})"] + bar)
        for k, r in enumerate(rb0):
            aux.setdefault(n3 + 2 + 2 * k, []).append(r)
        assert max(aux) < nm, (max(aux), nm)
        lines += [f"; ---- tile set {tset}, k-step 1 (pgr2)"] + self.mfmas(1, aux)
        return lines


    def tile_hbl(self, tset, final=0):
        """One K-tile in the shape hipBLASLt's hand-written 4-wave kernel shows (round 3, disassembly of its MT256x256x64 custom kernel):
        never more than TWO non-MFMA instructions between two MFMAs (a 16x16x32 MFMA occupies the matrix pipe for 16 cycles and the only
        wave of the SIMD issues one instruction per 4: anything beyond two pushes the next MFMA back), no vmcnt(0), three barriers:
          k-step 0 (MFMA buf0): read A(t, ks1) -> buf1 | lgkmcnt(0), BARRIER 1: every wave has read all of A(t) | pointer update, 8 x
                                load A(t+2) into this set (m0 for the next load is written AFTER each load: no hazard nop) interleaved
                                with the reads of B(t, ks1) | lgkmcnt(0), BARRIER 2 | 8 x load B(t+2)
          k-step 1 (MFMA buf1): vmcnt(16) (= tile t+1 has landed: the 16 loads of t+2 are the only ones allowed in flight), BARRIER 3 |
                                read (t+1, ks0) -> buf0, one per gap | lgkmcnt(0) in front of the last MFMA
        Every load is issued >= 165 MFMAs (1.3 tiles) before the wait that covers it."""
        nm = self.FI * self.FJ
        NL = self.NLOAD
        va, vb = self.vbase + 8, self.vbase + 8 + NL
        ra1 = [self.ds_read(self.afrag(1, x), False, tset, 1, x) for x in range(self.FI)]
        rb1 = [self.ds_read(self.bfrag(1, x), True, tset, 1, x) for x in range(self.FJ)]
        ra0 = [self.ds_read(self.afrag(0, x), False, tset ^ 1, 0, x) for x in range(self.FI)]
        rb0 = [self.ds_read(self.bfrag(0, x), True, tset ^ 1, 0, x) for x in range(self.FJ)]
        ptr = ["s_min_u32 s52, s44, s45", "s_add_u32 s48, s40, s52", "s_addc_u32 s49, s41, 0", "s_add_u32 s50, s42, s52",
               "s_addc_u32 s51, s43, 0", "s_add_u32 s44, s44, 128"]
        bar = [] if "nobarrier" in self.flags else ["s_barrier"]

        def put(aux, gap, *ins):
            if final:
                # peeled last iteration ("peel"): tiles nt-2 (final 1) and nt-1 (final 2) prefetch nothing -- no pointer update, no loads,
                # no barriers 1 / 2 (they only hand LDS regions back to the loads); the last tile does not read a next tile either
                drop = ("global_load_lds", "s_add_i32 m0", "s_min_u32 s52", "s_add_u32 s48", "s_addc_u32 s49", "s_add_u32 s50",
                        "s_addc_u32 s51", "s_add_u32 s44")
                ins = [x for x in ins if not x.startswith(drop)]
            if "noglds" in self.flags:      # timing only
                ins = ["s_nop 0" if x.startswith("global_load_lds") else x for x in ins]
            if "noreads" in self.flags:     # timing only
                ins = ["s_nop 0" if x.startswith("ds_read") else x for x in ins]
            aux.setdefault(gap, []).extend(ins)
            assert len(aux[gap]) <= 2, (gap, aux[gap])

        # ---- k-step 0. Gap positions are derived from nm (MFMAs per k-step) so that the 4-wave (nm 64, 8 + 8 reads, 16 loads) and the
        # 8-wave (nm 32, 8 + 4 reads, 8 loads) configurations share the schedule; for 4 waves they are the hand-placed ones of the
        # first version (reads 0..14, pointer update 15..19, barrier 1 at 21, loads from 22, barrier 2 at 43, barrier 3 at 94/95).
        aux = {}
        dense = 2 * self.FI >= nm // 2                            # 8 waves: one read per gap (a gap is 32 wall cycles with 2 waves/SIMD)
        rstep, lat = (1, 3) if dense else (2, 6)
        for k, r in enumerate(ra1):
            put(aux, rstep * k, r)
        last = rstep * (self.FI - 1)
        put(aux, last + 1, ptr[0], ptr[1])
        put(aux, last + (2 if dense else 3), ptr[2], ptr[3])
        put(aux, last + (3 if dense else 5), ptr[4], ptr[5])
        L1 = last + (4 if dense else 6)
        put(aux, L1, "s_waitcnt lgkmcnt(0)")
        bar1, bar2, bar3 = ([] if f"nb{k}" in self.flags else bar for k in (1, 2, 3))     # timing only: drop one of the barriers
        if final:
            bar1 = bar2 = []
            if final == 2:
                bar3, ra0, rb0 = [], [], []
        put(aux, L1 + 1, *bar1, f"s_add_i32 m0, s47, {tset * 65536}")
        g = L1 + 2
        # load stride in MFMA gaps ("ls<S>", default 2 = back to back behind barriers 1 / 2). The four waves of the workgroup run in
        # lockstep (every barrier re-aligns them), so a cluster of loads in one wave is a cluster of 4x as many in the CU's one
        # address path: 16 loads in 36 gaps stall the issuers (MfmaUtil 65 % with the barriers, 83 % without ANY barrier, 84 % with
        # barriers but without the loads -- round 3 timing variants). hipBLASLt spreads its 16 loads over ~100 of the 128 gaps.
        S = max([int(f[2:]) for f in self.flags if f.startswith("ls") and f[2:].isdigit()] or [0])
        nrb = self.FJ
        g2 = g + 2 * nrb + (4 if not dense else 2)                # lgkmcnt(0) for the B(ks1) reads
        if S:
            pos = [g + q * S for q in range(2 * NL)]
            assert pos[NL] >= g2 + 2 and pos[-1] < 2 * nm - 4, (pos, g2)
        else:
            pos = [g + 2 * q for q in range(NL)] + [g2 + 2 + 2 * q for q in range(NL)]
        m0_of = lambda q: tset * 65536 + (q * 1024 if q < NL else 32768 + (q - NL) * 1024)
        load_at = {}
        for q in range(2 * NL):
            ins = [f"global_load_lds_dwordx4 v{va + q if q < NL else vb + q - NL}, s[{'48:49' if q < NL else '50:51'}]"]
            if q + 1 < 2 * NL:
                ins.append(f"s_add_i32 m0, s47, {m0_of(q + 1)}")
            load_at[pos[q]] = ins
        for gp, ins in load_at.items():
            if gp < nm:
                put(aux, gp, *ins)
        free = [x for x in range(g + 1, g2 - (lat - 2)) if x not in load_at]
        assert len(free) >= nrb, (free, nrb)
        for q in range(nrb):                                       # B(ks1) reads in the gaps between the A loads
            put(aux, free[(q * len(free)) // nrb], rb1[q])
        put(aux, g2, "s_waitcnt lgkmcnt(0)")
        put(aux, g2 + 1, *bar2)
        assert max(aux) < nm, (max(aux), nm)
        lines = [f"; ---- tile set {tset}, k-step 0 (hbl)"] + self.mfmas(0, aux)
        # ---- k-step 1
        aux = {}
        w = max([int(f[2:]) for f in self.flags if f.startswith("wb") and f[2:].isdigit()] or [nm // 2 - 2])     # "wb<N>": gap of barrier 3
        while (nm + w) in load_at or (nm + w + 1) in load_at:
            w += 1
        issued = sum(1 for gp in load_at if gp < nm + w)           # loads of THIS iteration already issued at the wait
        if final == 1:
            issued = 0                                             # nothing of this iteration is in flight: tile nt-1 has to have landed
        if final != 2:
            put(aux, w, f"s_waitcnt vmcnt({issued})")
        put(aux, w + 1, *bar3)
        for gp, ins in load_at.items():
            if gp >= nm:
                put(aux, gp - nm, *ins)
        rd = []
        a, b = list(ra0), list(rb0)
        while a or b:
            if a:
                rd.append(a.pop(0))
            if b:
                rd.append(b.pop(0))
            if a and len(a) > len(b):
                rd.append(a.pop(0))
        free = [x for x in range(w + 2, nm - 2) if (x + nm) not in load_at]
        assert len(free) >= len(rd), (len(free), len(rd))
        for k, r in enumerate(rd):
            put(aux, free[(k * len(free)) // len(rd)], r)
        aux = {g_: i_ for g_, i_ in aux.items() if i_}
        put(aux, nm - 2, "s_waitcnt lgkmcnt(0)")
        assert max(aux) < nm, (max(aux), nm)
        lines += [f"; ---- tile set {tset}, k-step 1 (hbl)"] + self.mfmas(1, aux)
        return lines

    def prologue(self):
        vb = self.vbase
        knit = (["v_readfirstlane_b32 s45, %[kmax]", "v_readfirstlane_b32 s46, %[nit]"] if "tail256" in self.flags      # VGPR operands, see gemm256.hip
                else ["s_mov_b32 s45, %[kmax]", "s_mov_b32 s46, %[nit]"])
        lines = ["; ---- prologue", "s_mov_b64 s[40:41], %[pA]", "s_mov_b64 s[42:43], %[pB]", "s_mov_b32 s44, 0"] + knit + ["s_mov_b32 s47, %[ldsw]",
                 f"v_mov_b32 v{vb}, %[vA0]", f"v_add_u32 v{vb + 1}, 0x10000, v{vb}", f"v_xor_b32 v{vb + 2}, 64, v{vb}",
                 f"v_add_u32 v{vb + 3}, 0x10000, v{vb + 2}",
                 f"v_mov_b32 v{vb + 4}, %[vB0]", f"v_add_u32 v{vb + 5}, 0x10000, v{vb + 4}", f"v_xor_b32 v{vb + 6}, 64, v{vb + 4}",
                 f"v_add_u32 v{vb + 7}, 0x10000, v{vb + 6}"]
        # load voffsets: row q*8 further down = + q * (8 rows * ld * 2 B), clamped to the last valid row of the operand
        va, vbb = vb + 8, vb + 8 + self.NLOAD
        lines += [f"v_mov_b32 v{va}, %[oA0]", f"v_mov_b32 v{vbb}, %[oB0]"]
        for q in range(1, self.NLOAD):
            lines += [f"v_add_u32 v{va + q}, %[sA8], v{va + q - 1}", f"v_add_u32 v{vbb + q}, %[sB8], v{vbb + q - 1}"]
        for q in range(self.NLOAD):
            lines += [f"v_min_u32 v{va + q}, v{va + q}, %[oAmax]", f"v_min_u32 v{vbb + q}, v{vbb + q}, %[oBmax]"]
        if "l2pf" in self.flags:
            lines += [f"v_mov_b32 v{self.nvgpr}, %[pfA]", f"v_mov_b32 v{self.nvgpr + 1}, %[pfB]"]
        for g in range(2):                     # the first two K-tiles are on their way before the accumulators are cleared
            for b in self.load_group(g):
                lines.extend(b)
        for r in range(self.nacc):
            lines.append(f"v_accvgpr_write_b32 a{r}, 0")
        lines += [f"s_waitcnt vmcnt({2 * self.NLOAD + 2 * self.npf})", "s_barrier"]
        lines += self.reads_for(0, 0, 0)
        return lines

    def body(self, two_schedules=False):
        tile = self.tile_hbl if "hbl" in self.flags else self.tile_pgr2 if "pgr2" in self.flags else self.tile
        # code placement of a hand-written stream (MI355X_MICROARCH: a uniform shift = 4 mod 8 bytes costs up to 13 %): "align" puts the
        # loop head on a 256-B boundary, "align4" 4 bytes behind one
        head = [".p2align 8"] if "align" in self.flags else [".p2align 8", "s_nop 0"] if "align4" in self.flags else []
        if "hbl" in self.flags:
            head = ["s_waitcnt lgkmcnt(0)"] + head           # the other schedules wait at the top of every k-step 0
        if "peel" in self.flags:
            # the loop runs nit - 1 times; the last iteration is a copy without the (dummy) prefetch of tiles nt, nt + 1: 2 / nt of the
            # L2 -> LDS traffic and the tail's wait for loads nobody reads
            assert "hbl" in self.flags
            out = self.prologue() + head + ["s_cmp_le_u32 s46, 1", "s_cbranch_scc1 4f", "1:"] + tile(0) + tile(1)
            out += ["s_sub_u32 s46, s46, 1", "s_cmp_gt_u32 s46, 1", "s_cbranch_scc1 1b", "4:"] + self.tile_hbl(0, 1) + self.tile_hbl(1, 2)
        else:
            out = self.prologue() + head + ["1:"] + tile(0) + tile(1)
            out += ["s_sub_u32 s46, s46, 1", "s_cmp_lg_u32 s46, 0", "s_cbranch_scc1 1b"]
        if "tail256" in self.flags:
            return out + self.tail256()
        return out + ["s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 7", "s_nop 7"]

    # ---- gemm256 product loop: the accumulators leave the assembly through LDS, never through registers the compiler also allocates.
    # After the last K-tile the operand ring is dead, so the tile image the C++ epilogues read (gemm256.hip "fast epilogue") is written
    # right here: emode 0 = bf16 image [256][256] (row pitch 512 B, 16-B chunk ^ (row & 31)), emode 1 = fp32 image of accumulator rows
    # ri = 0..3 ([128][256], row pitch 1 KiB, chunk ^ (row & 63)); the second half (ri = 4..7) is written by the separate statement
    # image_f32(1) after the C++ side has consumed the first. %[vImg] = the lane's address of fragment (0, 0) in the image of the mode.
    def image_bf16(self, rowscale=False):
        ln = ["; ---- bf16 tile image" + (" with a per-row scale" if rowscale else ""), "v_mov_b32 v0, %[vImg]", "v_xor_b32 v1, 32, v0",
              "v_add_u32 v2, 0x10000, v0", "v_add_u32 v3, 0x10000, v1"]
        if rowscale:
            # emode 2 (RMSNorm folded into the projection, gemm_args.h rs_*): the kernel left rstd of the tile's 256 rows as floats at LDS
            # byte 0x20000 (behind the operand ring) before the loop; this lane's accumulator row ri is tile row
            # (ri >> 2) * 128 + (ri & 3) * 16 + (wr * 64 + li), and wr * 64 + li = vImg >> 9 in the bf16 image's addressing
            ln += ["v_lshrrev_b32 v20, 9, v0", "v_lshlrev_b32 v20, 2, v20", "v_add_u32 v20, 0x20000, v20"]
            ln += [f"ds_read_b32 v{21 + ri}, v20 offset:{((ri >> 2) * 128 + (ri & 3) * 16) * 4}" for ri in range(self.FI)]
            ln += ["s_waitcnt lgkmcnt(0)"]
        k = 0
        for ri in range(self.FI):
            for ci in range(self.FJ):
                t = 4 + 4 * (k % 4)
                k += 1
                a = (ri * self.FJ + ci) * 4
                base = (ci & 1) + 2 * (ri >> 2)
                off = (ri & 3) * 16 * 512 + (((ci >> 1) ^ (ri & 1)) << 8)
                ln += [f"v_accvgpr_read_b32 v{t + r}, a{a + r}" for r in range(4)]
                scale = f"v{21 + ri}" if rowscale else "%[alpha]"
                ln += [f"v_mul_f32 v{t + r}, {scale}, v{t + r}" for r in range(4)]
                ln += [f"v_cvt_pk_bf16_f32 v{t}, v{t}, v{t + 1}", f"v_cvt_pk_bf16_f32 v{t + 1}, v{t + 2}, v{t + 3}",
                       f"ds_write_b64 v{base}, v[{t}:{t + 1}] offset:{off}"]
        return ln

    def image_f32(self, half):
        ln = [f"; ---- fp32 image of accumulator rows {4 * half}..{4 * half + 3}", "v_mov_b32 v0, %[vImg]", "v_xor_b32 v1, 64, v0",
              "v_xor_b32 v2, 0x100, v0", "v_xor_b32 v3, 0x100, v1"]
        k = 0
        for r4 in range(4):
            for ci in range(self.FJ):
                t = 4 + 4 * (k % 4)
                k += 1
                a = ((4 * half + r4) * self.FJ + ci) * 4
                base = (ci & 1) + 2 * (r4 & 1)
                off = r4 * 16384 + (((ci >> 1) ^ (r4 >> 1)) << 9)
                ln += [f"v_accvgpr_read_b32 v{t + r}, a{a + r}" for r in range(4)]
                ln += [f"v_mul_f32 v{t + r}, %[alpha], v{t + r}" for r in range(4)]
                ln += [f"ds_write_b128 v{base}, v[{t}:{t + 3}] offset:{off}"]
        return ln

    def tail256(self):
        assert (self.FI, self.FJ) == (8, 4)
        return (["s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "s_barrier", "v_readfirstlane_b32 s53, %[emode]", "s_cmp_eq_u32 s53, 0",
                 "s_cbranch_scc0 2f"]
                + self.image_bf16() + ["s_branch 3f", "2:", "s_cmp_eq_u32 s53, 1", "s_cbranch_scc0 4f"] + self.image_f32(0)
                + ["s_branch 3f", "4:"] + self.image_bf16(rowscale=True) + ["3:", "s_waitcnt lgkmcnt(0)"])

    def emit_lines(self, fname, lines):
        with open(os.path.join(CSRC, fname), "w") as f:
            f.write("// generated by tools/gen_gemm_asm.py -- do not edit\n")
            for ln in lines:
                f.write('"' + ln + '\\n"\n')


VARIANTS = {1: {"noglds"}, 2: {"noreads"}, 3: {"noglds", "noreads"}, 4: {"nobarrier"}, 5: {"noprio"}, 6: {"noglds", "noreads", "nobarrier"}}


def main():
    c = CfgK64("8w", 8, 4, 4, set(os.environ.get("GEN8W_FLAGS", "hbl,noprio,ls6").split(",")))    # GEN8W_FLAGS: schedule experiments
    n = c.emit("gemm_asm_8w_loop.inc")
    c.emit_clobbers("gemm_asm_8w_clobbers.inc", "G8W_CLOBBERS")
    print(f"8w/k64: {n} lines, {c.nvgpr} VGPRs + {c.nacc} AGPRs")
    # 4 waves = 2 x 2, 128 x 128 per wave (8 x 8 fragments in a[0:255], two 64-VGPR fragment buffers), one wave per SIMD: -25 % LDS
    # PRODUCT main loop of gemm256's k-contiguous instantiations (gemm256.hip, template parameter ASM): the 8-wave loop with gemm256's
    # sub-tile map, the three-barrier / spread-load schedule, and the tile image written from the assembly
    ck = CfgK64("8w", 8, 4, 4, {"hbl", "map256", "tail256"} | set(os.environ.get("GEN256_FLAGS", "noprio,ls6,peel,wb8").split(",")))   # GEN256_FLAGS: A/B builds
    nk = ck.emit("gemm256_kloop.inc")
    ck.emit_lines("gemm256_kloop_half1.inc", ck.image_f32(1) + ["s_waitcnt lgkmcnt(0)"])
    ck.emit_clobbers("gemm256_kloop_clobbers.inc", "G256K_CLOBBERS")
    print(f"gemm256 k-loop: {nk} lines")
    # traffic per K-tile; every wave stages 64 rows of A and of B (8 + 8 loads per tile)
    c4 = CfgK64("4w", 8, 8, 8, set(os.environ.get("GEN4W_FLAGS", "hbl,noprio,ls6").split(",")))      # GEN4W_FLAGS: timing experiments
    n4 = c4.emit("gemm_asm_4w_loop.inc")
    c4.emit_clobbers("gemm_asm_4w_clobbers.inc", "G4W_CLOBBERS")
    print(f"4w/k64: {n4} lines, {c4.nvgpr} VGPRs + {c4.nacc} AGPRs")
    for v, fl in VARIANTS.items():
        CfgK64("8w", 8, 4, 4, fl | {"loadsfirst"}).emit(f"gemm_asm_8w_loop_v{v}.inc")
    CfgK64("8w", 8, 4, 4).emit("gemm_asm_8w_loop_v7.inc")                                     # reads before loads in k-step 1
    # L2 prefetch (one dword per 128-B line, PFD tiles ahead): measured 1.41 -> 1.05-1.23 PFLOP/s at every distance -- a scattered
    # dword load costs as many cache-line requests as eight LDS-DMA loads; kept as a documented negative result
    CfgK64("8w", 8, 4, 4, {"loadsfirst", "l2pf"}).emit("gemm_asm_8w_loop_v8.inc")
    CfgK64("8w", 8, 4, 4, {"loadsfirst", "novmwait"}).emit("gemm_asm_8w_loop_v9.inc")        # timing only: never wait for loads


if __name__ == "__main__":
    main()
