"""Instruction counts by class of every MFMA-bearing basic block of a kernel in a device-assembly listing (build/attention.s):
what one wave issues per tile -- the currency of the latency-bound attention kernels (one instruction per ~4-7 cycles).
Usage: python tools/count_loop_instrs.py <file.s> <kernel-name-substring> [min_mfma]"""
import collections
import re
import sys


def classify(l):
    t = l.strip().split()
    if not t or t[0].startswith((';', '.')) or t[0].endswith(':'):
        return None
    op = t[0]
    for pre, cls in (('v_mfma', 'mfma'), ('ds_read_b64_tr', 'ds_tr'), ('ds_read', 'ds_read'), ('ds_load', 'ds_read'), ('ds_write', 'ds_write'),
                     ('ds_store', 'ds_write'), ('ds_', 'ds_other')):
        if op.startswith(pre):
            return cls
    if 'lds' in l and op.startswith(('global_load', 'buffer_load')):
        return 'dma'
    if op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
        return 'vmem_ld'
    if op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store')):
        return 'vmem_st'
    if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')):
        return 'trans'
    if op.startswith('v_pk_'):
        return 'valu_pk'
    if op.startswith('v_accvgpr'):
        return 'accmov'
    if op.startswith('v_'):
        return 'valu'
    for pre, cls in (('s_waitcnt', 'waitcnt'), ('s_barrier', 'barrier'), ('s_nop', 'nop'), ('s_cbranch', 'branch'), ('s_branch', 'branch'), ('s_', 'salu')):
        if op.startswith(pre):
            return cls
    return 'other'


def main(path, sub, min_mfma=4):
    lines = open(path).read().splitlines()
    starts = [n for n, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and sub in l]
    for i in starts:
        j = next(n for n in range(i, len(lines)) if lines[n].startswith('.Lfunc_end'))
        blocks, cur, lab = [], [], 'entry'
        for l in lines[i + 1:j]:
            st = l.strip()
            if re.match(r'^\.LBB\d+_\d+:', st):
                blocks.append((lab, cur))
                cur, lab = [], st[:-1]
            else:
                cur.append(l)
        blocks.append((lab, cur))
        print(lines[i][:-1], f"({j - i} lines)")
        for lab, b in blocks:
            c = collections.Counter(filter(None, map(classify, b)))
            if c.get('mfma', 0) >= min_mfma:
                print(f"   {lab:<12} {sum(c.values()):5d} instr: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4)
