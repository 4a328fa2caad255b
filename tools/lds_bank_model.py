#!/usr/bin/env python3
"""LDS bank-conflict model for gfx950 (MI355X_MICROARCH.md, LDS table): checks the K / V tile layouts of csrc/attention.hip's
attn2 kernel.  ds_read_b128: four 16-lane groups, bank = (addr/4) % 64; ds_read_b64_tr_b16: two 32-lane groups, same banking."""
import sys

B128_GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
B64_GROUPS = [list(range(32)), list(range(32, 64))]


def cycles(addrs, groups, nbytes):
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs[l]
            for w in range(nbytes // 4):
                banks.setdefault(((a >> 2) + w) % 64, set()).add(a + 4 * w)
        tot += max(len(v) for v in banks.values())
    return tot


def k_addr(D, lane, ks, kb, swz):
    NCH = D // 8
    l31, hh = lane & 31, lane >> 5
    row = kb * 32 + l31
    c = 2 * ks + hh
    if c >= NCH:
        return 1 << 20      # const cell (broadcast)
    return row * D * 2 + swz(c, row, NCH) * 16


def v_addr(D, lane, kb, s, hi, db, swz):
    NCH = D // 8
    hh = lane >> 5
    q, p, g1 = (lane & 15) >> 2, lane & 3, (lane >> 4) & 1
    key = kb * 32 + 16 * s + 8 * hi + 4 * hh + q
    d = db * 32 + 16 * g1 + 4 * p
    if d >= D:
        return (1 << 20) + (8 if d == D else 16)
    c = d >> 3
    return key * D * 2 + swz(c, key, NCH) * 16 + (d & 7) * 2


def report(D, kswz, vswz):
    NKS, NDB = (D + 15) // 16, (D + 31) // 32
    kw = [cycles([k_addr(D, l, ks, kb, kswz) for l in range(64)], B128_GROUPS, 16) for kb in range(2) for ks in range(NKS)]
    vw = [cycles([v_addr(D, l, kb, s, hi, db, vswz) for l in range(64)], B64_GROUPS, 8)
          for kb in range(2) for s in range(2) for hi in range(2) for db in range(NDB)]
    print(f"D={D}: K b128 cycles per read (ideal 4): {sorted(set(kw))}   V tr_b64 cycles per read (ideal 2): {sorted(set(vw))}")


ident = lambda c, r, n: c
xor7 = lambda c, r, n: c ^ (r & 7)
rot1 = lambda c, r, n: (c + ((r >> 3) & 1)) % n
rotq = lambda c, r, n: (c + ((r >> 2) & 1)) % n
if __name__ == "__main__":
    report(40, ident, ident)
    report(64, xor7, xor7)
    report(64, xor7, lambda c, r, n: c ^ ((r >> 1) & 7))
    report(80, rot1, rot1)
    report(80, rot1, ident)
    report(80, rot1, rotq)
    report(160, ident, ident)
