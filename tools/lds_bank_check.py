"""LDS bank-conflict calculator for candidate image layouts (MI355X_MICROARCH.md, LDS section: lane groups and bank of a byte
address per instruction).  Used to choose the swizzles of the 16x16x32 operand reads of csrc/gemm.hip (gemm256q_kernel).
    python tools/lds_bank_check.py"""
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
HALVES = [list(range(32)), list(range(32, 64))]


def cycles(addrs, kind):
    """LDS cycles of one wave-instruction (conflict-free minimum: b128 4, b64 / tr 2)."""
    groups, nb, nbanks = {"b128": (B128_GROUPS, 4, 64), "b64": (HALVES, 2, 64), "tr": (HALVES, 2, 64)}[kind]
    tot = 0
    for g in groups:
        per_bank = {}
        for l in g:
            for k in range(nb):
                a = addrs[l] + 4 * k
                per_bank.setdefault((a // 4) % nbanks, set()).add(a // 4)
        tot += max(len(v) for v in per_bank.values())
    return tot


def kc_read16(sw, rb, s2):      # [rows][64 k] bf16, 128-B rows: lane l reads row rb + (l & 15), 16-byte chunk 4 s2 + (l >> 4)
    return [(rb + (l & 15)) * 128 + ((sw(rb + (l & 15), 4 * s2 + (l >> 4))) << 4) for l in range(64)]


def kc_read32(sw, rb, s):       # the 32x32x16 operand: row rb + (l & 31), chunk 2 s + (l >> 5)
    return [(rb + (l & 31)) * 128 + ((sw(rb + (l & 31), 2 * s + (l >> 5))) << 4) for l in range(64)]


def ks_read16(sw, rb, s2, hi):  # [64 k][128 rows] bf16, 256-B rows: group g = l >> 4 reads k rows 32 s2 + 8 g + 4 hi + q, columns rb .. rb + 15
    out = []
    for l in range(64):
        g, q, p = l >> 4, (l >> 2) & 3, l & 3
        kr = 32 * s2 + 8 * g + 4 * hi + q
        out.append(kr * 256 + (sw(kr, (rb >> 3) + (p >> 1)) << 4) + (p & 1) * 8)
    return out


def ks_read32(sw, rb, s, hi):
    out = []
    for l in range(64):
        h, gi, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
        kr = 16 * s + 8 * h + 4 * hi + q
        out.append(kr * 256 + (sw(kr, (rb >> 3) + 2 * gi + (p >> 1)) << 4) + (p & 1) * 8)
    return out


if __name__ == "__main__":
    kc_now = lambda r, c: c ^ ((r >> 1) & 7)
    ks_now = lambda kr, c: c ^ ((kr & 3) << 2)
    ks_b = lambda kr, c: c ^ (((kr & 3) << 2) | ((kr >> 2) & 3))      # cdna_hip_programming.md T10 image (b)
    ks_c = lambda kr, c: c ^ (((kr & 3) << 2) | ((kr >> 3) & 3))
    for name, sw in (("kc  c ^ ((r>>1)&7)   [shipped]", kc_now), ("kc  c ^ (r&7)", lambda r, c: c ^ (r & 7)),
                     ("kc  c ^ ((r>>1)&7 ^ ((r&1)<<2))", lambda r, c: c ^ (((r >> 1) & 7) ^ ((r & 1) << 2)))):
        w16 = max(cycles(kc_read16(sw, rb, s2), "b128") for rb in range(0, 128, 16) for s2 in range(2))
        w32 = max(cycles(kc_read32(sw, rb, s), "b128") for rb in range(0, 128, 32) for s in range(4))
        print(f"{name:40s} 16x16x32 row read: {w16} cycles (4 = conflict-free)   32x32x16 row read: {w32}")
    for name, sw in (("ks  c ^ ((kr&3)<<2)   [shipped]", ks_now), ("ks  T10 (b): c ^ (((kr&3)<<2)|((kr>>2)&3))", ks_b),
                     ("ks  c ^ (((kr&3)<<2)|((kr>>3)&3))", ks_c)):
        w16 = max(cycles(ks_read16(sw, rb, s2, hi), "tr") for rb in range(0, 128, 16) for s2 in range(2) for hi in range(2))
        w32 = max(cycles(ks_read32(sw, rb, s, hi), "tr") for rb in range(0, 128, 32) for s in range(4) for hi in range(2))
        print(f"{name:44s} 16x16x32 transposed read: {w16} cycles (2 = conflict-free)   32x32x16: {w32}")
