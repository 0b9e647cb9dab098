#!/usr/bin/env python3
"""LDS bank-conflict check of the operand-tile images of csrc/bm_gemm.h under the bank model of
/opt/skills/guides/MI355X_MICROARCH.md (LDS section): a wave64 access is served in fixed lane groups; inside a
group every extra DISTINCT dword address on a busy bank costs one more LDS cycle; identical addresses broadcast.

Checks every fragment read of `read_frags` (x-major: ds_read_b128; k-major: ds_read_b32 / ds_read_b64) and every
register-path image store of `r2s` (ds_write_b128) for all tile geometries.  Exit status 1 if any access has a
conflict.  (The hardware agrees: SQ_LDS_BANK_CONFLICT = 0 in profiles/r2_*_pmc.json.)"""
import sys

B128_READ_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]
HALVES = [list(range(0, 32)), list(range(32, 64))]
B128_WRITE_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def extra_cycles(addr_of_lane, groups, width, nbanks):
    """addr_of_lane: dword address of the first dword each lane touches; width: dwords per lane"""
    extra = 0
    for grp in groups:
        per_bank = {}
        for lane in grp:
            for d in range(width):
                a = addr_of_lane[lane] + d
                per_bank.setdefault(a % nbanks, set()).add(a)
        extra += max(len(s) for s in per_bank.values()) - 1
    return extra


def fx(bk, row):
    return (row & 15) if bk == 64 else ((row >> 1) & 7)


def fk(s, row):
    return s * ((row >> 2) & 1)


def xm_off(bk, row, c4):
    return row * bk + ((c4 ^ fx(bk, row)) << 2)


def km_off(tx, s, row, col):
    return row * tx + (((col >> 2) ^ fk(s, row)) << 2) + (col & 3)


def check_xm_reads(bk, n_rows_of_16):
    bad = 0
    for wbase in range(n_rows_of_16):
        for m in range(bk // 16):
            addr = [xm_off(bk, wbase * 16 + (l & 15), 4 * m + (l >> 4)) for l in range(64)]
            bad += extra_cycles(addr, B128_READ_GROUPS, 4, 64)
    return bad


def check_km_reads(tx, bk, sub):
    """sub = MI or NJ (1: ds_read_b32, 2: ds_read_b64)"""
    s = 8 if sub == 2 else 4
    bad = 0
    for w in range(tx // (16 * sub)):
        for m in range(bk // 16):
            for j in range(4):
                addr = [km_off(tx, s, 16 * m + 4 * (l >> 4) + j, w * 16 * sub + sub * (l & 15)) for l in range(64)]
                bad += extra_cycles(addr, HALVES, sub, 32 if sub == 1 else 64)
    return bad


def check_xm_stores(tx, bk, nth):
    rc = bk // 4
    bad = 0
    for n in range(tx * bk // (4 * nth)):
        for w in range(nth // 64):
            addr = []
            for l in range(64):
                f = w * 64 + l + n * nth
                addr.append(xm_off(bk, f // rc, f % rc))
            bad += extra_cycles(addr, B128_WRITE_GROUPS, 4, 32)
    return bad


def check_km_stores(tx, bk, nth, sub):
    s = 8 if sub == 2 else 4
    rc = tx // 4
    bad = 0
    for n in range(tx * bk // (4 * nth)):
        for w in range(nth // 64):
            addr = []
            for l in range(64):
                f = w * 64 + l + n * nth
                row, c4 = f // rc, f % rc
                addr.append(row * tx + ((c4 ^ fk(s, row)) << 2))
            bad += extra_cycles(addr, B128_WRITE_GROUPS, 4, 32)
    return bad


# name: (WI, WJ, MI, NJ, BK)
GEOS = {
    'GeoAct': (2, 2, 2, 1, 64), 'GeoAct8': (2, 4, 1, 1, 64), 'GeoActS': (2, 2, 1, 1, 64),
    'GeoActS32': (2, 2, 1, 1, 32), 'GeoGrad': (2, 2, 2, 2, 64), 'GeoGrad8': (2, 4, 2, 1, 64),
}


def main():
    total = 0
    for name, (wi, wj, mi, nj, bk) in GEOS.items():
        ti, tj, nth = wi * 16 * mi, wj * 16 * nj, 64 * wi * wj
        rows = [
            ('P k-major read', check_km_reads(ti, bk, mi)),
            ('Q k-major read', check_km_reads(tj, bk, nj)),
            ('Q x-major read', check_xm_reads(bk, tj // 16)),
            ('P k-major store', check_km_stores(ti, bk, nth, mi)),
            ('Q k-major store', check_km_stores(tj, bk, nth, nj)),
            ('Q x-major store', check_xm_stores(tj, bk, nth)),
        ]
        if mi == 1:
            rows.append(('P x-major read', check_xm_reads(bk, ti // 16)))
            rows.append(('P x-major store', check_xm_stores(ti, bk, nth)))
        for what, bad in rows:
            print('%-10s %-16s extra LDS cycles: %d' % (name, what, bad))
            total += bad if 'read' in what else 0
    print('total extra read cycles:', total)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
