#!/usr/bin/env python3
"""LDS bank-conflict model of the fused ResBlock kernel's B-operand reads (ds_read_b128), following the lane
groups / banking of MI355X_MICROARCH.md section LDS: a wave64 ds_read_b128 is serviced in 4 groups of 16 lanes,
bank = (addr / 4) mod 64; every extra distinct address on a busy bank inside a group costs one LDS cycle.
Prints conflict cycles per workgroup for a swizzle function, for conv1 (flattened T region) and conv2."""
import sys

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def read_cycles(addrs):
    """addrs[lane] = byte address of a 16-byte read. Returns (cycles, conflict_cycles)."""
    tot = 0
    for g in GROUPS:
        per_bank = {}
        for l in g:
            a = addrs[l]
            for d in range(4):
                per_bank.setdefault(((a >> 2) + d) & 63, set()).add((a >> 2) + d)
        tot += max(len(s) for s in per_bank.values())
    return tot, tot - 4


def sim(TH, TW, ROWP, NW, key, flat1=True):
    TR, TC = TH + 2, TW + 2
    NT1 = (TR * TC + 15) // 16
    PT1 = (NT1 + NW - 1) // NW
    NPT2 = TH * TW // 16
    PT2 = (NPT2 + NW - 1) // NW
    c1 = c2 = r1 = r2 = 0
    for tile in range(NW * PT1):
        for tap in range(9):
            dy, dx = divmod(tap, 3)
            for half in range(2):
                addrs = []
                for lane in range(64):
                    pl, q = lane & 15, lane >> 4
                    pix = min(tile * 16 + pl, TR * TC - 1)
                    ti, tj = divmod(pix, TC)
                    row, col = ti + dy, tj + dx
                    P = row * ROWP + col
                    addrs.append(P * 128 + (((q ^ key(row, col, P)) << 4) ^ (half << 6)))
                cyc, cf = read_cycles(addrs)
                c1 += cf; r1 += 1
    for tile in range(NW * PT2):
        t = min(tile, NPT2 - 1)
        prow, pcol = divmod(t, TW // 16)
        for tap in range(9):
            dy, dx = divmod(tap, 3)
            for half in range(2):
                addrs = []
                for lane in range(64):
                    pl, q = lane & 15, lane >> 4
                    row, col = prow + dy, pcol * 16 + pl + dx
                    P = row * ROWP + col
                    addrs.append(P * 128 + (((q ^ key(row, col, P)) << 4) ^ (half << 6)))
                cyc, cf = read_cycles(addrs)
                c2 += cf; r2 += 1
    return c1, r1, c2, r2


KEYS = {
    "current (P>>1)&7": lambda row, col, P: (P >> 1) & 7,
    "2*((col>>1)&3)": lambda row, col, P: 2 * ((col >> 1) & 3),
    "2*((P>>1)&3)": lambda row, col, P: 2 * ((P >> 1) & 3),
    "2*(((col>>1)+row)&3)": lambda row, col, P: 2 * (((col >> 1) + row) & 3),
    "2*(((col>>1)^row)&3)": lambda row, col, P: 2 * (((col >> 1) ^ row) & 3),
}

if __name__ == "__main__":
    for (TH, TW, ROWP, NW) in [(12, 48, 64, 8), (6, 96, 112, 8), (4, 32, 48, 4), (12, 48, 58, 8)]:
        print(f"tile {TH}x{TW} pitch {ROWP} waves {NW}")
        for name, k in KEYS.items():
            c1, r1, c2, r2 = sim(TH, TW, ROWP, NW, k)
            print(f"  {name:24s} conv1 {c1:6d} conflict cycles / {r1} reads ({c1 / (4 * r1):.2f} per read-cycle)   conv2 {c2:6d} / {r2} ({c2 / (4 * r2):.2f})")
