#!/usr/bin/env python3
"""LDS bank-conflict model of gfx950 (/opt/skills/guides/MI355X_MICROARCH.md, section LDS) for the access patterns of
the bf16 attention kernels: which row strides make the MFMA-fragment reads conflict-free.
  b128 : ds_read_b128, lane (r = lane & 15, g = lane >> 4) reads 16 B at row r, byte offset 16 g     (A-fragment rows)
  b64  : ds_read_b64,  lane (r, g) reads 8 B at row r, byte offset 8 g                              (paired-tile k order)
  tr   : ds_read_b64_tr_b16, lane group G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3 reads 8 B at
         row 4 G + q, byte offset 8 p                                                               (transposed operand)
Prints, per pattern, the extra LDS cycles (0 = conflict-free) for candidate row strides in bytes."""
import sys

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
HALVES = [list(range(0, 32)), list(range(32, 64))]


def extra_cycles(addr_of_lane, nbytes, groups, modulus=64):
    extra = 0
    for grp in groups:
        banks = {}
        for l in grp:
            a = addr_of_lane(l)
            for w in range(nbytes // 4):
                b = (a // 4 + w) % modulus
                banks.setdefault(b, set()).add(a // 4 + w)
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def b128(stride):
    return extra_cycles(lambda l: (l & 15) * stride + 16 * (l >> 4), 16, B128_GROUPS)


def b64(stride):
    return extra_cycles(lambda l: (l & 15) * stride + 8 * (l >> 4), 8, HALVES)


def tr(stride):
    return extra_cycles(lambda l: (4 * (l >> 4) + ((l & 15) >> 2)) * stride + 8 * (l & 3), 8, HALVES)


if __name__ == "__main__":
    for s in range(32, 321, 16):
        print(f"stride {s:4d} B: b128 {b128(s)}  b64 {b64(s)}  tr {tr(s)}")
