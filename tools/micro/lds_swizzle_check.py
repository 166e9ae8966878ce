"""Exhaustive bank-conflict check of the LDS layouts read with ds_read_b128 (lane groups and bank rule from
/opt/skills/guides/MI355X_MICROARCH.md: four groups of 16 lanes, bank = (byte address / 4) mod 64): prints the LDS cycles of
one wave-instruction per layout (4 = conflict-free).  Runs anywhere (no GPU)."""
G0 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
G1 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
GROUPS = [G0, G1, [l + 32 for l in G0], [l + 32 for l in G1]]


def cycles(addr_of_lane):
    total = 0
    for grp in GROUPS:
        banks = {}
        for lane in grp:
            a = addr_of_lane(lane)
            assert a % 16 == 0
            for b in range(4):
                banks.setdefault((a // 4 + b) % 64, set()).add(a)
        total += max(len(v) for v in banks.values())
    return total


def w2_swz(row):
    return ((row >> 1) & 1) | (((row >> 3) & 3) << 1)


if __name__ == "__main__":
    # swin_mlp.hip, W1 chunk: rows of 2C bytes, fragment (j, ks): row 16 j + fr, chunk (4 ks + quad) ^ fr
    for C in (128, 256):
        worst = max(cycles(lambda l, j=j, ks=ks: (16 * j + (l & 15)) * 2 * C + (((4 * ks + (l >> 4)) ^ (l & 15)) << 4))
                    for j in range(4) for ks in range(C // 32))
        print(f"swin_mlp W1 chunk, C = {C}: {worst} cycles per ds_read_b128")
        # W2 chunk: 128-byte rows, fragment (jo, s): row 32 (jo >> 1) + 8 (fr >> 2) + 4 (jo & 1) + (fr & 3), chunk (4 s + quad) ^ swz(row)
        def w2(l, jo, s):
            fr, quad = l & 15, l >> 4
            n = 32 * (jo >> 1) + 8 * (fr >> 2) + 4 * (jo & 1) + (fr & 3)
            return n * 128 + (((4 * s + quad) ^ w2_swz(n)) << 4)
        worst = max(cycles(lambda l, jo=jo, s=s: w2(l, jo, s)) for jo in range(C // 16) for s in range(2))
        print(f"swin_mlp W2 chunk, C = {C}: {worst} cycles per ds_read_b128")
    # attention.hip / swin.hip V^T: rows of 2 TP + 32 bytes, fragment (ct, u): row ct 16 + fr, byte (32 u + 8 quad) 2
    for tp in (64, 224, 256, 320):
        s = tp * 2 + 32
        worst = max(cycles(lambda l, ct=ct, u=u: (ct * 16 + (l & 15)) * s + (32 * u + 8 * (l >> 4)) * 2) for ct in range(2) for u in range(tp // 32))
        print(f"V^T rows of {s} bytes ({tp} keys): {worst} cycles per ds_read_b128")
