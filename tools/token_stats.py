#!/usr/bin/env python3
"""CPU-side diagnostic: the LZ77 token statistics of a PNG's IDAT stream (match lengths, distances, how many
matches overlap themselves, how long the dependency chains inside a 5 KB tile are).  Pure Python DEFLATE parser."""
import sys, zlib, collections

LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]


def build(lens):
    maxl = max(lens) if lens else 0
    cnt = [0] * (maxl + 2)
    for l in lens:
        if l: cnt[l] += 1
    code, nxt = 0, [0] * (maxl + 2)
    for b in range(1, maxl + 1):
        code = (code + cnt[b - 1]) << 1
        nxt[b] = code
    tab = {}
    for s, l in enumerate(lens):
        if l:
            tab[(l, nxt[l])] = s
            nxt[l] += 1
    return tab


def tokens(raw):
    pos = 0
    def bits(n):
        nonlocal pos
        v = 0
        for i in range(n):
            v |= ((raw[pos >> 3] >> (pos & 7)) & 1) << i
            pos += 1
        return v
    def sym(tab):
        nonlocal pos
        c, l = 0, 0
        while True:
            c = (c << 1) | ((raw[pos >> 3] >> (pos & 7)) & 1)
            pos += 1
            l += 1
            if (l, c) in tab: return tab[(l, c)]
    out = []
    while True:
        final, typ = bits(1), bits(2)
        if typ == 0:
            pos = (pos + 7) & ~7
            n = raw[pos >> 3] | (raw[(pos >> 3) + 1] << 8)
            pos += 32
            for k in range(n): out.append((0, raw[(pos >> 3) + k]))
            pos += 8 * n
        else:
            if typ == 1:
                lt = build([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
                dt = build([5] * 30)
            else:
                hl, hd, hc = bits(5) + 257, bits(5) + 1, bits(4) + 4
                cl = [0] * 19
                for i in range(hc): cl[[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15][i]] = bits(3)
                ct = build(cl)
                ls = []
                while len(ls) < hl + hd:
                    s = sym(ct)
                    if s < 16: ls.append(s)
                    elif s == 16: ls += [ls[-1]] * (3 + bits(2))
                    elif s == 17: ls += [0] * (3 + bits(3))
                    else: ls += [0] * (11 + bits(7))
                lt, dt = build(ls[:hl]), build(ls[hl:hl + hd])
            while True:
                s = sym(lt)
                if s < 256: out.append((0, s))
                elif s == 256: break
                else:
                    ln = LBASE[s - 257] + bits(LEXT[s - 257])
                    d = sym(dt)
                    out.append((ln, DBASE[d] + bits(DEXT[d])))
        if final: break
    return out


def main():
    data = open(sys.argv[1], "rb").read()
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 5104
    at, z = 8, b""
    while at + 8 <= len(data):
        ln = int.from_bytes(data[at:at + 4], "big")
        if data[at + 4:at + 8] == b"IDAT": z += data[at + 8:at + 8 + ln]
        at += 12 + ln
    toks = tokens(z[2:-4])
    nlit = sum(1 for t in toks if t[0] == 0)
    m = [t for t in toks if t[0]]
    outb = nlit + sum(t[0] for t in m)
    print(f"{sys.argv[1]}: {len(toks)} tokens, {nlit} literals, {len(m)} matches, {outb} bytes out, {outb / len(toks):.2f} bytes/token")
    print(f"  match length: mean {sum(t[0] for t in m) / len(m):.1f}; <=16: {sum(1 for t in m if t[0] <= 16) / len(m):.3f}; "
          f"self-overlapping (dist < len): {sum(1 for t in m if t[1] < t[0]) / len(m):.3f}; long (> 16) {sum(1 for t in m if t[0] > 16) / len(m):.3f}")
    dc = collections.Counter(t[1] for t in m)
    print("  commonest distances:", ", ".join(f"{d}: {c / len(m):.3f}" for d, c in dc.most_common(8)))
    bytes_by = collections.Counter()
    for ln, d in m:
        bytes_by["overlap" if d < ln else ("short" if ln <= 16 else "long")] += ln
    print("  output bytes by kind:", {k: round(v / outb, 3) for k, v in bytes_by.items()}, "literals", round(nlit / outb, 3))
    # dependency depth inside a tile: level of a match = 1 + max level of the matches (of the same tile) its source overlaps
    p, t0, lev = 0, 0, {}
    hist = collections.Counter()
    near = far = 0
    level_of_byte = bytearray(tile + 600)
    for ln, d in toks:
        if ln == 0:
            if p - t0 >= tile: t0 = p; level_of_byte = bytearray(tile + 600)
            level_of_byte[p - t0] = 0
            p += 1
            continue
        if p + ln - t0 > tile: t0 = p; level_of_byte = bytearray(tile + 600)
        s = p - d
        if s + min(ln, d) <= t0:
            far += 1; L = 0
        else:
            near += 1
            L = 1 + max(level_of_byte[max(s, t0) - t0:min(s + min(ln, d), p) - t0] or [0])
            L = min(L, 255)
            hist[min(L, 12)] += 1
        for k in range(ln): level_of_byte[p + k - t0] = L
        p += ln
    print(f"  tile {tile}: far {far / len(m):.3f}, near {near / len(m):.3f}; near matches by dependency level:",
          {k: round(v / max(near, 1), 3) for k, v in sorted(hist.items())})


main()
