"""TEST TOOLING: dynamic-Huffman DEFLATE blocks written bit by bit with RANDOMISED block headers
(RFC 1951 3.2.7): random prefix codes for the three alphabets and a randomised run-length
encoding of the code-length sequence (16 / 17 / 18 chosen at random where they apply, runs that
cross from the literal/length alphabet into the distance alphabet, optional overshoot behind the
last length, optional damage).  What the header decoders of the kernels have to agree on with
the oracle (src/inflate.c:1416-1520)."""
import random

CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
             4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class BitWriter:
    def __init__(self):
        self.acc = 0
        self.n = 0

    def put(self, v, nbits):  # LSB first
        self.acc |= (v & ((1 << nbits) - 1)) << self.n
        self.n += nbits

    def put_code(self, code, nbits):  # Huffman codes go MSB first
        for k in range(nbits - 1, -1, -1):
            self.put((code >> k) & 1, 1)

    def bytes(self):
        return self.acc.to_bytes((self.n + 7) // 8, "little")


def random_lengths(rng, n_syms, max_len):
    """code lengths of a complete prefix code over n_syms symbols (Kraft sum exactly 1)"""
    if n_syms == 1:
        return [1]
    lens = [1, 1]
    while len(lens) < n_syms:
        cand = [k for k, l in enumerate(lens) if l < max_len]
        k = rng.choice(cand)
        l = lens.pop(k) + 1
        lens += [l, l]
    rng.shuffle(lens)
    return lens


def canonical(lens):
    codes, code = {}, 0
    for l in range(1, 16):
        for s, sl in enumerate(lens):
            if sl == l:
                codes[s] = code
                code += 1
        code <<= 1
    return codes


def rle(rng, seq, overshoot):
    """[(symbol, extra bits, extra value)] for the code-length sequence, choices at random"""
    out, i, n = [], 0, len(seq)
    while i < n:
        v = seq[i]
        run = 1
        while i + run < n and seq[i + run] == v:
            run += 1
        opts = ["lit"]
        if v == 0 and run >= 3:
            opts += ["z17", "z18"] if run >= 11 else ["z17"]
        if i > 0 and seq[i - 1] == v and run >= 3:
            opts.append("rep")
        if overshoot and v == 0 and n - i < 30 and i + run == n:
            opts = ["over"]
        o = rng.choice(opts)
        if o == "lit":
            out.append((v, 0, 0))
            i += 1
        elif o == "z17":
            c = rng.randint(3, min(10, run))
            out.append((17, 3, c - 3))
            i += c
        elif o == "z18":
            c = rng.randint(11, min(138, run))
            out.append((18, 7, c - 11))
            i += c
        elif o == "rep":
            c = rng.randint(3, min(6, run))
            out.append((16, 2, c - 3))
            i += c
        else:  # a zero run that reaches behind the last length
            c = min(138, n - i + rng.randint(1, 20))
            out.append((18, 7, max(c, 11) - 11) if c >= 11 else (17, 3, max(c, 3) - 3))
            i = n
    return out


def tokens_for(rng, plain):
    """greedy toy LZ77 (enough to use length and distance codes)"""
    toks, i, n = [], 0, len(plain)
    while i < n:
        best = None
        if i >= 4 and rng.random() < 0.5:
            d = rng.choice([1, 2, 3, 4, min(i, 37), min(i, 300), i])
            l = 0
            while i + l < n and l < 258 and plain[i + l] == plain[i + l - d]:
                l += 1
            if l >= 3:
                best = (l, d)
        if best:
            toks.append(best)
            i += best[0]
        else:
            toks.append(plain[i])
            i += 1
    return toks


def make_block(seed, plain, final=True, overshoot=False, max_lit_len=15):
    """one dynamic block holding `plain`; returns the raw DEFLATE bytes (whole stream if final)"""
    rng = random.Random(seed)
    toks = tokens_for(rng, plain)
    used_lit, used_dist = {256}, set()
    enc = []
    for t in toks:
        if isinstance(t, tuple):
            l, d = t
            ls = max(k for k in range(29) if LEN_BASE[k] <= l)
            if l == 258:
                ls = 28
            ds = max(k for k in range(30) if DIST_BASE[k] <= d)
            used_lit.add(257 + ls)
            used_dist.add(ds)
            enc.append((257 + ls, LEN_EXTRA[ls], l - LEN_BASE[ls], ds, DIST_EXTRA[ds], d - DIST_BASE[ds]))
        else:
            used_lit.add(t)
            enc.append((t,))
    for _ in range(rng.randint(0, 12)):  # some symbols that never occur
        used_lit.add(rng.randrange(0, 286))
    for _ in range(rng.randint(0, 4)):
        used_dist.add(rng.randrange(0, 30))
    if not used_dist:
        used_dist.add(0)
    while len(used_lit) < 2:
        used_lit.add(rng.randrange(0, 256))
    ul, ud = sorted(used_lit), sorted(used_dist)
    ll = [0] * 286
    for s, l in zip(ul, random_lengths(rng, len(ul), max_lit_len)):
        ll[s] = l
    dl = [0] * 30
    for s, l in zip(ud, random_lengths(rng, len(ud), 15)):
        dl[s] = l
    hlit = max(257, max(ul) + 1 + (rng.randint(0, 3) if max(ul) < 282 else 0))
    hdist = max(ud) + 1
    seq = ll[:hlit] + dl[:hdist]
    items = rle(rng, seq, overshoot)
    cl_used = sorted({s for s, _, _ in items})
    if len(cl_used) == 1:
        cl_used = sorted(set(cl_used) | {(cl_used[0] + 1) % 19})
    cll = [0] * 19
    for s, l in zip(cl_used, random_lengths(rng, len(cl_used), 7)):
        cll[s] = l
    hclen = max(4, max(k for k in range(19) if cll[CL_ORDER[k]]) + 1)
    w = BitWriter()
    w.put(1 if final else 0, 1)
    w.put(2, 2)
    w.put(hlit - 257, 5)
    w.put(hdist - 1, 5)
    w.put(hclen - 4, 4)
    for k in range(hclen):
        w.put(cll[CL_ORDER[k]], 3)
    clc = canonical(cll)
    for s, nb, ev in items:
        w.put_code(clc[s], cll[s])
        w.put(ev, nb)
    lc, dc = canonical(ll), canonical(dl)
    for e in enc:
        w.put_code(lc[e[0]], ll[e[0]])
        if len(e) > 1:
            w.put(e[2], e[1])
            w.put_code(dc[e[3]], dl[e[3]])
            w.put(e[5], e[4])
    w.put_code(lc[256], ll[256])
    return w.bytes()


def payload(seed, n):
    rng = random.Random(seed * 7919 + 1)
    alpha = bytes(rng.sample(range(256), rng.choice([2, 5, 17, 60, 200])))
    out = bytearray()
    while len(out) < n:
        if out and rng.random() < 0.3:
            d = rng.randint(1, len(out))
            for _ in range(rng.randint(3, 40)):
                out.append(out[-d])
        else:
            out.append(rng.choice(alpha))
    return bytes(out[:n])


def cases(n, first=0):
    """[(raw, plain or None)]: None = the header is damaged or overshoots (the oracle decides)"""
    out = []
    for k in range(first, first + n):
        plain = payload(k, 40 + (k * 131) % 1500)
        over = k % 5 == 3
        raw = make_block(k, plain, overshoot=over, max_lit_len=15 if k % 3 else 9)
        if k % 7 == 6:  # damage one bit inside the header
            b = bytearray(raw)
            pos = 17 + (k * 37) % 200
            if pos // 8 < len(b):
                b[pos // 8] ^= 1 << (pos % 8)
            out.append((bytes(b), None))
        else:
            out.append((raw, None if over else plain))
    return out
