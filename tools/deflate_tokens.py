#!/usr/bin/env python3
"""Token statistics of the BGZF members of a BAM (pure Python DEFLATE tokeniser, RFC 1951): literals, matches, the lengths'
and distances' distributions -- what the inflate kernels' costs depend on (tools/bamdec_probe.py measures them).
usage: deflate_tokens.py [file.bam] [members]      (without a file: a synthetic configs[1]-shaped BAM of 0.3 Mbp)"""
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self, data):
        self.d, self.p = data, 0

    def take(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[self.p >> 3] >> (self.p & 7)) & 1) << i
            self.p += 1
        return v


def table(lens):
    """canonical code: (length, code) -> symbol"""
    cnt = [0] * 16
    for l in lens:
        cnt[l] += 1
    cnt[0] = 0
    code, nxt = 0, [0] * 16
    for l in range(1, 16):
        code = (code + cnt[l - 1]) << 1
        nxt[l] = code
    t = {}
    for s, l in enumerate(lens):
        if l:
            t[(l, nxt[l])] = s
            nxt[l] += 1
    return t


def sym(b, t):
    code = 0
    for l in range(1, 16):
        code = (code << 1) | b.take(1)
        if (l, code) in t:
            return t[(l, code)]
    raise ValueError("bad code")


def tokens(payload):
    """[(0, byte) | (length, distance)] of one deflate stream"""
    b, out = Bits(payload), []
    while True:
        last, typ = b.take(1), b.take(2)
        if typ == 0:
            b.p = (b.p + 7) & ~7
            n = b.take(16); b.take(16)
            out += [(0, payload[(b.p >> 3) + i]) for i in range(n)]
            b.p += 8 * n
        else:
            if typ == 1:
                lit = table([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8); dst = table([5] * 32)
            else:
                hlit, hdist, hclen = b.take(5) + 257, b.take(5) + 1, b.take(4) + 4
                pl = [0] * 19
                for i in range(hclen):
                    pl[ORDER[i]] = b.take(3)
                pt, lens = table(pl), []
                while len(lens) < hlit + hdist:
                    s = sym(b, pt)
                    if s < 16: lens.append(s)
                    elif s == 16: lens += [lens[-1]] * (3 + b.take(2))
                    elif s == 17: lens += [0] * (3 + b.take(3))
                    else: lens += [0] * (11 + b.take(7))
                lit, dst = table(lens[:hlit]), table(lens[hlit:])
            while True:
                s = sym(b, lit)
                if s < 256: out.append((0, s))
                elif s == 256: break
                else:
                    s -= 257
                    if s < 8: ln = 3 + s
                    elif s == 28: ln = 258
                    else:
                        x = (s >> 2) - 1; ln = 3 + ((4 + (s & 3)) << x) + b.take(x)
                    d = sym(b, dst)
                    if d < 4: dist = 1 + d
                    else:
                        x = (d >> 1) - 1; dist = 1 + ((2 + (d & 1)) << x) + b.take(x)
                    out.append((ln, dist))
        if last:
            return out


def main():
    from breakdancer_amd import bamdec
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        from breakdancer_amd.bamwrite import write_bam
        from breakdancer_amd.synth import make_chromosome
        td = tempfile.mkdtemp(prefix="bdx_tok_", dir="/tmp")
        path = os.path.join(td, "syn.bam")
        write_bam(path, make_chromosome(length=300000, seed=1), ["chrS"], seed=3, level=1)
    n_members = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    image = np.fromfile(path, dtype=np.uint8)
    members = bamdec.scan_bgzf(image)
    members = members[members["inflated_len"] > 60000][1:1 + n_members]
    for m in members:
        pay = bytes(image[int(m["payload"]):int(m["payload"]) + int(m["payload_len"])])
        tk = tokens(pay)
        lit = sum(1 for t in tk if t[0] == 0)
        mt = [(l, d) for l, d in tk if l]
        ln = np.array([l for l, d in mt]); ds = np.array([d for l, d in mt])
        pos, per_group, far_pg = 0, {}, {}
        for l, d in tk:
            if l:
                per_group[pos >> 6] = per_group.get(pos >> 6, 0) + 1
                if d > 3582: far_pg[pos >> 6] = far_pg.get(pos >> 6, 0) + 1
            pos += l if l else 1
        print("member: %d B -> %d B; %d literals, %d matches (mean length %.1f, %d%% <= 16 bytes); distances: %d%% <= 256, %d%% <= 1024, %d%% <= 3582, %d%% > 3582, %d%% > 16384"
              % (len(pay), int(m["inflated_len"]), lit, len(mt), ln.mean(), 100 * (ln <= 16).mean(), 100 * (ds <= 256).mean(), 100 * (ds <= 1024).mean(),
                 100 * (ds <= 3582).mean(), 100 * (ds > 3582).mean(), 100 * (ds > 16384).mean()))
        fpg = np.array(list(far_pg.values()))
        print("   matches per 64 output bytes: mean %.1f max %d; far ones per 64 bytes holding any: mean %.1f, %d%% of far matches are third or later in their 64 bytes"
              % (np.mean(list(per_group.values())), max(per_group.values()), fpg.mean(), 100 * np.maximum(fpg - 2, 0).sum() / max(fpg.sum(), 1)))


if __name__ == "__main__":
    main()
