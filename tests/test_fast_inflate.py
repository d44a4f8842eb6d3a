"""CPU test of the host's own DEFLATE decoder (breakdancer_amd/host/fast_inflate.cpp) against zlib, block by block, on BGZF
files written with every compression level and strategy zlib offers over several kinds of data (bin/bdx-inflate-check
decodes every block both ways and compares)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, ROOT

CHECK = os.path.join(ROOT, "bin", "bdx-inflate-check")
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf(data, level, strategy, mem_level=8, chunk=65280):
    out = bytearray()
    for i in range(0, len(data), chunk):
        d = data[i:i + chunk]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
        comp = co.compress(d) + co.flush()
        if len(comp) + 26 > 65536:   # (incompressible at this setting: store it)
            co = zlib.compressobj(0, zlib.DEFLATED, -15)
            comp = co.compress(d) + co.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp + \
            struct.pack("<II", zlib.crc32(d) & 0xFFFFFFFF, len(d))
    return bytes(out) + EOF_BLOCK


def datasets():
    rng = np.random.default_rng(11)
    text = (b"The quick brown fox jumps over the lazy dog. " * 4000)
    ramp = bytes(range(256)) * 600
    rnd = rng.integers(0, 256, 180000, dtype=np.uint8).tobytes()
    few = rng.integers(0, 4, 200000, dtype=np.uint8).tobytes()                       # 2-bit alphabet: very short codes
    runs = b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400)) for _ in range(1500))  # distance-1 matches
    skew = rng.choice(np.arange(256, dtype=np.uint8), 250000, p=np.r_[0.7, np.full(255, 0.3 / 255)]).tobytes()  # 15-bit codes
    far = (rnd[:30000] + text[:2000]) * 5                                               # matches at distances up to 32 K
    bam = open(os.path.join(GOLDEN, "chr21", "NA19238_chr21_del_inv.bam"), "rb").read()
    return dict(text=text, ramp=ramp, random=rnd, few=few, runs=runs, skew=skew, far=far, one=b"x", bam_bytes=bam)


@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
def test_every_block_decodes_like_zlib(tmp_path, strategy):
    if not os.path.exists(CHECK):
        import __graft_entry__ as g
        g.build()
    paths = []
    for name, data in datasets().items():
        for level in (0, 1, 4, 6, 9):
            for chunk, mem in ((65280, 8), (7001, 1)):
                p = str(tmp_path / ("%s_%d_%d.bgzf" % (name, level, chunk)))
                open(p, "wb").write(bgzf(data, level, strategy, mem, chunk))
                paths.append(p)
    r = subprocess.run([CHECK] + paths, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode()
    assert "mismatches 0" in out
    blocks = int(out.split()[1])
    left = int(out.split("left_to_zlib")[1].split()[0])
    assert blocks > 500 and left <= len(paths)   # (only a payload that ends within 32 bytes of the file's end is left to zlib)


def test_golden_bams_decode_like_zlib():
    gd = os.path.join(GOLDEN, "chr21")
    r = subprocess.run([CHECK, os.path.join(gd, "NA19238_chr21_del_inv.bam"), os.path.join(gd, "NA19240_chr21_del_inv.bam")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and "mismatches 0 left_to_zlib 0" in r.stdout.decode()
