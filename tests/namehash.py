"""Python restatements of the two read-name hashes of the product (host/bam_reader.cpp hash_name / check_name,
csrc/bdx_bam_dev.h name_hash_* / name_check_*): what the host reader and the device-side decoder are both held to."""
M = (1 << 64) - 1


def hash_name(b):
    n = len(b)
    h = 0x9E3779B97F4A7C15 ^ n
    i = 0
    while i + 8 <= n:
        w = int.from_bytes(b[i:i + 8], "little")
        h = ((h ^ w) * 0xff51afd7ed558ccd) & M
        h ^= h >> 32
        i += 8
    w = int.from_bytes(b[i:], "little") if i < n else 0
    h = ((h ^ w) * 0xc4ceb9fe1a85ec53) & M
    h ^= h >> 29
    h = (h * 0xbf58476d1ce4e5b9) & M
    h ^= h >> 32
    return h


def _step(h, w):
    h ^= w
    h = ((h << 27) | (h >> 37)) & M
    return (h * 0x9FB21C651E98DF25 + 0x52DCE729) & M


def check_name(b):
    n = len(b)
    h = (0xD6E8FEB86659FD93 + n * 0x9FB21C651E98DF25) & M
    i = 0
    while i + 8 <= n:
        h = _step(h, int.from_bytes(b[i:i + 8], "little"))
        i += 8
    h = _step(h, int.from_bytes(b[i:], "little") if i < n else 0)
    h ^= h >> 33
    h = (h * 0xC2B2AE3D27D4EB4F) & M
    h ^= h >> 29
    h = (h * 0x165667B19E3779F9) & M
    return h ^ (h >> 32)
