"""NumPy restatement of the counter-based index-set generator (TEST INFRASTRUCTURE ONLY).

The reference draws its sample indices from the global ``np.random`` stream
(src/ransac_algorithm.py:92,116; src/estimate_translation.py:65).  For hypothesis ranges that are
sharded over GPUs the library generates the sets on the device from (seed, hypothesis id) with
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
This file restates that generator; ``KNOWN_ANSWERS`` are the published known-answer vectors of
the Random123 distribution for philox4x32 with 10 rounds, which pin the restatement.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

# (counter[4], key[2], expected[4])
KNOWN_ANSWERS = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF),
     (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over equally shaped uint32 arrays; returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c1 = p1 & MASK
        c3 = p0 & MASK
        c0, c2 = n0, n2
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def index_sets(seed, first_id, nb, n, width=12):
    """int32[nb][width]: word w of id is philox(counter=(id_lo, id_hi, w // 4, 0), key=seed)[w % 4], index = word * n >> 32."""
    ids = np.uint64(first_id) + np.arange(nb, dtype=np.uint64)
    lo = ids & MASK
    hi = ids >> np.uint64(32)
    seed = int(seed) & (2**64 - 1)
    out = np.empty((nb, width), dtype=np.int32)
    zero = np.zeros(nb, dtype=np.uint64)
    for blk in range((width + 3) // 4):
        words = philox4x32_10(lo, hi, zero + np.uint64(blk), zero, seed & 0xFFFFFFFF, seed >> 32)
        for j in range(4):
            if 4 * blk + j < width:
                out[:, 4 * blk + j] = ((words[j].astype(np.uint64) * np.uint64(n)) >> np.uint64(32)).astype(np.int32)
    return out
