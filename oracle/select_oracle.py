"""select_oracle.py -- TEST INFRASTRUCTURE: numpy restatement of the library's counter-based generators.

The reference draws its training pixels with np.random.choice(population, N, replace=False) (train_nerf.py:185-189,
:219-221) and its sample jitter with torch.rand / torch.randn; a device-side implementation cannot reproduce those
host streams, so the library defines its own (documented in include/nerfhip.h) and this file restates them
independently so that tests can pin the kernels bit for bit:

* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), checked against
  the known-answer vectors published with Random123 (tests/test_oracle.py);
* the library's use of it: counter = (element_lo, element_hi, stream id, 0x9E3779B9), key = seed;
* the keyed permutation of [0, population): 6-round balanced Feistel network + cycle walking.
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = 0xFFFFFFFF


def philox4x32_10(counter, key):
    c = [int(x) & MASK32 for x in counter]
    k0, k1 = int(key[0]) & MASK32, int(key[1]) & MASK32
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK32, p1 & MASK32, ((p0 >> 32) ^ c[3] ^ k1) & MASK32, p0 & MASK32]
        k0, k1 = (k0 + W0) & MASK32, (k1 + W1) & MASK32
    return c


def nh_philox(seed, element, stream):
    return philox4x32_10([element & MASK32, (element >> 32) & MASK32, stream, 0x9E3779B9], [seed & MASK32, seed >> 32])


def uniform(seed, stream, first, n):
    """nerfhip_rng_fill(kind=uniform): top 24 bits of word 0."""
    return np.array([(nh_philox(seed, first + i, stream)[0] >> 8) / 16777216.0 for i in range(n)], np.float32)


def _mix(r, k):
    h = (r ^ k) & MASK32
    h = (h * 0x85EBCA6B) & MASK32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & MASK32
    h ^= h >> 16
    return h


def select_indices(seed, step, population, first, n):
    """nerfhip_select_indices."""
    a, b = nh_philox(seed, step, 4), nh_philox(seed, step, 5)
    keys = a + b[:2]
    bits = 2
    while (1 << bits) < population:
        bits += 2
    half = bits // 2
    mask = (1 << half) - 1
    out = np.empty(n, np.int64)
    for i in range(n):
        x = first + i
        while True:
            l, r = x >> half, x & mask
            for k in keys:
                l, r = r, l ^ (_mix(r, k) & mask)
            x = (l << half) | r
            if x < population:
                break
        out[i] = x
    return out
