"""Pure-Python Starknet Poseidon over Stark252 (test infrastructure): the Hades permutation with its 91 x 3 published round keys,
round by round, no shortcuts - the third, independent statement of the hash beside csrc/poseidon.h (compressed constants, lazy
reduction) and oracle/poseidon.hpp.  Pinned by public Starknet known answers in tests/test_poseidon.py."""
import hashlib

P = 2**251 + 17 * 2**192 + 1
FULL_HALF, PARTIAL = 4, 83
ROUNDS = 2 * FULL_HALF + PARTIAL
ROUND_KEYS = [[int(hashlib.sha256(f"Hades{3 * r + j}".encode()).hexdigest(), 16) % P for j in range(3)] for r in range(ROUNDS)]


def mix(s):
    t = (s[0] + s[1] + s[2]) % P
    return [(t + 2 * s[0]) % P, (t - 2 * s[1]) % P, (t - 3 * s[2]) % P]


def hades(state):
    s = list(state)
    for r in range(ROUNDS):
        s = [(s[i] + ROUND_KEYS[r][i]) % P for i in range(3)]
        if r < FULL_HALF or r >= FULL_HALF + PARTIAL:
            s = [pow(x, 3, P) for x in s]
        else:
            s[2] = pow(s[2], 3, P)
        s = mix(s)
    return s


def hash2(x, y):
    return hades([x, y, 2])[0]


def hash_single(x):
    return hades([x, 0, 1])[0]


def hash_many(values):
    v = list(values) + [1]
    if len(v) % 2:
        v.append(0)
    s = [0, 0, 0]
    for i in range(0, len(v), 2):
        s[0] = (s[0] + v[i]) % P
        s[1] = (s[1] + v[i + 1]) % P
        s = hades(s)
    return s[0]


def merkle_root(leaf_digests):
    level = list(leaf_digests)
    while len(level) > 1:
        level = [hash2(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
    return level[0]
