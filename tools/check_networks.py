#!/usr/bin/env python
"""Exhaustive 0/1-principle check of the comparator networks in flh_kernels.hip (merge8: lowest eight of two sorted
eight-lists, sorted).  A comparator network that handles every 0/1 input handles every input."""
import itertools

CEX = [(0, 4), (1, 5), (2, 6), (3, 7), (0, 2), (1, 3), (4, 6), (5, 7), (0, 1), (2, 3), (4, 5), (6, 7)]


def merge8(a, b):
    k = [min(a[j], b[7 - j]) for j in range(8)]
    for i, j in CEX:
        if k[i] > k[j]:
            k[i], k[j] = k[j], k[i]
    return k


def main():
    n = 0
    for za in range(9):
        for zb in range(9):
            a = [0] * za + [1] * (8 - za)  # sorted 0/1 lists
            b = [0] * zb + [1] * (8 - zb)
            assert merge8(a, b) == sorted(a + b)[:8], (a, b)
            n += 1
    # and on random integers with duplicates, for good measure
    import random

    rng = random.Random(1)
    for _ in range(20000):
        a = sorted(rng.randrange(12) for _ in range(8))
        b = sorted(rng.randrange(12) for _ in range(8))
        assert merge8(a, b) == sorted(a + b)[:8]
    print(f"merge8 ok ({n} 0/1 cases + 20000 random)")


if __name__ == "__main__":
    main()
