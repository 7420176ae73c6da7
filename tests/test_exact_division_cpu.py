"""The render kernel replaces the two jitter divisions (x + xi) / float(W) of rayview.cpp:80-81 by
float(double(n) * RN(1 / double(W))) (jitter_div in ray_amd/csrc/ray_amd.hip, with the argument why this is the IEEE
quotient). Here the same two expressions in numpy's IEEE arithmetic, on the numerators the kernel can produce
(a pixel coordinate plus a 24-bit fraction) and on adversarial ones."""
import numpy as np


def both(n, w):
    n = n.astype(np.float32)
    w32 = np.float32(w)
    inv = np.float64(1.0) / np.float64(w32)
    return (n.astype(np.float64) * inv).astype(np.float32), n / w32


def test_jitter_division_equals_ieee_division():
    rng = np.random.default_rng(7)
    for w in [1, 2, 3, 5, 7, 17, 33, 250, 360, 540, 640, 1080, 1920, 2160, 3840, 4096, 7680, 12345, 65535, 1 << 20, (1 << 24) - 1, (1 << 24) + 2, (1 << 31) - 1]:
        x = rng.integers(0, min(w, 1 << 24), 400_000).astype(np.float32)
        xi = (rng.integers(0, 1 << 24, x.size).astype(np.float32) * np.float32(2.0 ** -24))
        n = x + xi                                            # float(x) + u01(), as the kernel forms it
        a, b = both(n, w)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), w
    # adversarial: every float in a few binades, small odd divisors; zero and the smallest jitter
    for w in [3, 5, 7, 9, 11, 13, 360, 1080, 2160]:
        bits = np.arange(0x3f800000, 0x3f800000 + (1 << 22), dtype=np.uint32)
        for shift in (0, 5 << 23, 11 << 23):
            n = (bits + np.uint32(shift)).view(np.float32)
            a, b = both(n, w)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (w, shift)
        n = np.array([0.0, 2.0 ** -24, 2.0 ** -23, 1.0 - 2.0 ** -24], np.float32)
        a, b = both(n, w)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
