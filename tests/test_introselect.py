"""The GPU-side selection (csrc/introselect.hpp) must move data exactly like libstdc++'s std::nth_element."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sel():
    src = os.path.join(HERE, "host_helpers", "introselect_host.cpp")
    out = os.path.join(HERE, "host_helpers", "libintroselect_host.so")
    hdr = os.path.join(HERE, "..", "ucoslam-cv3_amd", "csrc", "introselect.hpp")
    if not os.path.exists(out) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(out):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src])
    return C.CDLL(out)


def _check(sel, oracle, keys, nth):
    n = len(keys)
    keys = np.asarray(keys, np.int32)
    ref = np.empty(n, np.int32)
    oracle.oracle_nth_element_perm(oracle_lib.P(keys), n, nth, oracle_lib.P(ref))
    # both product formulations: the sequential one and the pairing form the wave-cooperative GPU partition implements
    for fn in (sel.uh_host_nth_element, sel.uh_host_nth_element_pairing):
        packed = ((keys.astype(np.uint32) << 24) | np.arange(n, dtype=np.uint32)).astype(np.uint32)
        fn(oracle_lib.P(packed), n, nth)
        got = (packed & 0xFFFFFF).astype(np.int32)
        np.testing.assert_array_equal(got, ref)


def test_random_with_ties(sel, oracle):
    rng = np.random.default_rng(0)
    for _ in range(3000):
        n = int(rng.integers(1, 400))
        span = int(rng.choice([2, 5, 30, 200]))
        keys = rng.integers(7, 7 + span, n)
        _check(sel, oracle, keys, int(rng.integers(0, n)))


def test_structured_inputs(sel, oracle):
    for n in (1, 2, 3, 4, 5, 8, 64, 257, 1000):
        for keys in (np.arange(n), np.arange(n)[::-1], np.zeros(n, int), np.arange(n) % 3, (np.arange(n) * 7919) % 251):
            for nth in {0, n // 2, n - 1}:
                _check(sel, oracle, keys % 256, nth)


def test_depth_limit_path(sel, oracle):
    """Median-of-3 killer sequences exhaust the 2*log2(n) depth budget and take the heap-select branch."""
    for n in (64, 128, 500, 2000):
        k = n // 2
        a = np.zeros(n, int)
        for i in range(k):                      # classic Musser killer for median-of-3
            if i % 2 == 0:
                a[i] = i + 1
            else:
                a[i] = k + i + (0 if k % 2 else 1)
            a[k + i] = 2 * (i + 1)
        keys = (a.max() - a) * 255 // max(a.max(), 1)     # squash into [0,255] (introduces ties too)
        for nth in (0, n // 3, n - 1):
            _check(sel, oracle, keys, nth)
            _check(sel, oracle, (a % 256), nth)
