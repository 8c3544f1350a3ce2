"""The pool of host threads behind de_program_create's per-tree passes (csrc/de_api.cpp HostPool), without a GPU: every item of a
range is visited exactly once whatever the thread count, a busy pool runs the ranges inline, and a fork()ed child — which inherits the
pool's bookkeeping but none of its threads — gets a pool of its own instead of waiting for workers that do not exist."""
import ctypes as C
import os
import signal
import threading

import pytest

from dynamicexpressions_jl_amd import api


def _selftest(lib, n):
    r = C.c_int32(0)
    got = lib.de_host_pool_selftest(n, C.byref(r))
    return got, r.value


def test_every_item_once_whatever_the_split():
    lib = api.library()
    for n in (0, 1, 31, 32, 63, 64, 1000, 4097, 100000):
        got, ranges = _selftest(lib, n)
        assert got == n
        assert 0 <= ranges <= 32
    got, ranges = _selftest(lib, 100000)
    # std::thread::hardware_concurrency() = the affinity mask's count on current glibc, the online processors on older ones
    counts = {os.cpu_count() or 1}
    if hasattr(os, "sched_getaffinity"):
        counts.add(len(os.sched_getaffinity(0)))
    assert ranges in {24 if n >= 48 else min(16, n) for n in counts} or os.environ.get("DE_HOST_THREADS")


def test_concurrent_callers_get_complete_results():
    """two host threads (two contexts) at once: one owns the pool, the other runs its ranges inline — both see every item once"""
    lib = api.library()
    res = []

    def work():
        for _ in range(50):
            res.append(_selftest(lib, 20000)[0])
    th = [threading.Thread(target=work) for _ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert res and all(r == 20000 for r in res)


@pytest.mark.skipif(not hasattr(os, "fork"), reason="needs fork")
def test_a_forked_child_starts_its_own_pool():
    lib = api.library()
    assert _selftest(lib, 50000)[0] == 50000  # the parent's pool exists now
    pid = os.fork()
    if pid == 0:
        signal.alarm(20)  # a child waiting for the parent's (absent) workers would hang: die loudly instead
        ok = _selftest(lib, 50000)[0] == 50000 and _selftest(lib, 70000)[0] == 70000
        os._exit(0 if ok else 3)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, f"child status {status}"
    assert _selftest(lib, 50000)[0] == 50000  # the parent's pool is unharmed
