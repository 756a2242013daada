"""The hand-pipelined kernels (elprep_amd/csrc/gload.hpp: global loads written as inline asm so that they stay in flight while the
previous block is worked on) rely on an invariant the compiler knows nothing about: between such a load and the hand-placed
s_waitcnt nothing reads or overwrites the load's destination registers.  The first version of apply3 broke it (a tied asm operand made
the compiler copy a record register two instructions ahead of the wait; two runs over 12 M reads differed).  This test compiles the
translation units to ISA (hipcc cross-compiles without a GPU) and checks the invariant on what the compiler actually emitted."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_pipeline as chk  # noqa: E402


def test_no_use_of_in_flight_registers():
    for name in ("count3.hip", "apply3.hip"):
        n_loads, bad = chk.check(os.path.join(ROOT, "elprep_amd", "csrc", name))
        assert n_loads > 0, name
        assert not bad, f"{name}: {bad[:5]}"


_PROBE = """
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void k(const uint32_t *p, uint32_t *o) {
  uint32_t v;
  asm volatile("global_load_dword %%0, %%1, off" : "=v"(v) : "v"(p + threadIdx.x) : "memory");
  %s
  o[threadIdx.x] = v + 1u;
}
"""


def test_the_checker_sees_a_use_in_front_of_the_wait(tmp_path):
    """the guard is not vacuous: a load whose result is used without the hand-placed wait is reported, the same kernel with the wait is not"""
    bad, good = tmp_path / "bad.hip", tmp_path / "good.hip"
    bad.write_text(_PROBE % "")
    good.write_text(_PROBE % 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" : "+v"(v));')
    n_bad, v_bad = chk.check(str(bad))
    n_good, v_good = chk.check(str(good))
    assert n_bad == 1 and v_bad, "a use of an in-flight register went unnoticed"
    assert n_good == 1 and not v_good, v_good
