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
