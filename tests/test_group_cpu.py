"""The device group's communication library binds without a GPU: RCCL is found by dlopen and has every entry point csrc/group.hip
needs (the first run with more than one GPU must not be the first time this is tried)."""
import ctypes

from elprep_amd import _lib


def test_rccl_loads_and_has_the_entry_points():
    L = _lib.hip()
    assert L.elp_group_probe() == 0


def test_group_id_size_matches_rccl():
    # ELP_GROUP_ID_BYTES (include/elprep_hip.h) is static_assert-ed against sizeof(ncclUniqueId) in csrc/group.hip; here: the header
    # value the ctypes harness uses
    import re, os
    hdr = open(os.path.join(os.path.dirname(_lib.__file__), "..", "include", "elprep_hip.h")).read()
    assert int(re.search(r"#define ELP_GROUP_ID_BYTES (\d+)", hdr).group(1)) == 128
    rccl = open("/opt/rocm/include/rccl/rccl.h").read()
    assert int(re.search(r"#define NCCL_UNIQUE_ID_BYTES (\d+)", rccl).group(1)) == 128


def test_entry_points_reject_null_contexts_without_a_gpu():
    """argument checks come in front of any device call: a NULL context is an error code, not a crash (no compute without a GPU)"""
    L = _lib.hip()
    null = ctypes.c_void_p(0)
    n = ctypes.c_uint64(0)
    assert L.elp_copy_records(null, null, null, 0, -1, 0) != 0
    assert L.elp_emit_merged_bam(null, null, null, 0, ctypes.byref(n)) != 0
    assert L.elp_bqsr_lut_upload(null, 500, null, null) != 0
    assert L.elp_bqsr_apply(null, 500, null, null) != 0
    assert L.elp_group_init_transport(null, 0, 2, null, null) != 0
    assert L.elp_group_rank(null) == -1 and L.elp_group_size(null) == 0
