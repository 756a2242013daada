"""The registers and the LDS of the BGZF kernels decide how many of their waves a CU holds - and with that their rate (round 6: a four-line
change took the writer's kernel from 256 to 258 registers, one workgroup per CU instead of two, 22 -> 40 ms per launch; nothing failed).
The device code of bgzf.hip is compiled to assembly (hipcc cross-compiles without a GPU) and the kernels' resource records are checked."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _resources(tmp_path, source):
    out = tmp_path / (source + ".s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", "-w",
                           "-I", os.path.join(ROOT, "elprep_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-o", str(out), os.path.join(ROOT, "elprep_amd", "csrc", source)])
    text = out.read_text()
    recs = {}
    for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)",
                         text, re.S):
        recs[m.group(2)] = dict(lds=int(m.group(1)), scratch=int(m.group(3)), vgpr=int(m.group(4)), spill=int(m.group(5)))

    def rec(part):
        names = [k for k in recs if part in k]
        assert len(names) == 1, (part, names)
        return recs[names[0]]
    return rec


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_bgzf_kernels_keep_their_occupancy(tmp_path):
    rec = _resources(tmp_path, "bgzf.hip")
    tok = rec("k_bgzf_tokens")     # 24 waves per CU: <= 80 registers (six waves per SIMD), <= 6.6 KB of LDS
    assert tok["vgpr"] <= 80 and tok["lds"] <= 6600 and tok["spill"] == 0 and tok["scratch"] == 0, tok
    dfl = rec("k_bgzf_deflate")    # two workgroups per CU: <= 256 registers, <= 80 KB of LDS
    assert dfl["vgpr"] <= 256 and dfl["lds"] <= 81920 and dfl["spill"] == 0 and dfl["scratch"] == 0, dfl
    res = rec("k_bgzf_resolve")    # one workgroup of 16 waves per CU: <= 128 registers; 128 KB of dynamic LDS + the CRC's tables
    assert res["vgpr"] <= 128 and res["lds"] <= 8192 and res["spill"] == 0 and res["scratch"] == 0, res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_main_line_kernels_keep_their_occupancy(tmp_path):
    """the step's big kernels: the count kernel's main variant (a 1024-thread workgroup per CU: four waves per SIMD, 128 registers, no
    spills), ApplyBQSR (three 512-thread workgroups per CU: six waves per SIMD, <= 85 registers), the front pass of mark duplicates
    (eight waves per SIMD: <= 64 registers), the pair bucket kernel (LDS tables: eight workgroups per CU)"""
    rec = _resources(tmp_path, "count3.hip")
    cnt = rec("k_bqsr_count3ILi5ELb0EE")
    assert cnt["vgpr"] <= 128 and cnt["spill"] == 0 and cnt["scratch"] == 0, cnt
    rec = _resources(tmp_path, "apply3.hip")
    for variant in ("k_bqsr_apply3ILb1EE", "k_bqsr_apply3ILb0EE"):
        ap = rec(variant)
        assert ap["vgpr"] <= 85 and ap["spill"] == 0 and ap["scratch"] == 0, ap
    rec = _resources(tmp_path, "markdup.hip")
    for variant in ("k_md_frontILb1EE", "k_md_frontILb0EE"):
        mf = rec(variant)
        assert mf["vgpr"] <= 64 and mf["lds"] <= 8192 and mf["spill"] == 0, mf
    pb = rec("k_pair_bucket")
    assert pb["vgpr"] <= 64 and pb["lds"] <= 20480 and pb["spill"] == 0, pb
