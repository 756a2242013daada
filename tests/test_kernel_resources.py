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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_bgzf_kernels_keep_their_occupancy(tmp_path):
    out = tmp_path / "bgzf.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", "-w",
                           "-I", os.path.join(ROOT, "elprep_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-o", str(out), os.path.join(ROOT, "elprep_amd", "csrc", "bgzf.hip")])
    text = out.read_text()
    recs = {}
    for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)",
                         text, re.S):
        recs[m.group(2)] = dict(lds=int(m.group(1)), scratch=int(m.group(3)), vgpr=int(m.group(4)), spill=int(m.group(5)))

    def rec(part):
        names = [k for k in recs if part in k]
        assert len(names) == 1, (part, names)
        return recs[names[0]]
    tok = rec("k_bgzf_tokens")     # 24 waves per CU: <= 80 registers (six waves per SIMD), <= 6.6 KB of LDS
    assert tok["vgpr"] <= 80 and tok["lds"] <= 6600 and tok["spill"] == 0 and tok["scratch"] == 0, tok
    dfl = rec("k_bgzf_deflate")    # two workgroups per CU: <= 256 registers, <= 80 KB of LDS
    assert dfl["vgpr"] <= 256 and dfl["lds"] <= 81920 and dfl["spill"] == 0 and dfl["scratch"] == 0, dfl
    res = rec("k_bgzf_resolve")    # one workgroup of 16 waves per CU: <= 128 registers; 128 KB of dynamic LDS + the CRC's tables
    assert res["vgpr"] <= 128 and res["lds"] <= 8192 and res["spill"] == 0 and res["scratch"] == 0, res
