"""The inline-asm main loops of conv_v2.hip / wgrad_v2.hip own VGPRs by number (v[100:255], v[96:255], v[82:255]) with the compiler
capped below by `amdgpu_num_vgpr`.  A hipcc point release or a small edit could make the compiler spill or touch those registers
silently; this test compiles the two sources (make, -save-temps=obj into csrc/build/) and audits the fresh device assembly
(tools/audit_asm.py): no scratch, no spills, no compiler-emitted instruction naming an owned register."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_audit_flags_a_compiler_instruction_on_an_owned_register():
    import audit_asm
    fake = """
_Z9my_kernelv:
	v_mov_b32 v3, v2
	;;#ASMSTART
	v_mfma_f32_32x32x16_bf16 v[200:215], v[100:103], v[104:107], v[200:215]
	;;#ASMEND
	v_add_u32 v120, v1, v2
	scratch_store_dword off, v1, s0
.end_amdhsa_kernel
"""
    st = audit_asm.audit_text(fake, [("my_kernel", 100)])
    f = audit_asm.findings(st)
    assert any("owned register" in x and "v120" in x for x in f)
    assert any("scratch" in x for x in f)
    clean = fake.replace("v_add_u32 v120, v1, v2", "v_add_u32 v20, v1, v2").replace("scratch_store_dword off, v1, s0", "")
    assert audit_asm.findings(audit_asm.audit_text(clean, [("my_kernel", 100)])) == []


def test_fresh_assembly_of_the_owned_register_kernels_is_clean():
    import audit_asm
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "cellsegmentation_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "-j8"], check=True, capture_output=True, timeout=1200)
    bad = audit_asm.audit_build()
    assert bad == [], "\n".join(bad)
