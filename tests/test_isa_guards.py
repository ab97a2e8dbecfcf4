"""CPU (hipcc cross-compiles without a GPU): properties of the generated gfx950 code that hand-written inline assembly relies on."""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT

CSRC = os.path.join(ROOT, "gaussian_process_liouville_equation_amd", "csrc")


@pytest.fixture(scope="module")
def predict_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "gple_predict.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "--cuda-device-only", "-S", f"-I{os.path.join(ROOT, 'include')}",
                    os.path.join(CSRC, "gple_predict.hip"), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_ZN4gple\S+):.*?\.end_amdhsa_kernel", text, flags=re.S | re.M):
        kernels[m.group(1)] = m.group(0)
    return kernels


def test_m0_belongs_to_the_lds_dma_in_the_kernels_that_do_not_restore_it(predict_asm):
    """rownormp_kernel and predict_fused256_kernel write M0 in front of every global_load_lds_dwordx4 and never restore it (csrc/gple_predict.hip): sound only
    while nothing else in those kernels reads or writes M0 — hipcc treats it as reserved and sets it in front of each of its own uses, of which there must be none"""
    seen = 0
    for name, body in predict_asm.items():
        if "rownormp_kernel" not in name and "predict_fused256_kernel" not in name:
            continue
        seen += 1
        uses = [l.strip() for l in body.splitlines() if re.search(r"\bm0\b", l) and not l.strip().startswith(";")]
        assert uses, name
        other = [l for l in uses if not re.match(r"s_(add_u32|mov_b32) m0, s\d+", l)]
        assert not other, (name, other[:5])
        assert body.count("global_load_lds_dwordx4") == len(uses), name  # one M0 write per DMA instruction
    assert seen == 5  # <4,4> and <2,8>, static and queue mode, and the fused small-n kernel


def test_the_pipelined_contraction_keeps_its_accumulators_in_registers(predict_asm):
    """no scratch in the MFMA kernels (a spill of the 128 accumulator registers costs more than any schedule gains), and the register budget of two waves per SIMD"""
    for name, body in predict_asm.items():
        if "rownormp_kernel" in name or "predict_fused256_kernel" in name or "rownorm2_kernel" in name:  # (rownorm3_kernel of round 3 spills 9 dwords)
            assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), name
            vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
            assert vgpr <= 256, (name, vgpr)
