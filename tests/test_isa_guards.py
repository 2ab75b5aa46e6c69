"""Not-GPU: compile the LDS-DMA GEMM to gfx950 assembly (hipcc cross-compiles without a GPU) and check the
properties its speed depends on, which a small source change can silently break:
  * no DMA instruction inside a waterfall loop (a descriptor or scalar offset that the compiler could not
    keep in SGPRs is legalised with a v_readfirstlane / s_cbranch_execnz loop per instruction: measured
    2.7x slower, profiles/r01_w_waterfall.txt);
  * no scratch (register spills) in any instantiation;
  * the wave-specialised instantiations fit two waves per SIMD (<= 256 VGPRs)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "progressive-stable-diffusion_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    text = ""
    for name in ("igemm_dma", "conv_halo"):
        out = tmp_path_factory.mktemp("isa") / f"{name}.s"
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                        "--cuda-device-only", "-S", os.path.join(CSRC, name + ".hip"), "-o", str(out)],
                       check=True, capture_output=True, timeout=600)
        text += out.read_text()
    return text


def _functions(asm_text):
    cur, body, out = None, [], {}
    for line in asm_text.splitlines():
        m = re.match(r"^(_ZN\S*(?:igemm_dma_kernel|conv3x3_halo_kernel)\S*):", line)
        if m:
            cur, body = m.group(1), []
        elif line.startswith(".Lfunc_end") and cur:
            out[cur], cur = body, None
        elif cur:
            body.append(line)
    return out


def test_dma_loads_are_not_in_waterfall_loops(asm):
    funcs = _functions(asm)
    assert len(funcs) >= 9 and any("halo" in f for f in funcs), sorted(funcs)
    for name, body in funcs.items():
        dma = [i for i, l in enumerate(body) if "buffer_load_dwordx4" in l and " lds" in l]
        assert dma, name
        for i in dma:       # a waterfall loop: v_readfirstlane of the divergent operand before, exec-loop branch behind
            looped = any("s_cbranch_execnz" in l for l in body[i + 1:i + 4])
            picked = any("v_readfirstlane" in l for l in body[max(0, i - 10):i])
            assert not (looped and picked), (name, body[i - 10:i + 4])


def test_no_scratch_and_two_waves_per_simd(asm):
    meta = re.findall(r"\.name:\s+(\S*(?:igemm_dma_kernel|conv3x3_halo_kernel)\S*)\s.*?\.private_segment_fixed_size:\s+(\d+).*?"
                      r"\.vgpr_count:\s+(\d+)", asm, flags=re.S)
    assert len(meta) >= 8
    for name, scratch, vgpr in meta:
        assert int(scratch) == 0, (name, scratch)
        if name.endswith("Lb1EEEv9IgemmArgs") or "halo" in name:   # 512 threads, two waves per SIMD
            assert int(vgpr) <= 256, (name, vgpr)
