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


@pytest.fixture(scope="module")
def attn_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_attn") / "attention.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                    "--cuda-device-only", "-S", os.path.join(CSRC, "attention.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=900)
    return out.read_text()


def test_flash_attention_keeps_its_round3_shape(attn_asm):
    """What the d = 40 self-attention kernel's speed rests on (profiles/r03_y_mfma_valu_coissue.txt, r03_zf_pmc_flash_final.txt:
    MFMA and vector instructions do not overlap, so every vector instruction of the tile loop counts):
      * eight-wave workgroups at four waves per SIMD: <= 128 VGPRs, no scratch;
      * no lane exchange through the LDS crossbar (ds_bpermute / ds_swizzle) in the tile loops of the self-attention kernels;
      * one v_exp_f32 per score and NO fused multiply-add per score: the scale rides on Q and the reference maximum is the
        MFMA's C operand - at most as many v_fma_f32 as v_exp_f32 / 4 in the whole kernel (prologue and rescale branch)."""
    meta = dict((n, (int(s), int(v))) for n, s, v in re.findall(
        r"\.name:\s+(\S*flash_kernel\S*)\s.*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", attn_asm, flags=re.S))
    d40 = [n for n in meta if "flash_kernelILi40ELi2ELb1ELi8E" in n]
    assert len(d40) == 1, sorted(meta)
    assert meta[d40[0]] [0] == 0 and meta[d40[0]][1] <= 128, meta[d40[0]]
    cur, body, funcs = None, [], {}
    for line in attn_asm.splitlines():
        m = re.match(r"^(_ZN\S*flash_kernel\S*):", line)
        if m:
            cur, body = m.group(1), []
        elif line.startswith(".Lfunc_end") and cur:
            funcs[cur], cur = body, None
        elif cur:
            body.append(line)
    assert len(funcs) >= 7
    for name, body in funcs.items():      # (the one read-back of the row sums after the loop is a ds_bpermute per query fragment)
        text = "\n".join(body)
        assert text.count("ds_bpermute") <= 4 and "ds_swizzle" not in text, (name, text.count("ds_bpermute"))
    body = funcs[d40[0]]
    n_exp = sum(1 for l in body if "v_exp_f32" in l)
    n_fma = sum(1 for l in body if re.search(r"\bv_(fma|fmac|mac)_f32", l))
    assert n_exp >= 64 and n_fma * 4 <= n_exp, (n_exp, n_fma)
