"""CPU checks: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/dadd_hip.h declares (no compute calls — there is no GPU here); the ctypes prototype table
covers the header; the parameter inventory reproduces the published SD-1.4 / DADD counts; the
product has no CPU path and says so loudly."""
import ctypes
import os
import re

import pytest
import torch

from progressive_stable_diffusion_amd import lib as L
from progressive_stable_diffusion_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "dadd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dadd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    path = L.build()
    assert os.path.exists(path)
    handle = ctypes.CDLL(path)
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), f"{n} declared in dadd_hip.h but not exported"
    assert set(names) == set(L.PROTOTYPES), set(names) ^ set(L.PROTOTYPES)
    assert handle.dadd_version() == 100


def test_igemm_desc_layout_matches_header(tmp_path):
    """The ctypes mirror of ``dadd_igemm_desc`` against the C compiler's view of include/dadd_hip.h (size and the
    offset of every field)."""
    import subprocess
    fields = [f[0] for f in L.IgemmDesc._fields_]
    assert fields[:8] == ["x", "x2", "w", "out", "partial", "bias", "rowvec", "residual"]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dadd_hip.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(dadd_igemm_desc));\n'
                   + "".join(f'  printf("%zu\\n", offsetof(dadd_igemm_desc, {f}));\n' for f in fields)
                   + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert ctypes.sizeof(L.IgemmDesc) == out[0]
    assert [getattr(L.IgemmDesc, f).offset for f in fields] == out[1:]


def test_status_codes_map_to_reference_exceptions():
    L.load()
    L.check(0)
    with pytest.raises(ValueError):
        L.check(L.DADD_EINVAL)
    with pytest.raises(RuntimeError):
        L.check(L.DADD_EHIP)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_has_no_cpu_path():
    from progressive_stable_diffusion_amd.backend import HipBackend
    from progressive_stable_diffusion_amd.config import default_config
    from progressive_stable_diffusion_amd.diffusion_module_ip import DiffusionModuleWithIP
    with pytest.raises(RuntimeError):
        HipBackend(torch.device("cuda:0"))
    with pytest.raises(RuntimeError):
        DiffusionModuleWithIP(default_config(), state_dict={}, device="cuda:0")


def test_parameter_inventory_matches_published_counts():
    u = W.unet_shapes()
    base = {k: s for k, s in u.items() if ".processor." not in k}
    assert W.count_params(base) == 859_520_964                      # SD-1.x UNet2DConditionModel
    assert W.count_params(u) - W.count_params(base) == 2 * 768 * 12480   # to_k_dis/to_v_dis, 16 sites
    assert W.count_params(W.vae_shapes()) == 83_653_863             # SD-1.x AutoencoderKL
    assert W.count_params(W.vae_shapes(encoder=False)) == 49_490_199
    c = W.conditioning_shapes()
    for prefix, n in (("ordinal_embedder", 20_096_256), ("image_projection", 14_976_768),
                      ("feature_purifier", 5_908_224)):
        assert W.count_params({k: s for k, s in c.items() if k.startswith(prefix)}) == n
    assert W.count_params(W.conditioning_shapes(projection_plus=False, purifier=False)) - 20_096_256 == 9_451_008
    assert len([k for k in u if k.endswith("anat_gate")]) == 16


def test_seeded_init_is_order_independent_and_reference_like():
    sh = W.conditioning_shapes()
    a = W.init_state_dict(sh, 3)
    keys = list(sh)[::-1]
    b = W.init_state_dict(sh, 3, keys=keys)
    assert all(torch.equal(a[k], b[k]) for k in sh)
    assert not torch.equal(a["ordinal_embedder.base"], W.init_state_dict(sh, 4)["ordinal_embedder.base"])
    d = a["ordinal_embedder.deltas"]
    assert d.shape == (3, 768) and 0.03 < d[0].mean() < 0.07 and d[2].mean() > d[0].mean()   # monotone init
    u = W.unet_shapes()
    ap = "unet.unet.mid_block.attentions.0.transformer_blocks.0.attn2"
    ks = [k for k in u if k.startswith(ap)]
    warm = W.init_state_dict(u, 0, keys=ks)
    assert torch.equal(warm[ap + ".processor.to_k_dis.weight"], warm[ap + ".to_k.weight"])   # routing_gates.py:308-314
    cold = W.init_state_dict(u, 0, keys=ks, warm_start_dis=False)
    assert not torch.equal(cold[ap + ".processor.to_k_dis.weight"], cold[ap + ".to_k.weight"])
    g = W.init_state_dict(u, 0, keys=ks, gates={"disease": (0.9, 0.1)})
    assert float(g[ap + ".processor.anat_gate"]) == pytest.approx(0.9)
