"""-m gpu: every HIP kernel, called through the C ABI (ctypes), against the plain-torch fp32
reference of the same op (tests/torch_backend.py) on identical seeded inputs.

Tolerances (stated per test): operands are fp16, accumulation fp32; the reference computes in fp32
from the same fp16 operands and rounds once to fp16, so the expected difference is one fp16
rounding of the result (rel 2^-11 = 4.9e-4) plus accumulation-order noise.
"""
import math

import pytest
import torch

from tests.torch_backend import TorchRefBackend

pytestmark = pytest.mark.gpu

F16, F32 = torch.float16, torch.float32
REF = TorchRefBackend()


@pytest.fixture(scope="module")
def hip():
    from progressive_stable_diffusion_amd.backend import HipBackend
    return HipBackend(torch.device("cuda:0"))


def rnd(shape, seed, scale=1.0, dtype=F16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def dev(hip, t):
    return None if t is None else hip.to_device(t)


def close(got, ref, atol, rtol, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} off, max err "
                                 f"{err.max().item():.4e} at ref {ref.flatten()[err.argmax()].item():.4e}")


def run_igemm(hip, x, w, out_shape, **kw):
    tens = {k: kw.pop(k, None) for k in ("x2", "bias", "rowvec", "residual")}
    in_launch = kw.pop("in_launch_combine", True)     # split-K slabs combined by the last arriver
    repeats = kw.pop("repeats", 1)
    splitk = kw.get("splitk", 1)
    o_ref = torch.zeros(out_shape, dtype=F16)
    REF.igemm(x, w, o_ref, **tens, **kw)
    o = hip.zeros(out_shape, F16)
    m = out_shape[0] * out_shape[1] * out_shape[2]
    partial = hip.zeros((splitk * m * w.shape[0],), F32) if splitk > 1 else None
    counters = hip.zeros((4096,), torch.int32) if (splitk > 1 and in_launch) else None
    xd, wd, td = dev(hip, x), dev(hip, w), {k: dev(hip, v) for k, v in tens.items()}
    for _ in range(repeats):      # >1: the tickets must be back at zero after every launch
        hip.igemm(xd, wd, o, **td, partial=partial, counters=counters, **kw)
    hip.synchronize()
    if counters is not None:
        assert int(counters.abs().sum().item()) == 0, "split-K tickets not reset"
    return o, o_ref


@pytest.mark.parametrize("m,n,k,tile", [(200, 320, 320, 0), (130, 256, 128, 128), (64, 1280, 768, 160),
                                        (513, 960, 640, 0)])
def test_igemm_linear(hip, m, n, k, tile):
    x, w = rnd((1, m, 1, k), 1), rnd((n, k), 2, 1 / math.sqrt(k))
    bias = rnd((n,), 3, 0.1, F32)
    o, o_ref = run_igemm(hip, x, w, (1, m, 1, n), bias=bias, flags=1, tile_n=tile)
    close(o, o_ref, 2e-3, 2e-3, f"linear {m}x{n}x{k}")


@pytest.mark.parametrize("cin,cout,h,stride,ups,pad", [(320, 320, 16, 1, 0, 1), (128, 256, 12, 1, 0, 1),
                                                       (320, 320, 16, 2, 0, 1), (640, 640, 8, 1, 1, 1),
                                                       (128, 128, 16, 2, 0, 0)])
def test_igemm_conv3x3(hip, cin, cout, h, stride, ups, pad):
    b = 2
    ho = h * 2 if ups else (h // 2 if stride == 2 else h)
    x, w = rnd((b, h, h, cin), 4), rnd((cout, 9 * cin), 5, 1 / math.sqrt(9 * cin))
    bias, rowvec = rnd((cout,), 6, 0.1, F32), rnd((b, cout), 7, 0.3, F32)
    res = rnd((b, ho, ho, cout), 8)
    o, o_ref = run_igemm(hip, x, w, (b, ho, ho, cout), bias=bias, rowvec=rowvec, residual=res, taps=9,
                         stride=stride, ups=ups, pad=pad, flags=7)
    close(o, o_ref, 3e-3, 2e-3, f"conv3x3 cin{cin} s{stride} u{ups} p{pad}")


@pytest.mark.parametrize("tile_m,tile_n", [(64, 64), (64, 128), (64, 160), (128, 128), (128, 160)])
@pytest.mark.parametrize("case", ["linear_ragged", "linear_res_splitk", "conv3x3", "conv_stride2_concat"])
def test_igemm_dma_tile_shapes(hip, tile_m, tile_n, case):
    """Every (rows, columns) instantiation of the wave-specialised LDS-DMA kernel — 64-row tiles for the short GEMMs
    of the small maps, 128-row tiles elsewhere — on ragged M / N, residual + bias + time row, split-K through the
    finish kernel, a 3x3 gather with padding and a strided 3x3 over a skip-concat."""
    kw = dict(tile_m=tile_m, tile_n=tile_n)
    if case == "linear_ragged":
        m, n, k = 1000, 832 if tile_n != 160 else 800, 640          # M, N not multiples of the tile
        x, w = rnd((1, m, 1, k), 60), rnd((n, k), 61, 1 / math.sqrt(k))
        o, o_ref = run_igemm(hip, x, w, (1, m, 1, n), bias=rnd((n,), 62, 0.1, F32), flags=1, **kw)
    elif case == "linear_res_splitk":
        b, hw, n, k = 2, 256, 1280, 2560
        x, w = rnd((b, hw, 1, k), 63), rnd((n, k), 64, 1 / math.sqrt(k))
        o, o_ref = run_igemm(hip, x, w, (b, hw, 1, n), bias=rnd((n,), 65, 0.1, F32), residual=rnd((b, hw, 1, n), 66),
                             rowvec=rnd((b, n), 67, 0.3, F32), flags=7, splitk=4, in_launch_combine=False, **kw)
    elif case == "conv3x3":
        b, h, cin, cout = 2, 12, 128, 320
        x, w = rnd((b, h, h, cin), 68), rnd((cout, 9 * cin), 69, 1 / math.sqrt(9 * cin))
        o, o_ref = run_igemm(hip, x, w, (b, h, h, cout), bias=rnd((cout,), 70, 0.1, F32), taps=9, pad=1, flags=1, **kw)
    else:
        b, h, c1, c2, n = 2, 16, 192, 128, 640
        x, x2 = rnd((b, h, h, c1), 71), rnd((b, h, h, c2), 72)
        w = rnd((n, 9 * (c1 + c2)), 73, 1 / math.sqrt(9 * (c1 + c2)))
        o, o_ref = run_igemm(hip, x, w, (b, h // 2, h // 2, n), x2=x2, bias=rnd((n,), 74, 0.1, F32), taps=9, stride=2,
                             pad=1, flags=1, **kw)
    close(o, o_ref, 3e-3, 2e-3, f"dma tile {tile_m}x{tile_n} {case}")


@pytest.mark.parametrize("tile_m,tile_n,tune", [(64, 64, 0), (64, 128, 0), (64, 160, 0), (128, 128, 0), (128, 160, 0),
                                                (128, 160, 64), (64, 160, 32)])
@pytest.mark.parametrize("geglu", [False, True])
def test_igemm_layernorm_fold(hip, tile_m, tile_n, tune, geglu):
    """DADD_EPI_LNFOLD on every kernel that can carry it (LDS-DMA tiles, persistent ring = tune 64, register-staged
    = tune 32: served by the LDS-DMA kernel, the only one with the statistics): out = rstd (x (gamma o W)^T - mu c1) + (W beta + b) with mu / rstd accumulated from the A
    fragments, against LayerNorm -> Linear in fp32 torch.  Rows with a large common offset (|mean| = 3 sigma)
    exercise the cancellation of the mean term.  Tolerance: one fp16 rounding of the result + fp16 weights."""
    from progressive_stable_diffusion_amd import engine as E
    from progressive_stable_diffusion_amd import lib as L
    import torch.nn.functional as Fn
    if geglu and tile_n != 128:
        pytest.skip("the GEGLU epilogue runs on 128-column tiles")
    m, k = (20000 if tune != 64 else 40000), 640
    n = 1024 if tile_n != 160 else 960
    if geglu:
        n = 1024
    g = torch.Generator().manual_seed(80)
    x = (torch.randn(1, m, 1, k, generator=g) + 3.0 * torch.randn(1, m, 1, 1, generator=g)).to(F16)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    b = torch.randn(n, generator=g) * 0.1
    gamma, beta = 1.0 + 0.2 * torch.randn(k, generator=g), 0.2 * torch.randn(k, generator=g)
    ref = Fn.linear(Fn.layer_norm(x.float(), (k,), gamma, beta, 1e-5), w.to(F16).float(), b)
    if geglu:
        hid, gate = ref.chunk(2, dim=-1)
        ref = hid * Fn.gelu(gate)
    w16, c1, bias = E.fold_layernorm(w, b, gamma, beta)
    if geglu:
        idx = E.geglu_interleave(torch.arange(n)[:, None].float(), torch.zeros(n))[0][:, 0].long()
        w16, c1, bias = w16[idx], c1[idx], bias[idx]
    o = hip.zeros((1, m, 1, n // 2 if geglu else n), F16)
    hip.igemm(dev(hip, x), dev(hip, w16.contiguous()), o, bias=dev(hip, bias.contiguous()),
              flags=L.EPI_BIAS | L.EPI_LNFOLD | (L.EPI_GEGLU if geglu else 0) | tune, tile_m=tile_m, tile_n=tile_n,
              ln_c1=dev(hip, c1.contiguous()))
    hip.synchronize()
    close(o, ref, 6e-3, 6e-3, f"ln fold {tile_m}x{tile_n} tune{tune} geglu{geglu}")
    o2 = hip.zeros(tuple(o.shape), F16)                  # run to run: bit-identical (no race in the statistics)
    for _ in range(3):
        hip.igemm(dev(hip, x), dev(hip, w16.contiguous()), o2, bias=dev(hip, bias.contiguous()),
                  flags=L.EPI_BIAS | L.EPI_LNFOLD | (L.EPI_GEGLU if geglu else 0) | tune, tile_m=tile_m, tile_n=tile_n,
                  ln_c1=dev(hip, c1.contiguous()))
        hip.synchronize()
        assert torch.equal(o2, o)
    if not geglu:
        with pytest.raises(ValueError):       # a folded LayerNorm needs whole rows of A: no split-K
            hip.igemm(dev(hip, x), dev(hip, w16.contiguous()), o, flags=L.EPI_LNFOLD, splitk=2,
                      partial=hip.zeros((2 * m * n,), F32), ln_c1=dev(hip, c1.contiguous()))


@pytest.mark.parametrize("ptile,ptune", [((64, 64), 0), ((64, 160), 0), ((128, 160), 0), ((128, 128), 0), ((64, 160), 32),
                                         ((128, 128), 32)])
@pytest.mark.parametrize("ctile,ctune,geglu", [((128, 160), 64, False), ((128, 160), 32, False), ((128, 128), 32, True),
                                               ((128, 128), 64, True), ((64, 64), 0, False)])
def test_igemm_layernorm_statistics_from_producer(hip, ptile, ptune, ctile, ctune, geglu):
    """DADD_EPI_LNSTAT -> ln_stats_in: the producing linear (bias + residual epilogue) writes the row partials of its
    rounded output, [N / (tile_n/2)][M][2]; the consumer folds LayerNorm with them on ANY kernel (LDS-DMA, persistent
    ring, register-staged = tune 32).  Checks: partials == sums over the stored fp16 output (fp32 order: 1e-5 rel.),
    consumer == LayerNorm -> Linear (-> GEGLU) of the producer's output in fp32 torch, run-to-run bit identity, and the
    contract errors (wrong part count, GEGLU producer)."""
    from progressive_stable_diffusion_amd import engine as E
    from progressive_stable_diffusion_amd import lib as L
    import torch.nn.functional as Fn
    m, k0, c = 4096 + 64, 320, 640                 # ragged last row tile for the 128-row kernels
    n = 1024 if (geglu or ctile[1] != 160) else 960
    g = torch.Generator().manual_seed(81)
    x0 = torch.randn(1, m, 1, k0, generator=g).to(F16)
    w0 = (torch.randn(c, k0, generator=g) / math.sqrt(k0)).to(F16)
    res = (torch.randn(1, m, 1, c, generator=g) + 2.0 * torch.randn(1, m, 1, 1, generator=g)).to(F16)
    b0 = torch.randn(c, generator=g) * 0.1
    parts = c // (ptile[1] // 2)
    st = hip.zeros((parts, m, 2), F32)
    h = hip.zeros((1, m, 1, c), F16)
    pf = L.EPI_BIAS | L.EPI_RESIDUAL | L.EPI_LNSTAT | ptune
    hip.igemm(dev(hip, x0), dev(hip, w0), h, bias=dev(hip, b0), residual=dev(hip, res), flags=pf, tile_m=ptile[0],
              tile_n=ptile[1], ln_stats_out=st)
    hip.synchronize()
    hc = h.float().cpu().reshape(m, parts, -1)
    want = torch.stack([hc.sum(-1), (hc * hc).sum(-1)], dim=-1).permute(1, 0, 2)
    err = (st.cpu() - want).abs().max().item()
    assert err <= 1e-5 * want.abs().max().item() + 1e-4, (err, ptile, ptune)
    # consumer
    w = torch.randn(n, c, generator=g) / math.sqrt(c)
    b = torch.randn(n, generator=g) * 0.1
    gamma, beta = 1.0 + 0.2 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    ref = Fn.linear(Fn.layer_norm(h.float().cpu(), (c,), gamma, beta, 1e-5), w.to(F16).float(), b)
    if geglu:
        hid, gate = ref.chunk(2, dim=-1)
        ref = hid * Fn.gelu(gate)
    w16, c1, bias = E.fold_layernorm(w, b, gamma, beta)
    if geglu:
        idx = E.geglu_interleave(torch.arange(n)[:, None].float(), torch.zeros(n))[0][:, 0].long()
        w16, c1, bias = w16[idx], c1[idx], bias[idx]
    cf = L.EPI_BIAS | L.EPI_LNFOLD | (L.EPI_GEGLU if geglu else 0) | ctune
    wd, bd, cd = dev(hip, w16.contiguous()), dev(hip, bias.contiguous()), dev(hip, c1.contiguous())
    o = hip.zeros((1, m, 1, n // 2 if geglu else n), F16)
    hip.igemm(h, wd, o, bias=bd, flags=cf, tile_m=ctile[0], tile_n=ctile[1], ln_c1=cd, ln_stats_in=st)
    hip.synchronize()
    close(o, ref, 6e-3, 6e-3, f"ln stats {ptile}/{ptune} -> {ctile}/{ctune} geglu{geglu}")
    o2 = hip.zeros(tuple(o.shape), F16)
    for _ in range(2):
        hip.igemm(h, wd, o2, bias=bd, flags=cf, tile_m=ctile[0], tile_n=ctile[1], ln_c1=cd, ln_stats_in=st)
        hip.synchronize()
        assert torch.equal(o2, o)
    with pytest.raises(ValueError):           # the part count must be N / (tile_n / 2) of the launch that writes them
        hip.igemm(dev(hip, x0), dev(hip, w0), h, flags=L.EPI_LNSTAT | ptune, tile_m=ptile[0], tile_n=ptile[1],
                  ln_stats_out=hip.zeros((parts + 1, m, 2), F32))
    if ptile[1] == 128:
        with pytest.raises(ValueError):       # no row partials of a GEGLU output
            hip.igemm(dev(hip, x0), dev(hip, rnd((1024, k0), 5)), hip.zeros((1, m, 1, 512), F16),
                      flags=L.EPI_LNSTAT | L.EPI_GEGLU | ptune, tile_m=ptile[0], tile_n=128,
                      ln_stats_out=hip.zeros((16, m, 2), F32))


@pytest.mark.parametrize("taps", [1, 9])
def test_igemm_skip_concat(hip, taps):
    b, h, c1, c2, n = 2, 8, 640, 320, 640
    x, x2 = rnd((b, h, h, c1), 9), rnd((b, h, h, c2), 10)
    w = rnd((n, taps * (c1 + c2)), 11, 1 / math.sqrt(taps * (c1 + c2)))
    bias = rnd((n,), 12, 0.1, F32)
    o, o_ref = run_igemm(hip, x, w, (b, h, h, n), x2=x2, bias=bias, taps=taps, pad=taps // 9, flags=1)
    close(o, o_ref, 3e-3, 2e-3, f"concat taps{taps}")


@pytest.mark.parametrize("in_launch", [True, False])
@pytest.mark.parametrize("splitk", [2, 7, 16])
def test_igemm_splitk_matches_single_pass(hip, splitk, in_launch):
    """K slices combined inside the launch by the last-arriving slice (agent-scope ticket) or by the
    separate finish kernel; three back-to-back launches reuse the tickets; both orders of summation are
    fixed, so the two modes must agree bit for bit."""
    b, h, c, n = 2, 16, 1280, 1280
    x, w = rnd((b, h, h, c), 13), rnd((n, 9 * c), 14, 1 / math.sqrt(9 * c))
    bias, res = rnd((n,), 15, 0.1, F32), rnd((b, h, h, n), 16)
    o, o_ref = run_igemm(hip, x, w, (b, h, h, n), bias=bias, residual=res, taps=9, pad=1, flags=5,
                         splitk=splitk, in_launch_combine=in_launch, repeats=3)
    close(o, o_ref, 3e-3, 2e-3, f"splitk {splitk} in_launch={in_launch}")
    o2, _ = run_igemm(hip, x, w, (b, h, h, n), bias=bias, residual=res, taps=9, pad=1, flags=5,
                      splitk=splitk, in_launch_combine=not in_launch)
    assert torch.equal(o.cpu(), o2.cpu())


@pytest.mark.parametrize("tile_m,tune,splitk", [(128, 0, 1), (128, 0, 3), (128, 32, 1), (128, 32 | 16, 1),
                                                (64, 0, 1), (64, 16, 1), (64, 0, 5), (128, 32, 4)])
@pytest.mark.parametrize("shape", ["conv", "cat1x1", "ups", "ragged"])
def test_igemm_kernel_variants(hip, tile_m, tune, splitk, shape):
    """Every kernel variant the tiling policy can pick (LDS-DMA ring, register-staged with one or two
    K tiles in flight, 64/128-row tiles, split-K) on the same problems; tune: 16=shallow, 32=no DMA."""
    b = 2
    if shape == "conv":
        h, c1, c2, n, taps, ups, ho = 12, 320, 0, 320, 9, 0, 12
    elif shape == "cat1x1":
        h, c1, c2, n, taps, ups, ho = 16, 640, 320, 640, 1, 0, 16
    elif shape == "ups":
        h, c1, c2, n, taps, ups, ho = 8, 640, 0, 640, 9, 1, 16
    else:                                   # M = 2*7*7 = 98 rows: a single, mostly empty row tile
        h, c1, c2, n, taps, ups, ho = 7, 1280, 1280, 1280, 9, 0, 7
    x = rnd((b, h, h, c1), 60)
    x2 = rnd((b, h, h, c2), 61) if c2 else None
    k = taps * (c1 + c2)
    w = rnd((n, k), 62, 1 / math.sqrt(k))
    bias, res = rnd((n,), 63, 0.1, F32), rnd((b, ho, ho, n), 64)
    o, o_ref = run_igemm(hip, x, w, (b, ho, ho, n), x2=x2, bias=bias, residual=res, taps=taps, ups=ups,
                         pad=taps // 9, flags=5 | tune, splitk=splitk, tile_m=tile_m)
    close(o, o_ref, 3e-3, 2e-3, f"{shape} tm{tile_m} tune{tune} sk{splitk}")


@pytest.mark.parametrize("case", ["qkv", "k1", "k2_ragged", "conv", "geglu", "geglu_2d", "uneven"])
def test_igemm_persistent_ring(hip, case):
    """More 128-row output tiles than CUs and no split-K: on request (DADD_TUNE_PERSIST) the LDS-DMA kernel
    runs its ring as one stream over a contiguous range of output tiles per workgroup.  Cases: K of
    5 / 1 / 2 tiles (the DMA cursor is up to three OUTPUT tiles ahead of the MFMAs), ragged M and N,
    a 3x3 gather, the GEGLU epilogue (also at the 32x32 level's shape, whose 5.2 MB of A under 6.5 MB of W select the 2-D
    XCD tile groups of tile_decode), and a tile count that does not divide by the workgroup count."""
    from progressive_stable_diffusion_amd.engine import geglu_interleave
    kw, flags, tile_n = {}, 1, 0
    if case == "qkv":
        b, h, c, n = 4, 64, 320, 960                    # 128 x 6 = 768 tiles
    elif case == "k1":
        b, h, c, n = 4, 48, 64, 640                     # 72 x 4 = 288 tiles, one K tile each
    elif case == "k2_ragged":
        b, h, c, n = 3, 61, 128, 800                    # M = 11163 (ragged), N = 800 = 5 x 160, 88 x 5 tiles
    elif case == "conv":
        b, h, c, n = 4, 64, 64, 640                     # 128 x 4 tiles, K = 576
        kw = dict(taps=9, pad=1)
    elif case == "geglu":
        b, h, c, n = 4, 64, 320, 2560                   # 128 x 20 tiles of 128 x 128
        flags, tile_n = 1 | 8, 128
    elif case == "geglu_2d":
        b, h, c, n = 4, 32, 640, 5120                   # 32 x 40 tiles: groups of 16 x 10 per XCD
        flags, tile_n = 1 | 8, 128
    else:
        b, h, c, n = 1, 200, 192, 480                   # 313 x 3 = 939 tiles over 256 workgroups
    k = c * kw.get("taps", 1)
    x = rnd((b, h, h, c), 70)
    if case.startswith("geglu"):
        w32, b32 = rnd((n, k), 71, 1 / math.sqrt(k), F32), rnd((n,), 72, 0.1, F32)
        w32, bias = geglu_interleave(w32, b32)
        w = w32.to(F16)
    else:
        w, bias = rnd((n, k), 71, 1 / math.sqrt(k)), rnd((n,), 72, 0.1, F32)
    n_out = n // 2 if case.startswith("geglu") else n
    res = None
    if case in ("qkv", "conv", "uneven"):
        res, flags = rnd((b, h, h, n), 73), flags | 4
    o, o_ref = run_igemm(hip, x, w, (b, h, h, n_out), bias=bias, residual=res, flags=flags | 64, tile_n=tile_n,
                         tile_m=128, **kw)       # 64 = DADD_TUNE_PERSIST
    close(o, o_ref, 3e-3, 3e-3, f"persistent {case}")
    o3, _ = run_igemm(hip, x, w, (b, h, h, n_out), bias=bias, residual=res, flags=flags, tile_n=tile_n,
                      tile_m=128, **kw)          # default: wave-specialised ring, one tile per workgroup
    assert torch.equal(o.cpu(), o3.cpu()), "persistent and wave-specialised rings must agree bit for bit"
    o2, _ = run_igemm(hip, x, w, (b, h, h, n_out), bias=bias, residual=res, flags=flags | 32, tile_n=tile_n,
                      tile_m=128, **kw)        # register-staged kernel: same K order, same epilogue
    if case == "conv":     # 3x3: the LDS-DMA kernel sums taps-fastest, the register-staged one channels-fastest
        close(o, o2, 2e-3, 2e-3, "persistent vs register-staged (different K order)")
    else:
        assert torch.equal(o.cpu(), o2.cpu()), "persistent ring and register-staged kernel must agree bit for bit"


@pytest.mark.parametrize("b,h,c1,c2,n,splitk", [(2, 64, 128, 0, 320, 1), (1, 32, 128, 64, 480, 1), (2, 16, 256, 0, 160, 1),
                                                (1, 64, 64, 0, 160, 1), (2, 32, 320, 320, 320, 2), (3, 16, 640, 0, 320, 5)])
def test_conv3x3_halo_kernel(hip, b, h, c1, c2, n, splitk):
    """3x3 / stride 1 / pad 1 on 64-, 32- and 16-wide maps goes to conv3x3_halo_kernel (halo of the 128-pixel
    tile resident in LDS, nine taps by shifted fragment reads): image borders (zero fill), one and several
    channel chunks, skip-concat (chunks from two tensors), several column tiles, split-K over chunks; against the
    torch reference and against the register-staged implicit GEMM (other K order -> tolerance, not bits)."""
    x = rnd((b, h, h, c1), 80)
    x2 = rnd((b, h, h, c2), 81) if c2 else None
    k = 9 * (c1 + c2)
    w = rnd((n, k), 82, 1 / math.sqrt(k))
    bias, rowvec, res = rnd((n,), 83, 0.1, F32), rnd((b, n), 84, 0.3, F32), rnd((b, h, h, n), 85)
    kw = dict(x2=x2, bias=bias, rowvec=rowvec, residual=res, taps=9, pad=1, flags=7, tile_m=128)
    o, o_ref = run_igemm(hip, x, w, (b, h, h, n), splitk=splitk, in_launch_combine=False, **kw)
    close(o, o_ref, 3e-3, 2e-3, f"halo conv {b}x{h}x{h} c{c1}+{c2}->{n} sk{splitk}")
    kw["flags"] = 7 | 32
    o2, _ = run_igemm(hip, x, w, (b, h, h, n), **kw)
    close(o, o2, 2e-3, 2e-3, "halo conv vs register-staged implicit GEMM")
    # the two-MFMA-waves-per-SIMD build (tune 16: K halves summed apart -> tolerance) and its run to run bit identity
    kw["flags"] = 7 | 16
    o1, _ = run_igemm(hip, x, w, (b, h, h, n), splitk=splitk, in_launch_combine=False, **kw)
    close(o1, o_ref, 3e-3, 2e-3, "halo conv, two MFMA waves per SIMD")
    close(o, o1, 2e-3, 2e-3, "halo conv: two MFMA waves per SIMD vs one")
    o3, _ = run_igemm(hip, x, w, (b, h, h, n), splitk=splitk, in_launch_combine=False, **kw)
    assert torch.equal(o3.cpu(), o1.cpu())


@pytest.mark.parametrize("b,h,c,n,silu", [(2, 64, 320, 320, 1), (2, 32, 640, 320, 1), (3, 16, 1024, 160, 0), (1, 64, 64, 160, 1),
                                          (2, 16, 1280, 320, 1), (1, 32, 2048, 160, 1), (1, 64, 1152, 160, 0)])
def test_conv3x3_halo_groupnorm_on_the_way_in(hip, b, h, c, n, silu):
    """DADD_PRE_GN: the halo conv normalises (+ SiLU) its input in LDS from the producer's chunk partials.  Against
    groupnorm (from the same partials) -> conv as two launches: same rounding points, so a tight tolerance; image
    borders (the padding must be zeros of the NORMALISED tensor), 1 / 5 / 10 / 16 channel chunks, all three widths,
    bias + time row + residual epilogue with the GroupNorm statistics of the OUTPUT on top; contract errors."""
    from progressive_stable_diffusion_amd import lib as L
    x = (rnd((b, h, h, c), 86).float() * 1.7 + 0.6).to(F16)
    k = 9 * c
    w = rnd((n, k), 87, 1 / math.sqrt(k))
    bias, rowvec, res = rnd((n,), 88, 0.1, F32), rnd((b, n), 89, 0.3, F32), rnd((b, h, h, n), 90)
    gamma, beta = rnd((c,), 91, 0.2, F32) + 1.0, rnd((c,), 92, 0.2, F32)
    nch = max(1, min(128, h * h // 64))
    xc = x.float().reshape(b, nch, -1, 32, c // 32)
    part = torch.stack([xc.sum(dim=(2, 4)), (xc * xc).sum(dim=(2, 4))], dim=-1).contiguous()     # [b][nch][32][2]
    ws = dev(hip, part.reshape(-1))
    xd, wd = dev(hip, x), dev(hip, w)
    # two launches
    xn = hip.zeros((b, h, h, c), F16)
    hip.groupnorm(xd, None, dev(hip, gamma), dev(hip, beta), xn, ws, 32, 1e-5, silu, ws_chunks=nch)
    o_ref = hip.zeros((b, h, h, n), F16)
    hip.igemm(xn, wd, o_ref, bias=dev(hip, bias), rowvec=dev(hip, rowvec), residual=dev(hip, res), taps=9, pad=1,
              flags=7, tile_m=128, tile_n=160)
    # one launch, with the statistics of the output as well
    nch_o = h * h // 64
    ws_o = hip.zeros((b * nch_o * 64,), F32)
    o = hip.zeros((b, h, h, n), F16)
    fl = 7 | L.PRE_GN | (L.PRE_GN_SILU if silu else 0) | (L.EPI_GNSTAT if n == 320 else 0)
    kw = dict(gn_ws=ws_o, gn_nchunk=nch_o) if n == 320 else {}
    hip.igemm(xd, wd, o, bias=dev(hip, bias), rowvec=dev(hip, rowvec), residual=dev(hip, res), taps=9, pad=1, flags=fl,
              tile_m=128, tile_n=160, gn_in=(ws, nch, dev(hip, gamma), dev(hip, beta), 1e-5), **kw)
    hip.synchronize()
    close(o, o_ref.float().cpu(), 2e-3, 1e-3, f"conv with GroupNorm on the way in {b}x{h}x{h}x{c}->{n}")
    o2 = hip.zeros((b, h, h, n), F16)
    hip.igemm(xd, wd, o2, bias=dev(hip, bias), rowvec=dev(hip, rowvec), residual=dev(hip, res), taps=9, pad=1, flags=fl,
              tile_m=128, tile_n=160, gn_in=(ws, nch, dev(hip, gamma), dev(hip, beta), 1e-5), **kw)
    hip.synchronize()
    assert torch.equal(o2.cpu(), o.cpu())
    with pytest.raises(ValueError):           # not a halo conv: a 1x1
        hip.igemm(xd, dev(hip, rnd((n, c), 5)), o, flags=L.PRE_GN, gn_in=(ws, nch, dev(hip, gamma), dev(hip, beta), 1e-5))
    if 2 * c <= 1152:
        with pytest.raises(ValueError):       # two sources need the partials of both
            g2, b2 = dev(hip, torch.cat([gamma, gamma])), dev(hip, torch.cat([beta, beta]))
            hip.igemm(xd, dev(hip, rnd((n, 18 * c), 6)), o, x2=xd, taps=9, pad=1, flags=L.PRE_GN, tile_m=128, tile_n=160,
                      gn_in=(ws, nch, g2, b2, 1e-5))


@pytest.mark.parametrize("b,h,c1,c2,n", [(2, 64, 320, 320, 320), (1, 32, 640, 640, 160), (2, 16, 1280, 640, 160)])
def test_conv3x3_halo_groupnorm_over_skip_concat(hip, b, h, c1, c2, n):
    """DADD_PRE_GN over [x | x2]: each source brings the chunk partials of ITS OWN 32 groups (different chunk counts); the
    kernel unites them into the groups of the concatenation (widths nest: 20 = 2 x 10, 40 = 2 x 20; 1280 + 640 gives
    60 = 1.5 x 40 and must be refused).  Against groupnorm over both sources (its own statistics pass) -> conv."""
    from progressive_stable_diffusion_amd import lib as L
    x = (rnd((b, h, h, c1), 86).float() * 1.7 + 0.6).to(F16)
    x2 = (rnd((b, h, h, c2), 93).float() * 0.8 - 0.3).to(F16)
    k = 9 * (c1 + c2)
    w = rnd((n, k), 87, 1 / math.sqrt(k))
    bias = rnd((n,), 88, 0.1, F32)
    gamma, beta = rnd((c1 + c2,), 91, 0.2, F32) + 1.0, rnd((c1 + c2,), 92, 0.2, F32)

    def partials(t, nch):
        tc = t.float().reshape(b, nch, -1, 32, t.shape[-1] // 32)
        return torch.stack([tc.sum(dim=(2, 4)), (tc * tc).sum(dim=(2, 4))], dim=-1).contiguous().reshape(-1)
    nch1, nch2 = max(1, h * h // 64), max(1, h * h // 128)
    ws1, ws2 = dev(hip, partials(x, nch1)), dev(hip, partials(x2, nch2))
    xd, x2d, wd = dev(hip, x), dev(hip, x2), dev(hip, w)
    o = hip.zeros((b, h, h, n), F16)
    args = dict(x2=x2d, bias=dev(hip, bias), taps=9, pad=1, flags=1 | L.PRE_GN | L.PRE_GN_SILU, tile_m=128, tile_n=160,
                gn_in=(ws1, nch1, dev(hip, gamma), dev(hip, beta), 1e-5, ws2, nch2))
    if (c1 + c2) // 32 % (c1 // 32) or (c1 + c2) // 32 % (c2 // 32) or c1 % ((c1 + c2) // 32):
        with pytest.raises(ValueError):
            hip.igemm(xd, wd, o, **args)
        return
    hip.igemm(xd, wd, o, **args)
    xn = hip.zeros((b, h, h, c1 + c2), F16)
    hip.groupnorm(xd, x2d, dev(hip, gamma), dev(hip, beta), xn, hip.zeros((b * L.GN_MAX_CHUNKS * 64,), F32), 32, 1e-5, 1)
    o_ref = hip.zeros((b, h, h, n), F16)
    hip.igemm(xn, wd, o_ref, bias=dev(hip, bias), taps=9, pad=1, flags=1, tile_m=128, tile_n=160)
    hip.synchronize()
    close(o, o_ref.float().cpu(), 3e-3, 2e-3, f"conv with GroupNorm over [x | x2] {b}x{h}x{h}x({c1}+{c2})->{n}")


def test_igemm_geglu(hip):
    from progressive_stable_diffusion_amd.engine import geglu_interleave
    m, c = 300, 320
    x = rnd((1, m, 1, c), 17)
    w, bias = rnd((8 * c, c), 18, 1 / math.sqrt(c), F32), rnd((8 * c,), 19, 0.1, F32)
    wp, bp = geglu_interleave(w, bias)
    o, o_ref = run_igemm(hip, x, wp.to(F16), (1, m, 1, 4 * c), bias=bp, flags=1 | 8, tile_n=128)
    close(o, o_ref, 3e-3, 3e-3, "geglu")
    # and against the un-interleaved definition: hidden * gelu(gate)
    y = torch.nn.functional.linear(x.float().reshape(m, c), w.to(F16).float(), bias)
    hid, gate = y.chunk(2, dim=-1)
    close(o.reshape(m, 4 * c), hid * torch.nn.functional.gelu(gate), 3e-3, 3e-3, "geglu vs definition")


def test_igemm_rejects_bad_contract(hip):
    x, w = rnd((1, 8, 1, 96), 20), rnd((64, 96), 21)
    with pytest.raises(ValueError):
        hip.igemm(dev(hip, x), dev(hip, w), hip.zeros((1, 8, 1, 64), F16))      # C % 64 != 0


@pytest.mark.parametrize("c1,c2,hw,silu,eps", [(320, 0, 256, 1, 1e-5), (1280, 640, 64, 1, 1e-5),
                                               (640, 320, 100, 0, 1e-6), (128, 0, 4096, 1, 1e-6),
                                               (1280, 1280, 16, 1, 1e-5),
                                               (320, 320, 4096, 1, 1e-5),      # slab > LDS: two-pass path
                                               (640, 320, 4096, 0, 1e-6), (128, 0, 16384, 1, 1e-6)])
def test_groupnorm(hip, c1, c2, hw, silu, eps):
    b, side = 2, int(math.isqrt(hw))
    x1 = rnd((b, side, side, c1), 22, 1.5) + 0.3
    x2 = (rnd((b, side, side, c2), 23, 0.7) - 0.2) if c2 else None
    c = c1 + c2
    gamma, beta = rnd((c,), 24, 0.2, F32) + 1.0, rnd((c,), 25, 0.2, F32)
    o_ref = torch.zeros(b, side, side, c, dtype=F16)
    REF.groupnorm(x1, x2, gamma, beta, o_ref, None, 32, eps, silu)
    o = hip.zeros((b, side, side, c), F16)
    ws = hip.zeros((b * 256 * 32 * 2,), F32)
    hip.groupnorm(dev(hip, x1), dev(hip, x2), dev(hip, gamma), dev(hip, beta), o, ws, 32, eps, silu)
    hip.synchronize()
    close(o, o_ref, 3e-3, 2e-3, f"groupnorm c{c1}+{c2} hw{hw}")


@pytest.mark.parametrize("m,c", [(100, 320), (37, 640), (64, 1280)])
def test_layernorm(hip, m, c):
    x = rnd((m, c), 26, 2.0) + 0.5
    gamma, beta = rnd((c,), 27, 0.2, F32) + 1.0, rnd((c,), 28, 0.2, F32)
    o_ref = torch.zeros(m, c, dtype=F16)
    REF.layernorm(x, gamma, beta, o_ref)
    o = hip.zeros((m, c), F16)
    hip.layernorm(dev(hip, x), dev(hip, gamma), dev(hip, beta), o)
    hip.synchronize()
    close(o, o_ref, 2e-3, 2e-3, f"layernorm {m}x{c}")


@pytest.mark.parametrize("heads,d,n", [(8, 40, 256), (8, 40, 144), (8, 80, 64), (8, 160, 100),
                                       (8, 160, 16), (1, 512, 192), (8, 40, 1024)])
def test_self_attention(hip, heads, d, n):
    b, c = 2, heads * d
    qkv = rnd((b, n, 3 * c), 29, 1.0)
    qkv[0, n // 3, c:2 * c] *= 4.0          # a spiky key row: exercises the running-max rescale
    o_ref = torch.zeros(b, n, c, dtype=F16)
    REF.self_attn(qkv, o_ref, heads)
    o = hip.zeros((b, n, c), F16)
    hip.self_attn(dev(hip, qkv), o, heads)
    hip.synchronize()
    close(o, o_ref, 3e-3, 3e-3, f"self_attn h{heads} d{d} n{n}")


@pytest.mark.parametrize("d", [40, 80, 160])
@pytest.mark.parametrize("mode,lam", [(0, 0.0), (0, 3.0), (0, -0.5), (1, 0.0)])
def test_tri_xattn(hip, d, mode, lam):
    b, n, heads = 2, 300, 8
    c = heads * d
    t_tok, ld = (48, 4 * c) if mode == 0 else (32, 2 * c)
    q, kv = rnd((b, n, c), 30, 1.0), rnd((b, t_tok, ld), 31, 1.0)
    gates = torch.tensor([0.1, 0.9])
    o_ref = torch.zeros(b, n, c, dtype=F16)
    REF.tri_xattn(q, kv, o_ref, gates, lam, mode, heads)
    o = hip.zeros((b, n, c), F16)
    hip.tri_xattn(dev(hip, q), dev(hip, kv), o, dev(hip, gates) if mode == 0 else None, lam, mode, heads)
    hip.synchronize()
    close(o, o_ref, 4e-3, 4e-3, f"tri_xattn d{d} mode{mode} lam{lam}")


# the last case has B*HW/128 = 256 tiles >= the CU count: the dispatcher then picks the 128-token instantiation
# (attn2_fused_kernel<128>, the one the reference CLI default B = mes_steps = 13 selects); the others run <64>
@pytest.mark.parametrize("b,hw,c", [(2, 256, 320), (1, 1024, 640), (2, 128, 1280), (8, 4096, 320)])
def test_attn2_fused(hip, b, hw, c):
    """x (W_q K^T) -> 24 independent 16-wide softmaxes -> P (V W_o^T) + bias + residual in one launch, against the
    same arithmetic in torch (P rounded to fp16 in both)."""
    x, res = rnd((b, hw, c), 90), rnd((b, hw, c), 91)
    mcat = rnd((b, 384, c), 92, 2.0 / math.sqrt(c))          # scores of a few units (log2 domain)
    vw = rnd((b, c, 384), 93, 0.5)
    bias = rnd((c,), 94, 0.1, F32)
    o_ref = torch.zeros(b, hw, c, dtype=F16)
    REF.attn2_fused(x, mcat, vw, bias, res, o_ref)
    o = hip.zeros((b, hw, c), F16)
    hip.attn2_fused(dev(hip, x), dev(hip, mcat), dev(hip, vw), dev(hip, bias), dev(hip, res), o)
    hip.synchronize()
    close(o, o_ref, 4e-3, 3e-3, f"attn2_fused {b}x{hw}x{c}")
    # the LayerNorm row partials of the output (for the GEGLU projection behind norm3): same result, sums of what is stored
    st = hip.zeros((c // 80, b * hw, 2), F32)
    o_st = hip.zeros((b, hw, c), F16)
    hip.attn2_fused(dev(hip, x), dev(hip, mcat), dev(hip, vw), dev(hip, bias), dev(hip, res), o_st, ln_stats_out=st)
    hip.synchronize()
    assert torch.equal(o_st, o)
    oc = o.float().cpu().reshape(b * hw, c // 80, 80)
    want = torch.stack([oc.sum(-1), (oc * oc).sum(-1)], dim=-1).permute(1, 0, 2)
    assert (st.cpu() - want).abs().max().item() <= 1e-5 * want.abs().max().item() + 1e-4
    # norm2 folded into the score GEMM: un-normalised x with a common offset per row, its row partials (4 parts, as the
    # producing GEMM writes them), mcat carrying gamma, c1 / d — against LayerNorm -> the unfolded kernel arithmetic
    import torch.nn.functional as Fn
    gen = torch.Generator().manual_seed(95)
    xr = (x.float() * 1.5 + 2.0 * torch.randn(b, hw, 1, generator=gen)).to(F16)
    gamma, beta = 1.0 + 0.2 * torch.randn(c, generator=gen), 0.2 * torch.randn(c, generator=gen)
    m0 = mcat.float()
    mg = (m0 * gamma).to(F16)
    c1, dvec = mg.float().sum(-1), (m0 * beta).sum(-1)
    xp = xr.float().reshape(b * hw, 4, c // 4)
    st_in = torch.stack([xp.sum(-1), (xp * xp).sum(-1)], dim=-1).permute(1, 0, 2).contiguous()
    o_ref2 = torch.zeros(b, hw, c, dtype=F16)
    REF.attn2_fused(Fn.layer_norm(xr.float(), (c,), gamma, beta, 1e-5).to(F16), mcat, vw, bias, res, o_ref2)
    o_f = hip.zeros((b, hw, c), F16)
    hip.attn2_fused(dev(hip, xr), dev(hip, mg), dev(hip, vw), dev(hip, bias), dev(hip, res), o_f,
                    ln_stats_in=dev(hip, st_in), ln_c1=dev(hip, c1.contiguous()), ln_d=dev(hip, dvec.contiguous()))
    hip.synchronize()
    close(o_f, o_ref2, 8e-3, 6e-3, f"attn2_fused with norm2 folded {b}x{hw}x{c}")
    with pytest.raises(ValueError):
        hip.attn2_fused(dev(hip, x[:, :64]), dev(hip, mcat), dev(hip, vw), None, dev(hip, res[:, :64]),
                        hip.zeros((b, 64, c), F16))          # fewer than 128 tokens per sample


@pytest.mark.parametrize("b,hw", [(1, 64), (2, 256), (4, 4096)])
def test_ffn_block(hip, b, hw):
    """Transformer-block tail in one launch (csrc/ffn_block.hip): LayerNorm 3 -> GEGLU projection -> FF-out + residual ->
    proj_out + outer residual, against the same chain in torch with the unfused launches' rounding points (normalised
    rows, GEGLU output, h4, result in fp16; fp32 everywhere else) — through the packed weight stream the device reads.
    Tolerance: the result passes two K = 320 / 1280 fp16-operand GEMMs behind rounded intermediates: 6e-3 + 4e-3 |ref|.
    GroupNorm partials: sums of the STORED values (1e-5 relative); launches are bit-reproducible (the ring protocol)."""
    from progressive_stable_diffusion_amd.engine import pack_ffn_stream
    c, hid = 320, 1280
    x = (rnd((b, hw, c), 700, 1.0).float() + 0.5 * torch.randn(b, hw, 1, generator=torch.Generator().manual_seed(701))).to(F16)
    xres = rnd((b, hw, c), 702)
    w1, w2 = rnd((2 * hid, c), 703, 1.0 / math.sqrt(c)), rnd((c, hid), 704, 1.0 / math.sqrt(hid))
    wp = rnd((c, c, 1, 1), 705, 1.0 / math.sqrt(c))
    b1, b2, bp = rnd((2 * hid,), 706, 0.2, F32), rnd((c,), 707, 0.2, F32), rnd((c,), 708, 0.2, F32)
    g = torch.Generator().manual_seed(709)
    gam, bet = 1.0 + 0.2 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    stream, b1p = pack_ffn_stream(w1, b1, w2, wp)
    nchunk = hw // 32
    o_ref, ws_ref = torch.zeros(b, hw, c, dtype=F16), torch.zeros(b * nchunk * 64)
    REF.ffn_block(x, stream, gam, bet, b1p, b2, bp, xres, o_ref, gn_ws=ws_ref, gn_nchunk=nchunk)
    args = [dev(hip, t) for t in (x, stream, gam, bet, b1p, b2, bp, xres)]
    o, ws = hip.zeros((b, hw, c), F16), hip.zeros((b * nchunk * 64,), F32)
    hip.ffn_block(*args, o, gn_ws=ws, gn_nchunk=nchunk)
    hip.synchronize()
    close(o, o_ref, 6e-3, 4e-3, f"ffn_block {b}x{hw}")
    oc = o.float().cpu().reshape(b, nchunk, 32, 32, c // 32)
    want = torch.stack([oc.sum(dim=(2, 4)), (oc * oc).sum(dim=(2, 4))], dim=-1).reshape(-1)
    assert (ws.cpu() - want).abs().max().item() <= 1e-5 * want.abs().max().item() + 1e-4
    o2 = hip.zeros((b, hw, c), F16)
    for _ in range(3):
        hip.ffn_block(*args, o2)                   # no partials requested
    hip.synchronize()
    assert torch.equal(o2, o)
    with pytest.raises(ValueError):
        hip.ffn_block(dev(hip, x[:, :32].contiguous()), *args[1:7], dev(hip, xres[:, :32].contiguous()),
                      hip.zeros((b, 32, c), F16))      # fewer than 64 tokens per sample


@pytest.mark.parametrize("b,hw,nchunk", [(1, 64, 1), (2, 256, 4), (4, 4096, 64), (4, 4096, 128)])
def test_tf_head(hip, b, hw, nchunk):
    """Transformer-block head in one launch (csrc/tf_head.hip): GroupNorm from chunk partials -> proj_in -> LayerNorm 1 ->
    q|k|v, against the same chain in torch with the unfused launches' rounding points.  hs passes one K = 320 GEMM behind
    a rounded input (3e-3 + 3e-3 |ref|); q|k|v a second one behind LayerNorm of the rounded hs (6e-3 + 4e-3 |ref|)."""
    from progressive_stable_diffusion_amd.engine import pack_head_stream
    c = 320
    x = (rnd((b, hw, c), 720).float() * (1.0 + 0.5 * torch.randn(1, 1, c, generator=torch.Generator().manual_seed(721)))
         + torch.randn(b, 1, c, generator=torch.Generator().manual_seed(722))).to(F16)
    wp = rnd((c, c, 1, 1), 723, 1.0 / math.sqrt(c))
    wq, wk, wv = (rnd((c, c), 724 + i, 1.0 / math.sqrt(c)) for i in range(3))
    bp = rnd((c,), 727, 0.2, F32)
    g = torch.Generator().manual_seed(728)
    gg, gb = 1.0 + 0.2 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    lg, lb = 1.0 + 0.2 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    xs = x.float().reshape(b, nchunk, hw // nchunk, 32, c // 32)
    ws = torch.stack([xs.sum(dim=(2, 4)), (xs * xs).sum(dim=(2, 4))], dim=-1).reshape(-1).contiguous()     # [b][chunk][32][2]
    stream = pack_head_stream(wp, wq, wk, wv)
    hs_ref, qkv_ref = torch.zeros(b, hw, c, dtype=F16), torch.zeros(b, hw, 3 * c, dtype=F16)
    REF.tf_head(x, stream, ws, nchunk, gg, gb, bp, lg, lb, hs_ref, qkv_ref)
    args = [dev(hip, t) for t in (x, stream, ws)] + [nchunk] + [dev(hip, t) for t in (gg, gb, bp, lg, lb)]
    hs, qkv = hip.zeros((b, hw, c), F16), hip.zeros((b, hw, 3 * c), F16)
    hip.tf_head(*args, hs, qkv)
    hip.synchronize()
    close(hs, hs_ref, 3e-3, 3e-3, f"tf_head hs {b}x{hw}")
    close(qkv, qkv_ref, 6e-3, 4e-3, f"tf_head qkv {b}x{hw}")
    hs2, qkv2 = hip.zeros((b, hw, c), F16), hip.zeros((b, hw, 3 * c), F16)
    for _ in range(3):
        hip.tf_head(*args, hs2, qkv2)
    hip.synchronize()
    assert torch.equal(hs2, hs) and torch.equal(qkv2, qkv)


def test_tri_xattn_lambda_zero_equals_two_pathways(hip):
    """routing_gates.py:160,177-178: delta_scale == 0 must skip the delta pathway exactly; garbage
    (even NaN) in the delta tokens must not leak."""
    b, n, heads, d = 1, 64, 8, 40
    c = heads * d
    q, kv = rnd((b, n, c), 32), rnd((b, 48, 4 * c), 33)
    kv2 = kv.clone()
    kv2[:, 32:] = float("nan")
    g = dev(hip, torch.tensor([0.9, 0.1]))
    o1, o2 = hip.zeros((b, n, c), F16), hip.zeros((b, n, c), F16)
    hip.tri_xattn(dev(hip, q), dev(hip, kv), o1, g, 0.0, 0, heads)
    hip.tri_xattn(dev(hip, q), dev(hip, kv2), o2, g, 0.0, 0, heads)
    hip.synchronize()
    assert torch.equal(o1.cpu(), o2.cpu())
    # lambda as a DEVICE-side parameter (one captured graph serves a lambda sweep): the three-pathway kernel is
    # launched and skips the delta pathway itself when the value it reads is 0 — same bits, NaN tokens never read
    lam = dev(hip, torch.tensor([0.0]))
    o3, o4 = hip.zeros((b, n, c), F16), hip.zeros((b, n, c), F16)
    hip.tri_xattn(dev(hip, q), dev(hip, kv2), o3, g, 123.0, 0, heads, lam_dev=lam)       # the by-value lambda is ignored
    hip.copy_(lam, torch.tensor([1.5]))
    hip.tri_xattn(dev(hip, q), dev(hip, kv), o4, g, 0.0, 0, heads, lam_dev=lam)
    o5 = hip.zeros((b, n, c), F16)
    hip.tri_xattn(dev(hip, q), dev(hip, kv), o5, g, 1.5, 0, heads)
    hip.synchronize()
    assert torch.equal(o3.cpu(), o1.cpu()) and torch.equal(o4.cpu(), o5.cpu())
    assert not torch.equal(o4.cpu(), o1.cpu())


def test_thin_convs_and_pack(hip):
    b, s = 2, 16
    lat = rnd((b, 4, s, s), 34, 1.0, F32)
    mat, vec = rnd((4, 4), 35, 0.5, F32), rnd((4,), 36, 0.1, F32)
    for m, v, sc in ((None, None, 1.0), (mat, vec, 1.0 / 0.18215)):
        o_ref = torch.zeros(b, s, s, 8, dtype=F16)
        REF.pack_latents(lat, o_ref, sc, m, v)
        o = hip.zeros((b, s, s, 8), F16)
        hip.pack_latents(dev(hip, lat), o, sc, dev(hip, m), dev(hip, v))
        hip.synchronize()
        close(o, o_ref, 2e-3, 2e-3, "pack")
    x8 = o_ref
    w = rnd((320, 9, 8), 37, 0.2)
    w[:, :, 4:] = 0
    bias = rnd((320,), 38, 0.1, F32)
    o_ref = torch.zeros(b, s, s, 320, dtype=F16)
    REF.conv_cin8(x8, w, bias, o_ref)
    o = hip.zeros((b, s, s, 320), F16)
    hip.conv_cin8(dev(hip, x8), dev(hip, w), dev(hip, bias), o)
    hip.synchronize()
    close(o, o_ref, 3e-3, 2e-3, "conv_cin8")
    # the UNet's conv_in straight from the fp32 NCHW latents == pack (scale 1, no matrix) + conv_cin8, same rounding of
    # the input; ragged pixel count
    for bb, ss in ((2, 16), (3, 21)):
        lat2 = rnd((bb, 4, ss, ss), 39, 1.0, F32)
        o_ref2 = torch.zeros(bb, ss, ss, 320, dtype=F16)
        REF.conv_in_nchw(lat2, w, bias, o_ref2)
        o2 = hip.zeros((bb, ss, ss, 320), F16)
        hip.conv_in_nchw(dev(hip, lat2), dev(hip, w), dev(hip, bias), o2)
        hip.synchronize()
        close(o2, o_ref2, 3e-3, 2e-3, "conv_in_nchw")
    # ... and with the GroupNorm chunk partials of its output (256 pixels per chunk): same pixels bit for bit, the partials
    # equal to the sums over the kernel's own rounded outputs
    for bb, ss in ((2, 16), (3, 32)):
        lat2 = rnd((bb, 4, ss, ss), 45, 1.0, F32)
        nchunk = ss * ss // 256
        o_plain, o_gn = hip.zeros((bb, ss, ss, 320), F16), hip.zeros((bb, ss, ss, 320), F16)
        ws = hip.zeros((bb * nchunk * 64,), F32)
        hip.conv_in_nchw(dev(hip, lat2), dev(hip, w), dev(hip, bias), o_plain)
        hip.conv_in_nchw(dev(hip, lat2), dev(hip, w), dev(hip, bias), o_gn, gn_ws=ws, gn_nchunk=nchunk)
        hip.synchronize()
        assert torch.equal(o_plain.cpu(), o_gn.cpu())
        of = o_gn.cpu().double().reshape(bb, nchunk, 256, 32, 10)
        want = torch.stack([of.sum(dim=(2, 4)), (of * of).sum(dim=(2, 4))], dim=-1)
        got = ws.cpu().double().reshape(bb, nchunk, 32, 2)
        assert torch.allclose(got, want, rtol=2e-5, atol=1e-3), float((got - want).abs().max())
    with pytest.raises(ValueError):      # 21 x 21 pixels: not whole 256-pixel chunks
        hip.conv_in_nchw(dev(hip, rnd((1, 4, 21, 21), 46, 1.0, F32)), dev(hip, w), dev(hip, bias), hip.zeros((1, 21, 21, 320), F16),
                         gn_ws=hip.zeros((64,), F32), gn_nchunk=1)
    for c, co, mode in ((320, 4, 0), (128, 3, 1)):
        x = rnd((b, s, s, c), 39, 1.0)
        w = rnd((co, 9, c), 40, 1 / math.sqrt(9 * c) * (4.0 if mode else 1.0))
        bias = rnd((co,), 41, 0.1, F32)
        o_ref = torch.zeros(b, co, s, s, dtype=F32)
        REF.conv_cout4(x, w, bias, o_ref, mode)
        o = hip.zeros((b, co, s, s), F32)
        hip.conv_cout4(dev(hip, x), dev(hip, w), dev(hip, bias), o, mode)
        hip.synchronize()
        close(o, o_ref, 1e-3, 1e-3, f"conv_cout4 c{c} mode{mode}")


def test_time_rows_and_linear(hip):
    t = torch.tensor([999, 978, 500, 20, 0], dtype=torch.int64)
    f_ref = torch.zeros(5, 320)
    REF.timestep_features(t, f_ref)
    f = hip.zeros((5, 320), F32)
    hip.timestep_features(dev(hip, t), f)
    hip.synchronize()
    close(f, f_ref, 2e-4, 0.0, "timestep features")   # |angle| up to 999 rad in fp32
    for m, k, n, ai, ao in ((5, 320, 1280, 0, 1), (13, 1280, 1280, 0, 0), (50, 1280, 2000, 1, 0)):
        x, w, bias = rnd((m, k), 42, 1.0, F32), rnd((n, k), 43, 1 / math.sqrt(k)), rnd((n,), 44, 0.1, F32)
        o_ref = torch.zeros(m, n)
        REF.linear_rows(x, w, bias, o_ref, ai, ao)
        o = hip.zeros((m, n), F32)
        hip.linear_rows(dev(hip, x), dev(hip, w), dev(hip, bias), o, ai, ao)
        hip.synchronize()
        close(o, o_ref, 2e-4, 2e-4, f"linear_rows {m}x{k}x{n}")


def test_conv_out_fused_with_ddim_update(hip):
    """dadd_conv_out_ddim_f16 == dadd_conv3x3_cout4_f16 (mode 0) followed by dadd_ddim_update_f32, bit for bit (same
    conv arithmetic, the DDIM operations individually rounded in the same order), for a middle step and the last one;
    ragged pixel count (a block carries 32 pixels)."""
    b, h, c, co = 3, 21, 320, 4
    x = rnd((b, h, h, c), 40)
    w, bias = rnd((co, 9, c), 41, 1 / math.sqrt(9 * c)), rnd((co,), 42, 0.1, F32)
    lat = rnd((b, co, h, h), 43, 1.0, F32)
    for coef in (torch.tensor([0.8, 0.6, 0.9, 0.43589]), torch.tensor([0.99, 0.141, -1.0, 0.0])):
        eps, l_ref, l_fused = hip.zeros((b, co, h, h), F32), dev(hip, lat.clone()), dev(hip, lat.clone())
        hip.conv_cout4(dev(hip, x), dev(hip, w), dev(hip, bias), eps, 0)
        hip.ddim_update(l_ref, eps, None, 1.0, dev(hip, coef))
        hip.conv_out_ddim(dev(hip, x), dev(hip, w), dev(hip, bias), l_fused, dev(hip, coef))
        hip.synchronize()
        assert torch.equal(l_fused.cpu(), l_ref.cpu())
        assert not torch.equal(l_fused.cpu(), lat)


def test_ddim_update_is_bit_exact(hip):
    """The DDIM algebra is fp32 on both sides, op for op: the bar is bit-exactness."""
    from oracle.sampler import OracleCfg, ddim_update, noise_schedule
    _, ac, _, _ = noise_schedule(OracleCfg())
    x, e, u = rnd((4, 4, 64, 64), 45, 1.0, F32), rnd((4, 4, 64, 64), 46, 1.0, F32), rnd((4, 4, 64, 64), 47, 1.0, F32)
    for t, tp, last in ((999, 978, False), (500, 489, False), (20, 0, False), (0, None, True)):
        a_t = ac[t]
        coef = torch.stack([torch.sqrt(a_t), torch.sqrt(1 - a_t),
                            torch.tensor(-1.0) if last else torch.sqrt(ac[tp]),
                            torch.tensor(0.0) if last else torch.sqrt(1 - ac[tp])])
        for eu, g in ((None, 1.0), (u, 3.0)):
            eps = e if eu is None else eu + g * (e - eu)
            ref = ddim_update(x, eps, ac, t, tp, last)
            xd = dev(hip, x.clone())
            hip.ddim_update(xd, dev(hip, e), dev(hip, eu), g, dev(hip, coef))
            hip.synchronize()
            assert torch.equal(xd.cpu(), ref), f"ddim t={t} cfg={eu is not None}"


def test_begin_step_and_graph_replay(hip):
    table, coef = rnd((6, 1000), 48, 1.0, F32), rnd((6, 4), 49, 1.0, F32)     # 1000 columns: four blocks, ragged
    td, cd = dev(hip, table), dev(hip, coef)
    cur, cc = hip.zeros((3, 1000), F32), hip.zeros((4,), F32)
    step = hip.zeros((2,), torch.int32)       # (row, block ticket)
    acc = hip.zeros((1, 4, 1, 1), F32)
    ones = dev(hip, torch.tensor([1.0, 0.0, 1.0, 1.0]))   # x <- clamp(x) + eps
    eps = dev(hip, torch.full((1, 4, 1, 1), 0.25))
    hip.graph_begin()
    hip.begin_step(td, cur, cd, cc, step)
    hip.ddim_update(acc, eps, None, 1.0, ones)
    g = hip.graph_end()
    for i in range(5):
        hip.graph_launch(g)
        hip.synchronize()
        assert step.cpu().tolist() == [i + 1, 0]
        assert torch.equal(cur.cpu(), table[i][None].expand(3, -1))
        assert torch.equal(cc.cpu(), coef[i])
    assert torch.allclose(acc.cpu(), torch.full((1, 4, 1, 1), 1.25))
    hip.graph_destroy(g)


def test_weight_prefetch_branch(hip):
    """dadd_prefetch / dadd_prefetch_join: a read-only side branch (eager: a second stream; captured: a parallel branch of
    the graph).  Results of the kernels around it are unchanged, the graph replays, and a join without an open branch is
    a no-op.  (Off in the shipped plans: profiles/r03_zh_weight_prefetch_ab.txt.)"""
    x, w, w2 = rnd((2, 16, 16, 640), 130), rnd((640, 640), 131, 0.04), rnd((1280, 9 * 1280), 132, 0.01)
    xd, wd, w2d = dev(hip, x), dev(hip, w), dev(hip, w2)
    o_ref, o1, o2 = (hip.zeros((2, 16, 16, 640), F16) for _ in range(3))
    hip.igemm(xd, wd, o_ref)
    hip.prefetch_join()                      # nothing open
    hip.prefetch(w2d)                        # 29.5 MB: 48 workgroups
    hip.prefetch(wd[:1, :8])                 # 16 bytes: one load
    hip.igemm(xd, wd, o1)
    hip.prefetch_join()
    hip.synchronize()
    assert torch.equal(o1.cpu(), o_ref.cpu())
    hip.graph_begin()
    hip.prefetch(w2d)
    hip.igemm(xd, wd, o2)
    hip.prefetch_join()
    g = hip.graph_end()
    for _ in range(3):
        hip.zero_(o2)
        hip.graph_launch(g)
        hip.synchronize()
        assert torch.equal(o2.cpu(), o_ref.cpu())
    hip.graph_destroy(g)


def test_profiling_hooks(hip):
    """Every launch between prof_begin / prof_end is recorded with its kernel name, its own begin/end
    timestamps and its algorithmic flop / bytes; capture is refused while profiling."""
    x, w = rnd((1, 256, 1, 320), 50), rnd((320, 320), 51, 0.05)
    xd, wd, o = dev(hip, x), dev(hip, w), hip.zeros((1, 256, 1, 320), F16)
    ln = hip.zeros((1, 256, 1, 320), F16)
    g, b = dev(hip, torch.ones(320)), dev(hip, torch.zeros(320))
    hip.igemm(xd, wd, o)
    hip.synchronize()
    hip.prof_begin()
    for _ in range(3):
        hip.igemm(xd, wd, o)
        hip.layernorm(o, g, b, ln)
    with pytest.raises(ValueError):
        hip.graph_begin()
    rec = hip.prof_end()
    assert len(rec) == 6 and [r[0].split("<")[0] for r in rec[:2]] == [rec[0][0].split("<")[0], "layernorm_kernel"]
    assert "igemm" in rec[0][0]
    for name, us, flop, byt in rec:
        assert 0.5 < us < 500.0, (name, us)
        if "igemm" in name:
            assert flop == pytest.approx(2.0 * 256 * 320 * 320) and byt == pytest.approx(2.0 * (2 * 256 * 320 + 320 * 320))
        else:
            assert flop == 0.0 and byt == pytest.approx(256 * 320 * 4.0)
    hip.igemm(xd, wd, o)          # not recorded any more
    assert hip.prof_end.__self__ is hip


# ------------------------------------------------------------------------------------------------ conditioning front-end
@pytest.mark.parametrize("d,heads,nq,nk", [(64, 16, 257, 257), (64, 1, 257, 257), (96, 8, 16, 257), (96, 8, 16, 16), (40, 8, 100, 300)])
def test_attention_separate_query_and_key_lengths(hip, d, heads, nq, nk):
    """dadd_attn_f16: q rows and k/v rows of different counts, read as strided views of wider rows (the fused q|k|v
    and k|v projections of CLIP / nn.MultiheadAttention); d = 64 (CLIP) and 96 (768 / 8 heads)."""
    b, c = 2, heads * d
    qkv_q = rnd((b, nq, 3 * c), 40)
    kv = rnd((b, nk, 2 * c), 41)
    o_ref = torch.zeros(b, nq, c, dtype=F16)
    REF.attention(qkv_q[:, :, :c], kv[:, :, :c], kv[:, :, c:], o_ref, heads)
    qd, kvd = dev(hip, qkv_q), dev(hip, kv)
    o = hip.zeros((b, nq, c), F16)
    hip.attention(qd[:, :, :c], kvd[:, :, :c], kvd[:, :, c:], o, heads)
    hip.synchronize()
    close(o, o_ref, 3e-3, 3e-3, f"attention d{d} {nq}x{nk}")


@pytest.mark.parametrize("act,flag", [("quick_gelu", 256), ("gelu", 512), ("sigmoid", 1024)])
def test_igemm_activation_epilogues(hip, act, flag):
    import torch.nn.functional as Fn
    m, n, k = 300, 512, 256
    x, w = rnd((1, m, 1, k), 42), rnd((n, k), 43, 1 / math.sqrt(k) * 3)
    bias, res = rnd((n,), 44, 0.3, F32), rnd((1, m, 1, n), 45)
    y = Fn.linear(x.float(), w.float(), bias)
    y = {"quick_gelu": y * torch.sigmoid(1.702 * y), "gelu": Fn.gelu(y), "sigmoid": torch.sigmoid(y)}[act] + res.float()
    o = hip.zeros((1, m, 1, n), F16)
    hip.igemm(dev(hip, x), dev(hip, w), o, bias=dev(hip, bias), residual=dev(hip, res), flags=1 | 4 | flag)
    hip.synchronize()
    close(o, y, 3e-3, 3e-3, act)
    with pytest.raises(ValueError):      # no activation on split-K slabs
        hip.igemm(dev(hip, x), dev(hip, w), o, flags=flag, splitk=2, partial=hip.zeros((2 * m * n,), F32))


def test_conditioning_small_kernels(hip):
    import torch.nn.functional as Fn
    g = torch.Generator().manual_seed(46)
    # CLIP patch rows
    px = torch.randn(2, 3, 28, 42, generator=g)
    rows = hip.zeros((2, 1 + 2 * 3, 640), F16)
    hip.clip_patch_rows(dev(hip, px), rows, 14)
    ref = torch.zeros(2, 7, 640, dtype=F16)
    REF.clip_patch_rows(px, ref, 14)
    hip.synchronize()
    assert torch.equal(rows.cpu(), ref) and float(rows[:, 0].abs().max()) == 0.0
    # AOE interpolation (labels outside [0, 3] clamp, fractional labels lerp) — fp32, bit-for-bit the torch ops
    labels = torch.tensor([0.0, 0.25, 1.0, 1.6, 3.0, 3.7, -0.5])
    base, deltas = torch.randn(768, generator=g) * 0.02, torch.randn(3, 768, generator=g) * 0.05
    out, oref = hip.zeros((7, 768), F32), torch.zeros(7, 768)
    hip.aoe_interp(dev(hip, labels), dev(hip, base), dev(hip, deltas), out)
    REF.aoe_interp(labels, base, deltas, oref)
    hip.synchronize()
    assert (out.cpu() - oref).abs().max().item() < 1e-6
    # fp32-weight rows kernel with GELU (AOE projector)
    x, w, b = torch.randn(5, 768, generator=g), torch.randn(1536, 768, generator=g) / 28, torch.randn(1536, generator=g) * 0.1
    o = hip.zeros((5, 1536), F32)
    hip.linear_rows(dev(hip, x), dev(hip, w), dev(hip, b), o, 0, 2)
    hip.synchronize()
    assert (o.cpu() - Fn.gelu(Fn.linear(x, w, b))).abs().max().item() < 2e-5
    # purifier tail
    img, dis, gate = rnd((2, 16, 768), 47), rnd((2, 16, 768), 48), torch.rand(2, 16, 768, generator=g).to(F16)
    gam, bet = 1 + 0.1 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    o, oref = hip.zeros((2, 16, 768), F32), torch.zeros(2, 16, 768)
    hip.purifier_tail(dev(hip, img), dev(hip, dis), dev(hip, gate), dev(hip, gam), dev(hip, bet), o)
    REF.purifier_tail(img, dis, gate, gam, bet, oref)
    hip.synchronize()
    assert (o.cpu() - oref).abs().max().item() < 2e-5


@pytest.mark.parametrize("case", ["conv8x8", "down8x8", "lin4x4"])
def test_splitk_finish_with_groupnorm_apply(hip, case):
    """DADD_EPI_GNAPPLY: on a small map the split-K finish kernel also writes GroupNorm (+ SiLU) of the output.  Same
    arithmetic in the same order as splitk_finish_kernel followed by the single-launch GroupNorm: both tensors bit-identical
    to the two-launch path."""
    from progressive_stable_diffusion_amd import lib as L
    if case == "conv8x8":       # a ResNet conv of the 8x8 level: bias + time row + residual, SiLU
        b, hw, cin, n, taps, stride, sk, silu, fl = 4, 8, 1280, 1280, 9, 1, 9, 1, 7
    elif case == "down8x8":     # the stride-2 downsampler 16x16 -> 8x8: bias only
        b, hw, cin, n, taps, stride, sk, silu, fl = 2, 8, 640, 640, 9, 2, 5, 1, 1
    else:                       # a 1x1 linear on a 4x4 map, three slices (tail of the four-in-flight loop), no SiLU
        b, hw, cin, n, taps, stride, sk, silu, fl = 3, 4, 1280, 2560, 1, 1, 3, 0, 5
    hi = hw * stride
    x, w = rnd((b, hi, hi, cin), 120), rnd((n, taps * cin), 121, 1 / math.sqrt(taps * cin))
    bias, rowvec, res = rnd((n,), 122, 0.1, F32), rnd((b, n), 123, 0.3, F32), rnd((b, hw, hw, n), 124)
    gamma, beta = rnd((n,), 125, 0.1, F32) + 1.0, rnd((n,), 126, 0.1, F32)
    kw = dict(bias=dev(hip, bias), taps=taps, stride=stride, pad=taps // 9, tile_m=128, tile_n=160 if n % 160 == 0 else 128, splitk=sk)
    if fl & 2:
        kw["rowvec"] = dev(hip, rowvec)
    if fl & 4:
        kw["residual"] = dev(hip, res)
    xd, wd = dev(hip, x), dev(hip, w)
    o1, o2, y1, y2 = (hip.zeros((b, hw, hw, n), F16) for _ in range(4))
    hip.igemm(xd, wd, o1, flags=fl, partial=hip.zeros((sk * b * hw * hw * n,), F32), **kw)
    hip.groupnorm(o1, None, dev(hip, gamma), dev(hip, beta), y1, hip.zeros((b * L.GN_MAX_CHUNKS * 64,), F32), 32, 1e-5, silu)
    hip.igemm(xd, wd, o2, flags=fl | L.EPI_GNAPPLY | (L.EPI_GNAPPLY_SILU if silu else 0),
              partial=hip.zeros((sk * b * hw * hw * n,), F32), gn_apply=(y2, dev(hip, gamma), dev(hip, beta), 1e-5), **kw)
    hip.synchronize()
    o_ref, y_ref = torch.zeros(b, hw, hw, n, dtype=F16), torch.zeros(b, hw, hw, n, dtype=F16)
    REF.igemm(x, w, o_ref, flags=fl | (L.EPI_GNAPPLY_SILU if silu else 0), bias=bias, rowvec=rowvec if fl & 2 else None,
              residual=res if fl & 4 else None, taps=taps, stride=stride, pad=taps // 9, gn_apply=(y_ref, gamma, beta, 1e-5))
    close(o2, o_ref, 3e-3, 2e-3, f"finish+gn out {case}")
    close(y2, y_ref, 6e-3, 6e-3, f"finish+gn normalised {case}")
    assert torch.equal(o1.cpu(), o2.cpu()), case
    assert torch.equal(y1.cpu(), y2.cpu()), case
    with pytest.raises(ValueError):        # one K pass: there is no finish kernel to do it
        hip.igemm(xd, wd, o2, flags=fl | L.EPI_GNAPPLY, gn_apply=(y2, dev(hip, gamma), dev(hip, beta), 1e-5), **{**kw, "splitk": 1})


@pytest.mark.parametrize("case", ["halo64", "dma128x160", "dma64x160", "dma128x128", "reg64x160"])
def test_igemm_groupnorm_statistics_epilogue(hip, case):
    """DADD_EPI_GNSTAT on the three GEMM kernels: the chunk partials [B][Ho*Wo / (tile_m/2)][32][2] written by the
    epilogue equal the sums of the stored fp16 outputs (fp32 accumulation order differs: 1e-3 relative), and
    dadd_groupnorm_f16(ws_chunks=...) normalises from them exactly like the two-pass GroupNorm."""
    from progressive_stable_diffusion_amd import lib as L
    b = 2
    if case == "halo64":
        hw, cin, n, taps, tm, tn, tune = 64, 64, 320, 9, 128, 160, 0
    elif case == "dma128x160":
        hw, cin, n, taps, tm, tn, tune = 32, 320, 640, 1, 128, 160, 0
    elif case == "dma64x160":
        hw, cin, n, taps, tm, tn, tune = 32, 640, 640, 1, 64, 160, 0
    elif case == "dma128x128":
        hw, cin, n, taps, tm, tn, tune = 64, 128, 256, 9, 128, 128, 0
    else:
        hw, cin, n, taps, tm, tn, tune = 32, 320, 640, 1, 64, 160, L.TUNE_NODMA
    x, w = rnd((b, hw, hw, cin), 90), rnd((n, taps * cin), 91, 1 / math.sqrt(taps * cin))
    bias, rowvec, res = rnd((n,), 92, 0.1, F32), rnd((b, n), 93, 0.3, F32), rnd((b, hw, hw, n), 94)
    nchunk = hw * hw // (tm // 2)
    ws = hip.zeros((b * nchunk * 64,), F32)
    o = hip.zeros((b, hw, hw, n), F16)
    hip.igemm(dev(hip, x), dev(hip, w), o, bias=dev(hip, bias), rowvec=dev(hip, rowvec), residual=dev(hip, res), taps=taps,
              pad=taps // 9, flags=7 | L.EPI_GNSTAT | tune, tile_m=tm, tile_n=tn, gn_ws=ws, gn_nchunk=nchunk)
    o_ref = torch.zeros(b, hw, hw, n, dtype=F16)
    REF.igemm(x, w, o_ref, bias=bias, rowvec=rowvec, residual=res, taps=taps, pad=taps // 9, flags=7)
    hip.synchronize()
    close(o, o_ref, 3e-3, 2e-3, f"gnstat out {case}")
    oc = o.float().cpu().reshape(b, nchunk, -1, 32, n // 32)
    part = torch.stack([oc.sum(dim=(2, 4)), (oc * oc).sum(dim=(2, 4))], dim=-1)
    got = ws.cpu().reshape(b, nchunk, 32, 2)
    assert (got - part).abs().max().item() <= 1e-3 * part.abs().max().item() + 1e-3, case
    gamma, beta = rnd((n,), 95, 0.1, F32) + 1.0, rnd((n,), 96, 0.1, F32)
    y1, y2 = hip.zeros((b, hw, hw, n), F16), hip.zeros((b, hw, hw, n), F16)
    ws2 = hip.zeros((b * L.GN_MAX_CHUNKS * 64,), F32)
    hip.groupnorm(o, None, dev(hip, gamma), dev(hip, beta), y1, ws, 32, 1e-5, 1, ws_chunks=nchunk)
    hip.groupnorm(o, None, dev(hip, gamma), dev(hip, beta), y2, ws2, 32, 1e-5, 1)
    hip.synchronize()
    close(y1, y2.float().cpu(), 2e-3, 2e-3, f"gn from partials {case}")
    # split-K: the finish kernel writes the partials (chunks of 16 rows — the engine's choice — or 64, blocks of 160 / 128
    # columns; three slices: the four-slabs-in-flight loop with a clamped tail)
    for rows, sk in ((16, 3), (64, 2)):
        if case == "halo64" and rows == 16:      # one 64-channel chunk: the halo conv cannot split K, its own epilogue
            continue                             # writes the partials (64-row chunks)
        nch2 = hw * hw // rows
        ws3 = hip.zeros((b * (nch2 + 64) * 64,), F32)         # (+ room for the fold of > 128 chunks to 64)
        o3 = hip.zeros((b, hw, hw, n), F16)
        hip.igemm(dev(hip, x), dev(hip, w), o3, bias=dev(hip, bias), rowvec=dev(hip, rowvec), residual=dev(hip, res), taps=taps,
                  pad=taps // 9, flags=7 | L.EPI_GNSTAT | (tune & ~L.TUNE_NODMA), tile_m=tm, tile_n=tn, gn_ws=ws3, gn_nchunk=nch2,
                  splitk=sk, partial=hip.zeros((sk * b * hw * hw * n,), F32))
        y3 = hip.zeros((b, hw, hw, n), F16)
        hip.groupnorm(o3, None, dev(hip, gamma), dev(hip, beta), y3, ws3, 32, 1e-5, 1, ws_chunks=nch2)
        hip.synchronize()
        close(o3, o_ref, 3e-3, 2e-3, f"gnstat split-K out {case} rows {rows}")
        oc3 = o3.float().cpu().reshape(b, nch2, -1, 32, n // 32)
        part3 = torch.stack([oc3.sum(dim=(2, 4)), (oc3 * oc3).sum(dim=(2, 4))], dim=-1)
        assert (ws3.cpu()[:b * nch2 * 64].reshape(b, nch2, 32, 2) - part3).abs().max().item() <= 1e-3 * part3.abs().max().item() + 1e-3
        close(y3, y2.float().cpu(), 3e-3, 3e-3, f"gn from finish partials {case} rows {rows}")
    if case == "dma128x128":               # many chunks (VAE maps): folded to 64 per sample by gn_reduce_kernel first
        nbig = 320
        noise = torch.randn(b, nbig, 32, 2, generator=torch.Generator().manual_seed(5)) * 0.01
        part_big = noise.clone()
        part_big[:, :nchunk] += part           # same totals as the real partials, spread over 320 chunks
        part_big[:, -1] -= noise.sum(dim=1)
        wsb = hip.zeros((b * (nbig + 64) * 64,), F32)
        wsb[: b * nbig * 64].copy_(part_big.reshape(-1))
        yb = hip.zeros((b, hw, hw, n), F16)
        hip.groupnorm(o, None, dev(hip, gamma), dev(hip, beta), yb, wsb, 32, 1e-5, 1, ws_chunks=nbig)
        hip.synchronize()
        close(yb, y2.float().cpu(), 3e-3, 3e-3, "gn from 320 chunk partials (reduced)")
    with pytest.raises(ValueError):        # contract: the chunk count must match the path that writes the partials
        hip.igemm(dev(hip, x), dev(hip, w), o, taps=taps, pad=taps // 9, flags=L.EPI_GNSTAT, tile_m=tm, tile_n=tn,
                  gn_ws=hip.zeros((b * (nchunk + 1) * 64,), F32), gn_nchunk=nchunk + 1)
