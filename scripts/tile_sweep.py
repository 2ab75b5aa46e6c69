#!/usr/bin/env python
"""In-situ tile sweep for the implicit GEMMs of one UNet step (B=4, 512x512).

For every distinct GEMM signature of the plan a list of (tile_m, tile_n, split-K, tune) candidates is formed;
candidate v of ALL signatures is installed through engine.TILING_OVERRIDE, the plan is rebuilt (packed weights are
shared), the step is launched eagerly with per-launch begin/end timestamps (backend.prof_begin) and every launch
(plus the split-K finish launch that belongs to it) is attributed to its signature.  In situ = inputs freshly written
by the previous kernel, weights cold — what the captured graph sees, unlike an isolated loop.
Writes the winners to progressive-stable-diffusion_amd/tiling_table.py and a report to gpurun_out/<tag>/tile_sweep.txt.

    python scripts/tile_sweep.py [tag] [--batch 4] [--image-size 512] [--no-write]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def candidates(key, N_CU=256):
    from progressive_stable_diffusion_amd import lib as L
    m, n, k, taps, geglu, residual, ups, stride = key
    nkt = k // 64
    out = []
    tns = [128] if geglu else ([160] if n % 160 == 0 else [128])
    for tn in tns:
        t128 = math.ceil(m / 128) * math.ceil(n / tn)
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > 1 and (geglu or nkt // sk < 4 or t128 * sk > 4 * N_CU):
                continue
            out.append((128, tn, sk, 0))
        if t128 > N_CU and not ups:
            out.append((128, tn, 1, L.TUNE_PERSIST))
            out.append((128, tn, 1, L.TUNE_PERSIST | L.TUNE_SHALLOW))   # persistent ring, grouped (row-fastest) tile order
        if not ups:
            out.append((64, tn, 1, 0))                      # 64-row LDS-DMA tiles
            out.append((64, tn, 1, L.TUNE_NODMA))           # 64-row register-staged tiles (two workgroups per CU)
            if geglu:
                out.append((64, tn, 1, L.TUNE_NODMA | L.TUNE_SHALLOW))
        out.append((128, tn, 1, L.TUNE_NODMA))
    if not geglu and not ups and n % 64 == 0:
        for sk in (1, 2, 4, 8, 16):
            if sk > 1 and (nkt // sk < 8 or math.ceil(m / 64) * (n // 64) * sk > 8 * N_CU):
                continue
            out.append((64, 64, sk, 0))
        if n % 160 == 0 and m <= 1024:       # the 16x16 / 8x8 maps: 64-row tiles with K slices (two workgroups per CU)
            for sk in (2, 4, 8):
                if nkt // sk >= 8 and math.ceil(m / 64) * (n // 160) * sk <= 4 * N_CU:
                    out.append((64, 160, sk, 0))
        if n % 128 == 0 and n % 160:
            out.append((64, 128, 1, 0))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag", nargs="?", default="sweep")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-write", action="store_true")
    a = ap.parse_args()
    from progressive_stable_diffusion_amd import engine as E
    from progressive_stable_diffusion_amd import weights as W
    from progressive_stable_diffusion_amd.backend import HipBackend
    import progressive_stable_diffusion_amd.diffusion_module_ip as DM
    dev = torch.device("cuda:0")
    be = HipBackend(dev)
    side = a.image_size // 8
    sd = W.init_state_dict(W.unet_shapes(), 0, gates={"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)})
    wcache = {}
    _, ac = DM.build_noise_schedule(DM.DiffusionIPConfig(1000, 0.00085, 0.012))
    g = torch.Generator().manual_seed(0)
    cond = (torch.randn(a.batch, 48, 768, generator=g) * 0.5).to(dev)
    lat = torch.randn(a.batch, 4, side, side, generator=g).to(dev)

    def measure():
        """-> {key: [us per op instance]} for the current override."""
        plan = E.UNetPlan(be, sd, a.batch, side, wcache=wcache)
        loop = E.DdimLoop(plan)
        plan.set_cond(cond, 0)
        loop.prepare(torch.linspace(999, 0, 50, dtype=torch.long), ac)
        be.copy_(plan.lat_in, lat)
        be.zero_(loop.step)
        plan.lam = 3.0
        plan.prepare_attn2(3.0)
        keys = []
        for fn, args, kw in plan.ops:
            if getattr(fn, "__name__", "") == "igemm":
                x, w, out = args[0], args[1], args[2]
                m = out.shape[0] * out.shape[1] * out.shape[2]
                keys.append(E.tiling_key(m, w.shape[0], w.shape[1], kw.get("taps", 1), bool(kw.get("flags", 0) & 8),
                                         kw.get("residual") is not None, kw.get("ups", 0), kw.get("stride", 1)))
        plan.run()
        be.synchronize()
        acc = {}
        total = 0.0
        for _ in range(a.steps):
            be.prof_begin()
            plan.run()
            rec = be.prof_end()
            total += sum(r[1] for r in rec)
            it = iter(keys)
            cur = None
            for name, us, _, _ in rec:
                if name.startswith(("igemm", "conv3x3_halo")):
                    cur = next(it)
                    acc.setdefault(cur, []).append(us)
                elif name == "splitk_finish_kernel" and cur is not None:
                    acc[cur][-1] += us
                else:
                    cur = None
        del loop, plan
        return {k: sum(v) / a.steps for k, v in acc.items()}, {k: len(v) // a.steps for k, v in acc.items()}, total / a.steps

    E.TILING_OVERRIDE.clear()
    saved_table = dict(E.TILING_TABLE)
    E.TILING_TABLE.clear()                       # sweep from the rules, not from a previous table
    E.TILING_TABLE_R3.clear()
    base, counts, base_total = measure()
    keys = list(base)
    cands = {k: candidates(k) for k in keys}
    nv = max(len(v) for v in cands.values())
    best = {k: (base[k], None) for k in keys}
    results = {k: [] for k in keys}
    for v in range(nv):
        E.TILING_OVERRIDE.clear()
        for k in keys:
            if v < len(cands[k]):
                E.TILING_OVERRIDE[k] = cands[k][v]
        try:
            t, _, _ = measure()
        except Exception as ex:      # noqa: BLE001  (a candidate the C side refuses: skip the round, report it)
            print(f"round {v}: {type(ex).__name__}: {ex}")
            for k in keys:           # retry one signature at a time to find the offender cheaply? no: drop the round
                pass
            continue
        for k in keys:
            if v < len(cands[k]):
                results[k].append((t[k], cands[k][v]))
                if t[k] < best[k][0]:
                    best[k] = (t[k], cands[k][v])
        print(f"round {v + 1}/{nv} done", flush=True)
    # verify the winners together
    E.TILING_OVERRIDE.clear()
    for k in keys:
        if best[k][1] is not None and best[k][0] < 0.97 * base[k]:       # keep the rule unless the win is > 3 %
            E.TILING_OVERRIDE[k] = best[k][1]
    final, _, final_total = measure()
    lines = [f"tile sweep B={a.batch} {a.image_size}x{a.image_size}: step kernel time rules {base_total / 1e3:.3f} ms -> table {final_total / 1e3:.3f} ms",
             f"{'M':>6s} {'N':>6s} {'K':>6s} t g r u s {'cnt':>3s} {'rule us':>8s} {'best us':>8s} {'final us':>8s}  best (tile_m, tile_n, sk, tune) | next"]
    for k in sorted(keys, key=lambda k: -base[k]):
        m, n, kk, taps, geglu, res, ups, stride = k
        top = sorted(results[k])[:3]
        lines.append(f"{m:6d} {n:6d} {kk:6d} {taps} {int(geglu)} {int(res)} {ups} {stride} {counts[k]:3d} {base[k]:8.1f} "
                     f"{best[k][0]:8.1f} {final[k]:8.1f}  {best[k][1]} | {[(round(t, 1), c) for t, c in top]}")
    out_dir = os.path.join(ROOT, "gpurun_out", a.tag)
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "tile_sweep.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if not a.no_write:
        table = dict(saved_table)
        table.update(E.TILING_OVERRIDE)
        body = ",\n".join(f"    {k!r}: {tuple(v)!r}" for k, v in sorted(table.items()))
        with open(os.path.join(out_dir, "tiling_table.py"), "w") as f:
            f.write('"""Measured GEMM tile table (written by scripts/tile_sweep.py on MI355X; see engine.plan_tiling).\n'
                    'key = (M, N, K, taps, geglu, residual, ups, stride) -> (tile_m, tile_n, splitk, tune)."""\n'
                    "TABLE = {\n" + body + "\n}\n")


if __name__ == "__main__":
    main()
