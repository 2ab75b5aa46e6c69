"""Measured GEMM tile table (scripts/tile_sweep.py on MI355X, in-situ per-launch timings of the B=4 / 512x512 step,
gpurun_out/r02b/tile_sweep.txt -> profiles/r02_b_tile_sweep.txt; see engine.plan_tiling).
key = (M, N, K, taps, geglu, residual, ups, stride) -> (tile_m, tile_n, splitk, tune).
Only entries that beat the rules of engine.choose_tiling / plan_tiling by > 3 % are listed; where a split-K variant
won by less than the ~1.5 us a second (finish) launch costs inside the graph, the single-pass variant is kept."""
TABLE = {
    (256, 1280, 2560, 1, False, False, 0, 1): (64, 64, 2, 0),
    (1024, 1280, 2560, 1, False, False, 0, 1): (128, 160, 4, 0),
    (1024, 10240, 1280, 1, True, False, 0, 1): (128, 128, 1, 64),
    (4096, 640, 320, 1, False, False, 0, 1): (64, 160, 1, 0),
    (4096, 640, 640, 1, False, False, 0, 1): (64, 160, 1, 0),
    (4096, 640, 640, 1, False, True, 0, 1): (64, 160, 1, 0),
    (4096, 640, 960, 1, False, False, 0, 1): (64, 160, 1, 0),
    (4096, 640, 1280, 1, False, False, 0, 1): (64, 160, 1, 0),
    (4096, 640, 1920, 1, False, False, 0, 1): (64, 160, 1, 0),
    (4096, 640, 2560, 1, False, True, 0, 1): (64, 160, 1, 0),
    (16384, 320, 960, 1, False, False, 0, 1): (128, 160, 1, 0),
    # second sweep (gpurun_out/r02i -> profiles/r02_i_tile_sweep.txt)
    (1024, 3840, 1280, 1, False, False, 0, 1): (128, 160, 1, 0),
    (4096, 1920, 640, 1, False, False, 0, 1): (128, 160, 1, 32),
    (16384, 320, 320, 1, False, False, 0, 1): (64, 160, 1, 32),
    (16384, 320, 640, 1, False, False, 0, 1): (128, 160, 1, 0),
    (16384, 320, 1280, 1, False, True, 0, 1): (128, 160, 1, 0),
    (16384, 2560, 320, 1, True, False, 0, 1): (128, 128, 1, 32),
}

# Round 3; these win over TABLE when engine.TILING_R3 is set (the A/B switch of the same-box comparison):
TABLE_R3 = {
    # round 3 (gpurun_out/r4t -> profiles/r03_zl_tile_sweep.txt): the sweep also tries 64-row tiles WITH K slices on the
    # 16x16 / 8x8 maps (two workgroups per CU keep twice the bytes in flight) - the finish kernels got cheaper this round.
    # Not taken: (1024, 1280, 1280, residual) -> (64, 160, 2, 0), 2 us per launch for 15 more finish launches per step
    (1024, 1280, 1920, 1, False, False, 0, 1): (64, 160, 2, 0),
    (1024, 1280, 11520, 9, False, True, 0, 1): (128, 160, 4, 0),
    (256, 1280, 11520, 9, False, True, 0, 1): (64, 160, 8, 0),
    (256, 1280, 11520, 9, False, False, 0, 1): (64, 160, 8, 0),
    (256, 1280, 5120, 1, False, True, 0, 1): (64, 160, 8, 0),
    (4096, 640, 2880, 9, False, False, 0, 1): (128, 160, 2, 0),
    (256, 1280, 2560, 1, False, False, 0, 1): (64, 64, 4, 0),
    (4096, 640, 1920, 1, False, False, 0, 1): (128, 160, 2, 0),
    (4096, 640, 2560, 1, False, True, 0, 1): (128, 160, 2, 0),
}
