"""How much of the distance map changes from one frame of the dynamic-obstacle stream to the next (BASELINE configs[4]:
32 of the rectangles move by <= 2 cells per frame)?  An incremental ("dirty band") EDT can only save the bands in which
nothing changes.  CPU only: uses the oracle (checker) to compute the maps -- this is an analysis script, not product code."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np
from sea_current_amd import synth
from oracle import oracle
for W in (1024, 4096):
    rects = synth.block_rects(W, W)
    prev = oracle.edt(synth.raster_rects(rects, W, W))
    fr_cells, fr_bands = [], []
    for f in range(1, 5):
        rects = synth.move_rects(rects, f, W, W)
        cur = oracle.edt(synth.raster_rects(rects, W, W))
        ch = cur != prev
        fr_cells.append(ch.mean())
        fr_bands.append(ch.reshape(W // 32, 32, W).any(axis=(1, 2)).mean())
        prev = cur
    print("%d^2, %d rectangles, 32 moved per frame: cells of d2 that change %.1f %% (mean of 4 frames), 32-row bands with a change %.1f %%" % (
        W, rects.shape[0], 100 * np.mean(fr_cells), 100 * np.mean(fr_bands)))
