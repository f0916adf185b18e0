"""Scratch harness: time EDT variants (different builds of the library) on the three map families."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
lib = sys.argv[1]
sc.LIB_PATH = os.path.join(ROOT, "sea-current_amd", lib)
W = H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ctx = sc.Context(0)
res = {}
FAMS = sys.argv[4].split(",") if len(sys.argv) > 4 else ("salt05", "salt20", "blocks")
for fam in FAMS:
    grids = torch.from_numpy(np.stack([
        synth.salt_grid(W, H, 0.05, seed=100 + i) if fam == "salt05" else
        synth.salt_grid(W, H, float(fam[6:]), seed=100 + i) if fam.startswith("sparse") else   # sparse2e-5 ...: rows the packed cascade cannot settle
        synth.salt_grid(W, H, 0.20, seed=100 + i) if fam == "salt20" else
        synth.block_grid(W, H, 0.20, seed=100 + i) for i in range(B)])).cuda()
    d2 = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.edt(grids, out=d2)
        ctx.synchronize()          # (a context adapts to open space when it is synchronised)
    torch.cuda.synchronize(); ctx.set_timing(True); ctx.reset_timing()
    for _ in range(20): ctx.edt(grids, out=d2)
    torch.cuda.synchronize()
    a, _ = ctx.get_timing(sc.K_EDT_COLBITS); b, _ = ctx.get_timing(sc.K_EDT_BAND); ctx.set_timing(False)
    res[fam] = dict(colbits_us=a / 20 * 1e3, band_us=b / 20 * 1e3, GBps=5 * B * W * H / ((a + b) / 20 * 1e-3) / 1e9)
print(lib, json.dumps(res))
