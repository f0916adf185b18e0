"""Scratch: cycles per phase of the A* kernel's wide steps for the slowest queries (needs a library built with
-DASTAR_STAMPS as sea-current_amd/libsc_stamps.so; the stamps overwrite the first words of each path)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
sc.LIB_PATH = os.path.join(ROOT, "sea-current_amd", "libsc_stamps.so")
ctx = sc.Context(0)
names = (["loop overhead, refills", "pop + issue", "wait load", "successor list", "round-0 prep", "wait atomic", "pushes", "-", "narrow steps", "next level", "-", "-"] if os.environ.get("SC_ASTAR_DUAL") == "0" else
         ["loop overhead", "-", "-", "-", "-", "-", "wide steps", "level-end wait", "narrow steps", "next level", "refills", "path extraction"])
for fam in sys.argv[1].split(","):
    occ = synth.salt_grid(1024, 1024, 0.05) if fam == "salt05" else synth.salt_grid(1024, 1024, 0.2) if fam == "salt20" else synth.block_grid(1024, 1024, 0.2)
    d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
    s, g = synth.queries(d2.cpu().numpy() >= 1, 1024)
    out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()); torch.cuda.synchronize()
    ex, pop, kc, stp = ctx.astar_debug_stats(1024)
    path = out["path"].cpu().numpy()
    for k in list(np.argsort(-kc)[:2]) + list(np.argsort(kc)[[512, 256]]):
        st = path[k, :20].astype(np.int64)
        print('   wavefront 0: %d sleeps on a full hand-over ring, %d sleeps at level ends | wavefront 1: %d batches, %d records, %d idle polls' % (st[14], st[15], st[16], st[17], st[18]))
        nwide, nrounds = st[12], st[13]
        print(fam, "query %d (path %d cells): %d kcycles, %d steps of which %d wide (%d push rounds), %d popped, %d expanded" % (k, int(out["len"][k]), kc[k], stp[k], nwide, nrounds, pop[k], ex[k]))
        print('   refills: %d' % st[19])
        for i in range(12):
            print("     %-20s %7d kcycles  %5.1f %%   %6.0f cycles per wide step" % (names[i], st[i], 100.0 * st[i] / max(kc[k], 1), st[i] * 1024.0 / max(nwide, 1)))
