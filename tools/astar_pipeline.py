"""Scratch: throughput of D concurrent A* batches (one context + stream + host thread each)."""
import sys, os, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sea_current_amd as sc
from sea_current_amd import synth
fam = sys.argv[1]
occ_h = synth.salt_grid(1024, 1024, 0.05) if fam == "salt05" else synth.salt_grid(1024, 1024, 0.2) if fam == "salt20" else synth.block_grid(1024, 1024, 0.2)
occ = torch.from_numpy(occ_h).cuda()
c0 = sc.Context(0); d2_0 = c0.edt(occ); torch.cuda.synchronize()
s, g = synth.queries(d2_0.cpu().numpy() >= 1, 1024)
sd, gd = torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda()
for D in (1, 2, 4, 8):
    ctxs = [sc.Context(0, use_torch_stream=False) for _ in range(D)]
    d2s = [torch.empty((1024, 1024), dtype=torch.int32, device="cuda") for _ in range(D)]
    outs = [dict(path=torch.empty((1024, 4096), dtype=torch.int32, device="cuda"), len=torch.empty(1024, dtype=torch.int32, device="cuda"),
                 cost=torch.empty(1024, dtype=torch.int32, device="cuda"), status=torch.empty(1024, dtype=torch.int32, device="cuda")) for _ in range(D)]
    torch.cuda.synchronize()
    steps = 4 * D
    def work(j, n):
        for i in range(n):
            ctxs[j].edt(occ, out=d2s[j].view(1, 1024, 1024))
            ctxs[j].astar_batch(d2s[j], sd, gd, out=outs[j])
        ctxs[j].synchronize()
    ths = [threading.Thread(target=work, args=(j, 1)) for j in range(D)]
    [t.start() for t in ths]; [t.join() for t in ths]
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(j, steps // D)) for j in range(D)]
    [t.start() for t in ths]; [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print(fam, "depth", D, "steps", steps, "%.2f ms/step  %.0f plans/s" % (dt * 1e3 / steps, 1024 * steps / dt))
    for c in ctxs: c.close()
