"""Run-to-run spread of the eager and the hipGraph-replayed training trajectory of tests/test_gpu_graph.py (fp32 atomics in the weight
gradients): prints the pairwise loss / parameter / EMA differences of six runs.  usage: graph_variation.py (on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_graph as T
dev = torch.device("cuda:0")
runs = []
for graph in (False, True, False, True, False, True):
    model, opt, ema, stepper = T._setup(dev, graph)
    losses = []
    for it in range(8):
        if it == 5:
            for g in opt.param_groups:
                g["lr"] = 0.001; g["momentum"] = 0.8
        out = stepper()
        losses.append(float(out["tot_loss"].item()))
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    eflat = torch.cat([p.detach().reshape(-1) for p in ema.ema.parameters()]).cpu().numpy()
    runs.append((graph, np.array(losses), flat, eflat))
for i in range(len(runs)):
    for j in range(i + 1, len(runs)):
        gi, li, pi, ei = runs[i]; gj, lj, pj, ej = runs[j]
        print(f"{'G' if gi else 'E'}{i} vs {'G' if gj else 'E'}{j}: loss rel first2 {np.abs(lj[:2]/li[:2]-1).max():.2e} all {np.abs(lj/li-1).max():.2e}  param {np.abs(pj-pi).max()/np.abs(pi).max():.2e}  ema {np.abs(ej-ei).max()/np.abs(ei).max():.2e}")
