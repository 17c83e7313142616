"""Diagnostic (needs a library built with EXTRA=-DPR_DIAG): pixel-activity statistics of the backward
compositing loop for one render configuration.  args: n w h f ts mode(rgb|feat|fused)"""
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering, _lib
from pogs_amd.scenes import make_scene
dev = torch.device("cuda:0")
n, w, h, f, ts, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
sc = make_scene(n, w, h, n_features=max(f, 1), seed=0)
p = {k: v.to(dev).requires_grad_(True) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
def step():
    if mode == "rgb":
        r, a, meta = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, w, h, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=ts)
    elif mode == "feat":
        r, a, meta = rendering.rasterization(p["means"], p["quats"].detach(), p["ls"].exp().detach(), torch.sigmoid(p["ol"]).detach(), p["f"], vm, K, w, h, packed=False, tile_size=ts)
    else:
        r, a, meta = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, w, h, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=ts, features=p["f"])
    (r.square().mean() + a.mean()).backward()
    return meta
step(); torch.cuda.synchronize(); lib.pr_diag_read(out, 1)
meta = step(); torch.cuda.synchronize(); lib.pr_diag_read(out, 1)
it, npx, nslots, c_alpha, c_body, nlanes, c_tail, c_wave = [int(x) for x in out[:8]]
print(json.dumps({"cfg": sys.argv[1:], "I": meta["n_composited_isects"], "contributing_iters": it,
                  "half_tile_iters (PR_DIAG=1 builds)": c_alpha, "iterations_entered (PR_DIAG=1)": c_body,
                  "batches_staged (PR_DIAG=1)": c_tail, "list_entries_looked_at (PR_DIAG=1)": c_wave, "active_px_per_iter": npx / it,
                  "active_slots_per_iter": nslots / it, "active_lanes_per_iter": nlanes / it,
                  "cycles_per_contributing_iter": {"record+alpha": c_alpha / it, "contractions+geometry": c_body / it,
                                                   "readback+atomics": c_tail / it},
                  "wave_cycles_total_per_iter": c_wave / it}))
