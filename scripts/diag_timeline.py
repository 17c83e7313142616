"""Diagnostic (needs a library built with EXTRA=-DPR_DIAG=3): launch timeline of the backward compositor --
when every wavefront started and ended, on which XCD / CU, and how long its tile's list was.
args: n w h f ts"""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering, _lib
from pogs_amd.scenes import make_scene
dev = torch.device("cuda:0")
n, w, h, f, ts = [int(x) for x in sys.argv[1:6]]
sc = make_scene(n, w, h, n_features=max(f, 1), seed=0)
p = {k: v.to(dev).requires_grad_(True) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
def step():
    r, a, meta = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, w, h, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=ts, features=p["f"], absgrad=True)
    (r.square().mean() + a.mean()).backward()
    return meta
for _ in range(3):
    step()
torch.cuda.synchronize()
n_tiles = ((w + ts - 1) // ts) * ((h + ts - 1) // ts)
ppl = _lib.raster_plan(3 + f + 1, ts, n_tiles, backward=True)[0]
n_waves = n_tiles * (4 // ppl if ts == 16 else 1)
buf = (ctypes.c_ulonglong * (4 * n_waves))()
assert lib.pr_diag_trace_read(buf, n_waves) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 4).astype(np.int64)
start, end, hw, length = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
dur = end - start
xcc = (hw >> 32) & 0xf
# the time-stamp counters of the XCDs are not synchronised: every XCD's timeline is taken relative to its first start
rel_start, rel_end = start.copy(), end.copy()
for x in np.unique(xcc):
    m = xcc == x
    rel_start[m] -= start[m].min()
    rel_end[m] -= start[m].min()
span = rel_end.max()
out = {"pixels_per_lane": ppl, "waves": int(n_waves), "span_ticks": int(span), "mean_dur": float(dur.mean()), "max_dur": int(dur.max()),
       "mean_resident_waves": float(dur.sum() / span), "list_len_mean": float(length.mean()), "list_len_max": int(length.max())}
bins = 25
occ = []
for b in range(bins):
    a, c = span * b // bins, span * (b + 1) // bins
    ov = np.clip(np.minimum(rel_end, c) - np.maximum(rel_start, a), 0, None).sum() / (c - a)
    occ.append(int(ov))
out["resident_waves_over_time"] = occ
out["per_xcc"] = {int(x): {"waves": int((xcc == x).sum()), "end_frac": round(float(rel_end[xcc == x].max() / span), 3),
                           "wave_ticks_frac": round(float(dur[xcc == x].sum() / dur.sum()), 3)} for x in np.unique(xcc)}
out["corr_dur_len"] = float(np.corrcoef(dur, length)[0, 1])
out["dur_of_block_index_deciles"] = [int(dur[i * n_waves // 10:(i + 1) * n_waves // 10].mean()) for i in range(10)]
out["start_of_block_index_deciles"] = [round(float(rel_start[i * n_waves // 10:(i + 1) * n_waves // 10].mean() / span), 3) for i in range(10)]
# per CU: when did the CU's last wavefront end
cu_key = (xcc << 16) | (hw & 0xff00) | ((hw >> 13) & 0x7) << 4
ends = np.array([rel_end[cu_key == k].max() / span for k in np.unique(cu_key)])
out["n_cus_seen"] = int(len(ends))
out["cu_last_end_frac_quantiles"] = [round(float(np.quantile(ends, q)), 3) for q in (0, 0.1, 0.25, 0.5, 0.75, 0.9, 1)]
spans = np.array([rel_end[cu_key == k].max() - rel_start[cu_key == k].min() for k in np.unique(cu_key)])
res = np.array([dur[cu_key == k].sum() / (rel_end[cu_key == k].max() - rel_start[cu_key == k].min()) for k in np.unique(cu_key)])
out["cu_span_ticks_quantiles"] = [int(np.quantile(spans, q)) for q in (0, 0.1, 0.5, 0.9, 1)]
out["cu_mean_resident_quantiles"] = [round(float(np.quantile(res, q)), 2) for q in (0, 0.1, 0.5, 0.9, 1)]
out["cu_first_start_rel_quantiles"] = [int(np.quantile([rel_start[cu_key == k].min() for k in np.unique(cu_key)], q)) for q in (0, 0.5, 0.9, 1)]
out["waves_per_cu_quantiles"] = [int(np.quantile([(cu_key == k).sum() for k in np.unique(cu_key)], q)) for q in (0, 0.5, 1)]
print(json.dumps(out))
