import numpy as np, sys
sys.path.insert(0,'/root/repo')
import motionplanning_jl_amd as mp
w = mp.workloads.north_star()
X, r = w.X, w.r
lo, hi = w.lohi[:,0,:], w.lohi[:,1,:]
rng = np.random.default_rng(0)
from scipy.spatial import cKDTree
tree = cKDTree(X)
res = []
for t in range(300):
    # a "tile": 64 samples of one grid cell (side 0.2)
    c = rng.integers(0,5,size=6)*0.2
    inside = np.flatnonzero(np.all((X >= c) & (X < c+0.2), axis=1))
    if len(inside) < 64: continue
    tile = inside[:64]
    tl, th = X[tile].min(0), X[tile].max(0)
    surv = np.flatnonzero(np.all((hi >= tl - r) & (lo <= th + r), axis=1))
    # hits of the tile: all (q, c) pairs within r; a drain = 64 of them that come from one or two candidate cells
    nb = tree.query_ball_point(X[tile], r)
    pairs = np.array([(q, cc) for q, l in zip(tile, nb) for cc in l if cc != q])
    # group candidates by their cell, take drains of 64 consecutive hits in candidate-cell order
    cellid = (np.floor(X[pairs[:,1]]*5).astype(int) * (5**np.arange(6))).sum(1)
    o = np.argsort(cellid, kind='stable'); pairs = pairs[o]
    def mask(idx):
        x = X[idx]
        return np.all((hi[None,:,:] >= x[:,None,:] - r) & (lo[None,:,:] <= x[:,None,:] + r), axis=2)  # (n, M)
    for s in range(0, len(pairs)-64, 64*7):
        d = pairs[s:s+64]
        mq, mc = mask(d[:,0]), mask(d[:,1])
        both = mq & mc
        # exact: boxes met by the segment box
        sl = np.minimum(X[d[:,0]], X[d[:,1]]); sh = np.maximum(X[d[:,0]], X[d[:,1]])
        met = np.all((hi[None] >= sl[:,None,:]) & (lo[None] <= sh[:,None,:]), axis=2)
        # hashed 64-bit union
        uni = np.flatnonzero(both.any(0))
        hb = set((uni % 64).tolist())
        surv_h = [b for b in surv if (b % 64) in hb]
        res.append((len(surv), both.sum(1).mean(), len(uni), len(surv_h), met.sum(1).mean(), met.any(0).sum()))
res = np.array(res, float)
print("drains", len(res))
print("tile-cull survivors %.1f | per-lane mask_q&mask_c %.2f | union over the drain %.1f | hashed-union-filtered survivors %.1f | boxes met per hit %.3f | union of met %.2f" % tuple(res.mean(0)))
