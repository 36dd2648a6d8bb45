import importlib, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
pkg = importlib.import_module("lins---lidar-inertial-slam_amd")
ieskf = importlib.import_module("lins---lidar-inertial-slam_amd.ieskf")
from oracle import oracle
from test_gpu_edge_cases import make_pair
prm = pkg.default_params()
rng = np.random.default_rng(202)
ctx = ieskf.IeskfContext(pkg.default_params(num_iter=30), device=0, max_batch=8, max_targets=30000, search="lds")
n_sl = int(rng.integers(50, 3000))
pair = make_pair(pkg, rng, int(rng.integers(1, 200)), int(rng.integers(1, 200)), n_sl, int(rng.integers(5, 600)), "dup")
print("sizes", pair.sizes())
surf, corner = ctx.correspondences(pair, pair.state, 0)
ws, wc = oracle.correspondences(prm, pair, pair.state, 0, oracle.NN_BRUTE)
i = 82
print("got", corner[i]); print("want", wc[i])
tg = pair.corner_last
print("tg[110]", tg[110], "tg[66]", tg[66])
bad = np.nonzero(corner["ind1"] != wc["ind1"])[0]; print("bad", bad)
# which original points have px == reported
px = corner["sel"][i,3]; py = corner["coeff"][i,0]
print("points with that xy:", np.nonzero((tg[:,0]==px)&(tg[:,1]==py))[0])
