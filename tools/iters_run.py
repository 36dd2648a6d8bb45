import importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
n = 1024; iters = int(sys.argv[1]); search = sys.argv[2] if len(sys.argv) > 2 else "mr"
with ThreadPoolExecutor(16) as ex:
    pairs = list(ex.map(host.synth_pair, range(n)))
with ieskf.IeskfContext(pkg.default_params(num_iter=iters, fixed_iters=1), max_batch=n, max_targets=16384, search=search) as c:
    c.upload(pairs)
    for _ in range(3):
        c.run(); c.sync()
    print(iters, c.last_kernel_ms())
