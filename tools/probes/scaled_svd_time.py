import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.data import ArrayData
from polara_amd.models import ScaledSVD, SVDModel
from polara_amd.synth import make_workload, csr_to_numpy, csr_to_coo_triplets
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml20m', device='cuda:0')
c = csr_to_numpy(csr)
u = np.repeat(np.arange(c['shape'][0]), np.diff(c['indptr'])); i = c['indices'].astype(np.int64); v = c['values'].astype(np.float64)
d = ArrayData((u, i, v), n_users=c['shape'][0], n_items=c['shape'][1], test=(u, i, v))
for cls in (SVDModel, ScaledSVD):
    m = cls(d, ops=ops); m.verbose = False; m.rank = 50
    m.build(); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.build(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(cls.__name__, 'build %.3f s' % (t1 - t0), 'sigma1 %.3f' % m.factors['singular_values'][0])
