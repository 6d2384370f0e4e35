"""(round 3, CPU study) Share of the stored entries that falls on the H most popular / highest-norm items of the ML-20M-shaped
catalogue: what an LDS-resident head of the fp32 factor image could serve in the fold-in (738 rows of 208 B fit in 150 KB).
Result: 256: 0.17, 512: 0.26, 738: 0.31, 1024: 0.37, 2048: 0.50, 4096: 0.65 (norm order ~ popularity order).  CPU only, ~3 min."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sps
from scipy.sparse.linalg import svds
from polara_amd.synth import make_workload, csr_to_numpy
csr, cfg = make_workload('ml20m', device='cpu')
c = csr_to_numpy(csr)
A = sps.csr_matrix((c['values'].astype(np.float64), c['indices'], c['indptr']), shape=c['shape'])
_, s, vt = svds(A, k=50)
V = vt.T
norms = np.linalg.norm(V, axis=1)
order = np.argsort(-norms, kind='stable')
inv = np.empty(len(order), np.int64); inv[order] = np.arange(len(order))
cnt = np.bincount(A.indices, minlength=A.shape[1])
pop = np.sort(cnt)[::-1]
byn = cnt[order]
tot = cnt.sum()
for H in (256, 512, 600, 738, 1024, 2048, 4096):
    print(H, 'popularity order share %.3f' % (pop[:H].sum() / tot), 'norm order share %.3f' % (byn[:H].sum() / tot))
