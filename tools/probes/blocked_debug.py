import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
from polara_amd.solver import svd_topk
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml20m', device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
Y = ops.randn(c['shape'][0], 64, 1)
Z0 = ops.spmm(A.T, Y)
Z1 = ops.spmm(A.transpose_operator(), Y)
print('blocked vs plain max rel diff', float((Z0 - Z1).abs().max() / Z0.abs().max()), type(A.transpose_operator()).__name__)
_, s, V, st = svd_topk(ops, A, 50, verbose=True)
print(st)
