"""(round 2) does the result copy of pass i slow the fold-in of pass i + 1?  Times the fp32 fold-in SpMM of the ML-20M-shaped
pass (HIP events) alone and with a device-to-host copy in flight on another stream: 11 MB / 5.5 MB through the copy
engine (copy_ to pinned memory), and 11 MB written by a KERNEL straight into the mapped pinned buffer.
usage: python tools/probes/d2h_overlap_probe.py"""
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
from polara_amd.ops import HipOps
from polara_amd.synth import make_workload, csr_to_numpy
ops = HipOps('cuda:0')
csr, cfg = make_workload('ml20m', device='cuda:0')
c = csr_to_numpy(csr); del csr
A = ops.csr(c['indptr'], c['indices'], c['values'], c['shape'])
n_users, n_items = c['shape']
V32 = torch.randn(n_items, 52, dtype=torch.float32, device='cuda:0')
out = ops.empty(n_users, 52)
res = torch.randint(0, n_items, (n_users, 10), dtype=torch.int64, device='cuda:0')
res32 = res.to(torch.int32)
host = torch.empty((n_users, 10), dtype=torch.int64).pin_memory()
host32 = torch.empty((n_users, 10), dtype=torch.int32).pin_memory()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def fold_ms(before=None, reps=20):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if before is not None:
            with torch.cuda.stream(side):
                before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        ops.spmm(A, V32, out=out)
        e1.record(main)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def copy_ms(fn, reps=10):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    return ts[len(ts) // 2]


print('fold-in alone                         %.3f ms' % fold_ms())
print('  + copy engine, 11 MB int64          %.3f ms   (copy alone %.3f ms)' % (fold_ms(lambda: host.copy_(res, non_blocking=True)), copy_ms(lambda: host.copy_(res, non_blocking=True))))
print('  + copy engine, 5.5 MB int32         %.3f ms   (copy alone %.3f ms)' % (fold_ms(lambda: host32.copy_(res32, non_blocking=True)), copy_ms(lambda: host32.copy_(res32, non_blocking=True))))
# a kernel that writes the mapped pinned buffer: torch cannot address host memory from a device kernel, so go through the
# library's own row scatter if it exists
if hasattr(ops.lib, 'pk_scatter_rows_i64'):
    import ctypes as C
    perm = torch.randperm(n_users, device='cuda:0').to(torch.int64)
    def k():
        ops.lib.pk_scatter_rows_i64(C.c_void_p(torch.cuda.current_stream().cuda_stream), n_users, 10, C.c_void_p(res.data_ptr()),
                                    C.c_void_p(perm.data_ptr()), C.c_void_p(host.data_ptr()))
    print('  + kernel writing pinned memory      %.3f ms   (kernel alone %.3f ms)' % (fold_ms(k), copy_ms(k)))
    dev_out = torch.empty_like(res)
    def kd():
        ops.lib.pk_scatter_rows_i64(C.c_void_p(torch.cuda.current_stream().cuda_stream), n_users, 10, C.c_void_p(res.data_ptr()),
                                    C.c_void_p(perm.data_ptr()), C.c_void_p(dev_out.data_ptr()))
    print('  (same kernel, device destination: %.3f ms)' % copy_ms(kd))


# the bench loop's structure: pass kernels on the main stream, then an event, a side stream that waits for it and copies
def pipelined(mode, n=30):
    evs = []
    torch.cuda.synchronize()
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        ops.spmm(A, V32, out=out)
        e1.record(main)
        evs.append((e0, e1))
        res.add_(1)                                   # the pass "produces" its result
        if mode == 'none':
            continue
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            if mode in ('wait+copy', 'wait only'):
                side.wait_event(ready)
            if mode in ('wait+copy', 'copy only'):
                host.copy_(res, non_blocking=True)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs[5:])
    return ts[len(ts) // 2]


for mode in ('none', 'copy only', 'wait only', 'wait+copy'):
    print('pipelined, %-10s fold-in %.3f ms' % (mode, pipelined(mode)))
