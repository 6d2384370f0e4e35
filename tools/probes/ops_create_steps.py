"""What `HipOps()` costs a fresh process, piece by piece (bench.py's cold.ops_create_s).
    python tools/probes/ops_create_steps.py"""
import sys, time
sys.path.insert(0, '.')
t = time.perf_counter()
import torch
print('import torch                     %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); torch.cuda.set_device(0); torch.cuda.synchronize(); print('set_device + synchronize         %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter()
from polara_amd import _lib
import polara_amd.ops as O
print('import polara_amd.ops            %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); lib = _lib.load(); print('_lib.load() (dlopen + bindings)  %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); x = torch.empty(2, dtype=torch.int32, device='cuda'); torch.cuda.synchronize(); print('torch.empty (allocator start)    %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); x.zero_(); torch.cuda.synchronize(); print('first torch kernel (zero_)       %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); _lib.check(lib.pk_warm_up(), 'warm'); torch.cuda.synchronize(); print('pk_warm_up                       %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); ops = O.HipOps('cuda:0'); torch.cuda.synchronize(); print('HipOps() after all that          %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); y = torch.arange(10, device='cuda').max(); torch.cuda.synchronize(); print('arange + max (other torch TUs)   %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); y = torch.cat([x, x]); torch.cuda.synchronize(); print('cat                              %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); s = torch.cuda.Stream(); torch.cuda.synchronize(); print('torch.cuda.Stream()              %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter()
with torch.cuda.stream(s):
    x.zero_()
torch.cuda.synchronize(); print('first kernel on the new stream   %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); s2 = ops.monitor_stream(); torch.cuda.synchronize(); print('ops.monitor_stream()             %8.1f ms' % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); c = ops.recurrence_ctx() if hasattr(ops, 'recurrence_ctx') else None; torch.cuda.synchronize(); print('ops.recurrence_ctx()             %8.1f ms' % (1e3 * (time.perf_counter() - t)))
