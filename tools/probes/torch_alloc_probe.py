"""The same through torch's caching allocator (what the host layer's buffers go through): fresh segments of growing size, each held.
    python tools/probes/torch_alloc_probe.py"""
import time
import torch
torch.cuda.set_device(0)
x = torch.zeros(1, device='cuda')
torch.cuda.synchronize()
held = []
for mb in [76, 76, 1, 76, 76, 471, 400, 100, 100, 512, 512, 1024, 1024, 2048, 2048]:
    t = time.perf_counter()
    held.append(torch.empty(mb << 20, dtype=torch.uint8, device='cuda'))
    torch.cuda.synchronize()
    print('%5d MB on top of %6.0f MB reserved: %8.3f ms' % (mb, torch.cuda.memory_reserved() / 2**20 - mb, 1e3 * (time.perf_counter() - t)))
