"""Cost of the per-step token exchange through torch.distributed's "nccl" backend (= RCCL) on a communicator of ONE rank
(the only one a one-GPU box can form): host time per call and device-side completion for dist.gather (grouped send / recv),
dist.all_gather_into_tensor (one collective kernel) and a plain device copy.  usage: python profiles/rccl_probe.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29551')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
tok = torch.arange(32 * 256, dtype=torch.int32, device=dev).view(32, 256)
bufs = [torch.empty_like(tok)]
flat = torch.empty(1, 32, 256, dtype=torch.int32, device=dev)
N = 200


def run(name, fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'{name:46s} host {1e6 * t_host / N:8.1f} us/call   until the device is done {1e6 * t_all / N:8.1f} us/call')


run('dist.gather(tokens, [buf], dst=0)', lambda: dist.gather(tok, bufs, dst=0))
run('dist.all_gather_into_tensor(flat, tokens)', lambda: dist.all_gather_into_tensor(flat, tok))
run('dist.all_gather([buf], tokens)', lambda: dist.all_gather(bufs, tok))
run('buf.copy_(tokens)', lambda: bufs[0].copy_(tok))
w = dist.all_gather_into_tensor(flat, tok, async_op=True)
w.wait()
run('all_gather_into_tensor(async_op=True) + wait()', lambda: dist.all_gather_into_tensor(flat, tok, async_op=True).wait())
side = torch.cuda.Stream()


def on_side():
    with torch.cuda.stream(side):
        dist.all_gather_into_tensor(flat, tok)


run('all_gather_into_tensor on a side stream', on_side)
assert torch.equal(flat[0], tok) and torch.equal(bufs[0], tok)

# does a call return before its input is ready?  A long job on a side stream, the current stream waits for it, then the call
big = torch.randn(8192, 8192, device=dev)


def behind_busy_stream(name, fn):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(6):
            big @ big                                            # a few ms of device work
        ev = torch.cuda.Event()
        ev.record(side)
    torch.cuda.current_stream().wait_event(ev)
    t0 = time.perf_counter()
    fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f'{name:46s} returned after {1e3 * t_host:7.3f} ms; device idle after {1e3 * (time.perf_counter() - t0):7.3f} ms')


behind_busy_stream('buf.copy_(tokens) behind a busy stream', lambda: bufs[0].copy_(tok))
behind_busy_stream('dist.gather behind a busy stream', lambda: dist.gather(tok, bufs, dst=0))
behind_busy_stream('dist.all_gather_into_tensor behind a busy stream', lambda: dist.all_gather_into_tensor(flat, tok))
behind_busy_stream('dist.all_gather behind a busy stream', lambda: dist.all_gather(bufs, tok))
dist.destroy_process_group()
