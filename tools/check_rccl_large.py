"""tools/check_rccl_large.py -- does torch.distributed.all_to_all_single (RCCL) deliver large messages whole?
World size 1 on one GPU (send to self), message sizes around and above 2^31 / 2^32 bytes, as bytes and as int64."""
import os, sys, time
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29657")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
bad = 0
for gb in (1.0, 2.5, 4.5, 9.0, 13.77):
    n8 = int(gb * 1e9) // 32 * 4                       # int64 elements, a whole number of 32-byte records
    src = torch.randint(1, 1 << 62, (n8,), dtype=torch.int64, device=dev)
    for name, view in (("u8", torch.uint8), ("i64", torch.int64)):
        for mode in ("sync", "async"):
            dst = torch.zeros_like(src)
            s, d = src.view(view), dst.view(view)
            torch.cuda.synchronize()
            t = time.perf_counter()
            if mode == "sync":
                dist.all_to_all_single(d, s, [d.numel()], [s.numel()])
            else:
                w = dist.all_to_all_single(d, s, [d.numel()], [s.numel()], async_op=True)
                w.wait()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            wrong = int((dst != src).sum().item())
            first = int((dst != src).nonzero()[0].item()) * 8 if wrong else -1
            print(f"{gb:6.2f} GB as {name:3s} {mode:5s}: {dt * 1e3:8.1f} ms, {wrong} of {n8} words differ"
                  + (f" (first at byte {first}, {first / (n8 * 8):.3f} of the message)" if wrong else ""), flush=True)
            bad += wrong != 0
            del dst
    del src
dist.destroy_process_group()
sys.exit(1 if bad else 0)
