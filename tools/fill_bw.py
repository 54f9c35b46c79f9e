import torch, statistics
for n in (4096*4096, 16384*8192, 16384*16384):
    buf = torch.empty(n, dtype=torch.int32, device="cuda")
    ts=[]
    for _ in range(8):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); buf.fill_(7); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms=statistics.median(ts[2:])
    print(f"torch fill {n*4/1e6:.0f} MB: {ms:.4f} ms {n*4/ms/1e6:.0f} GB/s")
    ts=[]
    for _ in range(8):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); buf.zero_(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms=statistics.median(ts[2:])
    print(f"torch zero_ (memset) {n*4/1e6:.0f} MB: {ms:.4f} ms {n*4/ms/1e6:.0f} GB/s")
