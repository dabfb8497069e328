"""Write bandwidth a plain fill reaches on this GPU for a buffer the size of one bench block of
complex128 samples (1.28 GB): the store floor the modulator is compared with in DESIGN.md."""
import torch
n=80_000_008
x=torch.empty((n,2),dtype=torch.float64,device='cuda')
for name,fn in [('fill_',lambda: x.fill_(1.0)),('zero_',lambda: x.zero_())]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for k in range(10):
        fn(); ev[k+1].record()
    torch.cuda.synchronize()
    ms=sorted(ev[k].elapsed_time(ev[k+1]) for k in range(10))
    print(name, ms[5], 'ms', n*16/ms[5]/1e6, 'GB/s')
