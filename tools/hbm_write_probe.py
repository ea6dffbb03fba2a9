import torch, time
dev=torch.device("cuda:0")
n=4096*3*256*192
x=torch.empty(n,device=dev); y=torch.empty(n,device=dev)
def timed(fn,it=10):
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/it*1e3
for name,fn,byts in (("fill_ (write only)",lambda: x.fill_(1.5),4*n),("zero_ (memset)",lambda: x.zero_(),4*n),("copy_ (read+write)",lambda: y.copy_(x),8*n),("sum (read only)",lambda: x.sum(),4*n)):
    us=timed(fn); print(f"{name:24s} {us:8.1f} us  {byts/us/1e6:6.2f} TB/s")
