import torch, time
dev=torch.device('cuda:0')
n=800*1024*1024  # floats: 3.2 GB
a=torch.empty(n,device=dev); b=torch.empty(n,device=dev)
def t(fn,byts,name,it=10):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/it
    print(f"{name:28s} {ms*1e3:9.1f} us  {byts/ms/1e9:6.2f} TB/s")
t(lambda: a.zero_(), n*4, "zero_ 3.2GB (write)")
t(lambda: a.fill_(1.5), n*4, "fill_ 3.2GB (write)")
t(lambda: b.copy_(a), 2*n*4, "copy_ 3.2GB (r+w)")
t(lambda: a.sum(), n*4, "sum 3.2GB (read)")
t(lambda: torch.add(a,b,out=b), 3*n*4, "add out= (2r+1w)")
import ctypes
hip=ctypes.CDLL('libamdhip64.so')
t(lambda: hip.hipMemsetAsync(ctypes.c_void_p(a.data_ptr()),0,ctypes.c_size_t(n*4),ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), n*4, "hipMemsetAsync 3.2GB")
