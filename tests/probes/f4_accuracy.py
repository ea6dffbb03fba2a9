"""Accuracy gate of the F(4x4,3x3) route, checked before the kernel existed (CPU, minutes): the ORACLE network of SimplePose-R50 with its stage-2 / 3 3x3 layers replaced by an
fp32 emulation of F(4x4,3x3) against the float64 network.  A checker script, not a collected test (it lives under tests/ because only tests may run the oracle).

    python tests/probes/f4_accuracy.py"""
import os, sys, numpy as np, torch, torch.nn as nn, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vatl4pose-wacv2024_amd"))
from oracle import nets, synth
torch.set_num_threads(8)
BT=torch.tensor([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],dtype=torch.float64)
G=torch.tensor([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],dtype=torch.float64)
AT=torch.tensor([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],dtype=torch.float64)
class F4Conv(nn.Module):
    def __init__(self, conv):
        super().__init__()
        self.U=torch.einsum('xi,ocij,yj->xyoc',G,conv.weight.detach().double(),G).float()   # filter transform in f64, stored f32
    def forward(self,x):
        n,c,h,w=x.shape
        xp=F.pad(x,(1,1,1,1))
        t=xp.unfold(2,6,4).unfold(3,6,4)              # n,c,th,tw,6,6
        bt=BT.float()
        V=torch.einsum('xi,nchwij->nchwxj',bt,t)      # fp32 row pass
        V=torch.einsum('nchwxj,yj->nchwxy',V,bt)      # fp32 col pass
        M=torch.einsum('nchwxy,xyoc->nohwxy',V,self.U)
        at=AT.float()
        Y=torch.einsum('ax,nohwxy->nohway',at,M)
        Y=torch.einsum('nohway,by->nohwab',Y,at)
        th,tw=Y.shape[2],Y.shape[3]
        return Y.permute(0,1,2,4,3,5).reshape(n,-1,th*4,tw*4)
def build(dtype):
    m=nets.SimplePoseRef(50); sd=synth.state_dict_for(m); m.load_state_dict(sd); return m.to(dtype).eval()
x=torch.from_numpy(synth.crops(4))
with torch.no_grad():
    ref64=build(torch.float64)(x.double())
    m32=build(torch.float32); y32=m32(x)
    mf=build(torch.float32)
    which=sys.argv[1:] or ['layer3']
    cnt=0
    for name in which:
        layer=getattr(mf.preact,name)
        for b in layer:
            if b.conv2.stride==(1,1):
                b.conv2=F4Conv(b.conv2); cnt+=1
    yf=mf(x)
def err(a): return float((a.double()-ref64).abs().max()/ref64.abs().max())
print("replaced",cnt,"layers:",which)
print("torch fp32 vs f64:",err(y32)," F4 fp32 vs f64:",err(yf), " argmax equal:", bool(torch.equal(yf.flatten(2).argmax(-1), ref64.flatten(2).argmax(-1))), bool(torch.equal(y32.flatten(2).argmax(-1), ref64.flatten(2).argmax(-1))))
