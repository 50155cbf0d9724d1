import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mono_vifi_amd import ops
dev=torch.device("cuda:0")
B,H,W,C,s=36,192,640,64,2
g=torch.Generator().manual_seed(5)
amp=float(sys.argv[1])
up=lambda t: torch.nn.functional.interpolate(t,size=(H,W),mode="bilinear")
f1,f2=up(torch.randn(B,2,H//16,W//16,generator=g)*amp).to(dev),up(torch.randn(B,2,H//16,W//16,generator=g)*amp).to(dev)
m=torch.sigmoid(up(torch.randn(B,1,H//16,W//16,generator=g))).to(dev)
preps=ops.fusion_prep(f1,f2,m,[(H//s,W//s)])
h,w=H//s,W//s
gout=torch.randn(B,2*(C+42),h,w,device=dev)
lists=preps.lists.level(0)
gn=torch.empty(B,C,h,w,device=dev); gp=torch.empty_like(gn)
from mono_vifi_amd import _native as nat
def run():
    nat.check(nat.lib().mvf_fusion_level_bwd_lists(nat.ptr(gout),nat.ptr(lists),nat.ptr(gn),nat.ptr(gp),B,C,h,w,None),"x")
for _ in range(3): run()
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("amp",amp,"exp",os.environ.get("MVF_ANC_EXP"),"us",e0.elapsed_time(e1)/20*1e3)
