import os, sys, math
os.environ['COTR_HIP_EXPERIMENTAL']='1'
sys.path.insert(0, '/root/repo')
import torch
from cotr_amd import _lib
from tests import gpu_helpers as G
lib=_lib.load_library()
d=torch.device('cuda:0')
g=torch.Generator().manual_seed(0)
def pk(t):
    o=torch.empty_like(t); assert lib.cotr_op_split_h2(G.P(t),G.P(o),t.numel(),G.sptr())==0; return o
def up(t):
    o=torch.empty_like(t); assert lib.cotr_op_unsplit_h2(G.P(t),G.P(o),t.numel(),G.sptr())==0; return o
r=torch.randn(1<<20,generator=g).to(d)
print('unsplit(split(r)) max rel diff', float(((up(pk(r))-r).abs()/r.abs().clamp_min(1e-3)).max()))
(B,H,cin,cout,k,stride)=(2,64,64,256,1,1)
x=torch.relu(torch.randn(B,H,2*H,cin,generator=g)).to(d)
w=(torch.randn(cout,k*k*cin,generator=g)/math.sqrt(k*k*cin)).to(d)
sc,bi=(torch.rand(cout,generator=g)+0.5).to(d), torch.randn(cout,generator=g).to(d)
Ho=H//stride
for name, r in (('randn', torch.randn(B,Ho,2*Ho,cout,generator=g).to(d)), ('abs', torch.randn(B,Ho,2*Ho,cout,generator=g).abs().to(d)), ('const 1.5', torch.full((B,Ho,2*Ho,cout),1.5,device=d)),
                ('zero', torch.zeros(B,Ho,2*Ho,cout,device=d))):
    want=torch.empty(B,Ho,2*Ho,cout,device=d)
    assert lib.cotr_op_conv_cfg(G.P(x),G.P(w),G.P(sc),G.P(bi),G.P(r),0,G.P(want),B,H,H,cin,cout,k,stride,27,G.sptr())==0
    base=torch.empty_like(want)
    assert lib.cotr_op_conv_cfg(G.P(x),G.P(w),G.P(sc),G.P(bi),None,0,G.P(base),B,H,H,cin,cout,k,stride,27,G.sptr())==0
    xp,wp,rp=pk(x),pk(w),pk(r)
    lib.cotr_op_set_h2_flags(2)
    got=torch.empty_like(want)
    rc=lib.cotr_op_conv_cfg(G.P(xp),G.P(wp),G.P(sc),G.P(bi),G.P(rp),0,G.P(got),B,H,H,cin,cout,k,stride,46,G.sptr())
    lib.cotr_op_set_h2_flags(0)
    radd=(got-base)
    print(name, 'max diff', float((got-want).abs().max()), ' residual as added: ', radd.flatten()[:6].tolist(), ' true: ', r.flatten()[:6].tolist())
