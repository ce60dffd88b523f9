import sys; sys.path[:0]=["deepsphere-weather_amd",".","tests","tests/golden"]
import torch, numpy as np
from oracle import cheb_oracle as orc
from test_hip_parity import _rand_case
from modules.layers import ConvCheb
def run(V,B,Fin,Fout,K,dt):
    (rp,ci,va),x,w,b,gy=_rand_case(V,B,Fin,Fout,K,seed=77+Fin,bias=True)
    q=lambda a: torch.from_numpy(a).to(dt)
    xq,wq,bq,gyq=q(x),q(w),q(b),q(gy)
    lap=orc.coo_from_csr_arrays(rp,ci,va,(V,V))
    layer=ConvCheb(Fin,Fout,K,laplacian=lap,bias=True); layer.set_parameters(wq.float(),bq.float()); layer=layer.to("cuda").to(dt)
    va_q=layer.laplacian.coalesce().values().float().cpu().numpy()
    xx=xq.to("cuda").requires_grad_(True); y=layer(xx); y.backward(gyq.to("cuda")); torch.cuda.synchronize()
    f=lambda t:t.float().numpy()
    y64=orc.cheb_forward_f64(rp,ci,va_q,f(xq),f(wq),f(bq)); dx64,dw64,db64=orc.cheb_backward_f64(rp,ci,va_q,f(xq),f(wq),f(gyq),True)
    yy=y.detach().float().cpu().numpy(); e=np.abs(yy-y64)/np.abs(y64).max()
    print(V,B,Fin,Fout,K,dt,"y %.2e dx %.2e dw %.2e"%(orc.max_rel_err(y.float(),y64),orc.max_rel_err(xx.grad.float(),dx64),orc.max_rel_err(layer.weight.grad.float(),dw64)),
          "| y err by col-block32:", np.round(e.reshape(B*V,-1,min(32,Fout)).max(axis=(0,2)),3), "by row%128 block32:", np.round(e.reshape(-1,4,32,Fout)[:, :, :, :].max(axis=(0,2,3)),3) if (B*V)%128==0 else "")
for dt in (torch.float32, torch.bfloat16):
    run(768,4,64,128,5,dt); run(768,4,64,128,3,dt); run(768,4,64,64,5,dt); run(768,4,32,128,5,dt); run(768,4,64,256,2,dt)
