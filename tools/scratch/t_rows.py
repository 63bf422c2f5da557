import torch, sys
sys.path.insert(0, '/root/repo')
from coarse3d_amd import ops
torch.manual_seed(0)
dev='cuda'
b,hs,ws,d=2,16,64,256
H,W=32,128
low=torch.randn(b,hs,ws,d,device=dev)
dense=ops.bilinear(low,H,W,out_dtype=torch.float32)
n=H*W
idx=torch.randint(0,b*n,(1000,),device=dev)
cnt=torch.tensor([900],device=dev,dtype=torch.int32)
rows=ops.bilinear_rows(low,H,W,idx,count=cnt)
ref=dense.view(b*n,d)[idx]
print('rows equal', torch.equal(rows[:900],ref[:900]), float(rows[900:].abs().max()))
# anchors form
A=50; T=7
img=torch.randint(0,b,(b*20,),device=dev,dtype=torch.int32)
aidx=torch.randint(0,n,(b*20,A),device=dev,dtype=torch.int32)
t=torch.tensor([T],device=dev,dtype=torch.int32)
o1,n1=ops.bilinear_rows(low,H,W,aidx,img=img,a=A,count=t,l2=True)
o2,n2=ops.gather_rows_l2(dense,img,aidx,t,b*20,A,n)
print('anchors equal', torch.equal(o1,o2), torch.equal(n1,n2))
# backward
dx=torch.randn(b*20*A,d,device=dev)
# make pixel sets disjoint across pairs: use img = pair-specific? pairs with same img may collide: use distinct pixel ranges
for p in range(b*20):
    aidx[p]=torch.randint(p*100,(p+1)*100,(A,),device=dev,dtype=torch.int32)
gs=torch.tensor([0.37],device=dev)
dfeat=torch.zeros(b,H,W,d,device=dev)
rm=torch.zeros((b*n+31)//32,device=dev,dtype=torch.int32)
ops.scatter_add_rows(dx,img,aidx,t,b*20,A,n,dfeat,gs,rowmask=rm)
d1=torch.empty_like(low); ops.bilinear_bwd(d1,dfeat,rowmask=rm)
d0=torch.empty_like(low); ops.bilinear_bwd(d0,dfeat)
drows,cmap,rm2=ops.scatter_rows_compact(dx,img,aidx,t,b*20,A,n,b,gs)
d2=torch.empty_like(low); ops.bilinear_bwd_rows(d2,drows,cmap,rm2,H,W)
print('bwd equal', torch.equal(d1,d2), torch.equal(d0,d2), torch.equal(rm,rm2), float(d2.abs().max()))
