import sys, torch, time
sys.path.insert(0,'taichi-nerfs_amd')
from ngp_hip import ops, lib
lib.load()
from modules.utils import morton3D
lv=ops.make_levels(2**19,16,16,1024,2)
table=torch.rand(lv.total_entries*2,device='cuda')
M=128**3//4
def bench(x,name):
    for _ in range(3): ops.hash_fwd_f32(x,table,lv)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): ops.hash_fwd_f32(x,table,lv)
    torch.cuda.synchronize(); print(name, (time.perf_counter()-t)/20*1e6,'us', x.shape[0])
coords=torch.randint(128,(2*M,3),dtype=torch.int32,device='cuda')
x=((coords.float()+torch.rand_like(coords.float()))/128).contiguous()
bench(x,'random cells')
idx=morton3D(coords).long()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): order=torch.argsort(idx)
torch.cuda.synchronize(); print('argsort', (time.perf_counter()-t)/10*1e6,'us')
bench(x[order].contiguous(),'morton-sorted cells')
# occupied-like: half the points drawn with replacement from 80k cells
occ=torch.randint(128,(80000,3),dtype=torch.int32,device='cuda')
c2=occ[torch.randint(80000,(M,),device='cuda')]
cc=torch.cat([coords[:M],c2]); x2=((cc.float()+torch.rand_like(cc.float()))/128).contiguous()
bench(x2,'uniform+occupied random')
o2=torch.argsort(morton3D(cc).long()); bench(x2[o2].contiguous(),'uniform+occupied sorted')
g=torch.arange(128,device='cuda',dtype=torch.int32)
allc=torch.stack(torch.meshgrid(g,g,g,indexing='ij'),-1).reshape(-1,3)
xa=((allc.float()+0.5)/128).contiguous(); bench(xa,'all cells meshgrid order')
oa=torch.argsort(morton3D(allc).long()); bench(xa[oa].contiguous(),'all cells morton order')
