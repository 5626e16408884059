import sys, time, torch, numpy as np
sys.path.insert(0,'taichi-nerfs_amd')
from ngp_hip import ops, lib, synthetic
lib.load()
bits=torch.from_numpy(np.load('tests/golden/lego_density_bitfield.npz')['density_bitfield']).cuda()
o,d=synthetic.lego_rays(8192,seed=5); o=torch.from_numpy(o).cuda(); d=torch.from_numpy(d).cuda()
hits=ops.ray_aabb(o,d,0.5); noise=torch.rand(8192,device='cuda')
for _ in range(3): r=ops.march_train(o,d,hits,bits,noise,1,0.5,0.0,128,1024)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(50): r=ops.march_train(o,d,hits,bits,noise,1,0.5,0.0,128,1024)
torch.cuda.synchronize(); print('march_train total (count+scan+sync+write) us', (time.perf_counter()-t)/50*1e6, int(r[5]))
