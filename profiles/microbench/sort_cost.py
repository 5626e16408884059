import torch, time
def t(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); s=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-s)/n*1e6
M=1<<20
k64=torch.randint(0,2**21,(M,),device='cuda')
k32=k64.int(); k16=(k64>>9).short()
print('argsort int64', t(lambda: torch.argsort(k64)))
print('sort int64', t(lambda: torch.sort(k64)))
print('sort int32', t(lambda: torch.sort(k32)))
print('sort int16 (12-bit keys)', t(lambda: torch.sort(k16)))
print('sort int32 stable', t(lambda: torch.sort(k32, stable=True)))
print('bincount 4096', t(lambda: torch.bincount((k64>>9), minlength=4096)))
print('unique_consecutive', t(lambda: torch.unique_consecutive(k32)))
