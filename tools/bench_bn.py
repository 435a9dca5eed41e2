import sys, torch
sys.path.insert(0, '/root/repo')
from pitchextractor_amd import ops
dev = torch.device('cuda:0')
def timed(fn, n=6):
    ts=[]
    for r in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if r: ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
for (F,C,pool) in [(80,64,1),(80,64,2),(40,128,1),(40,128,2),(20,192,2),(10,256,4)]:
    for a16 in (False, True):
        with ops.matmul_bf16(True,'bf16',act16=a16):
            dt = ops.act_dtype()
            x = torch.randn(256,192,F,C,device=dev).to(dt)
            dy = torch.randn(256,192,F//pool,C,device=dev).to(dt)
            g,b = torch.ones(C,device=dev), torch.zeros(C,device=dev)
            st = ops.bn_train_stats(x, g, b, torch.zeros(C,device=dev), torch.ones(C,device=dev))
            dg, db = torch.zeros(C,device=dev), torch.zeros(C,device=dev)
            dx = torch.empty_like(x)
            t_b = timed(lambda: ops.bn_act_pool_bwd(x, dy, st, dg, db, pool=pool, slope=0.01, dx=dx))
            y = torch.empty(256,192,F//pool,C,device=dev,dtype=dt)
            t_f = timed(lambda: ops.bn_act_pool_fwd(x, st, pool=pool, slope=0.01, out=y))
            esz = 2 if a16 else 4
            nb = x.numel()*esz; nd = dy.numel()*esz
            bytes_b = 2*(nb+nd) + nb      # two sweeps of x and dy, one write of dx
            bytes_f = nb + nd
            print(f"F={F} C={C} pool={pool} {'bf16' if a16 else 'fp32'}: bwd {t_b:.3f} ms = {bytes_b/t_b/1e9:.2f} TB/s   fwd {t_f:.3f} ms = {bytes_f/t_f/1e9:.2f} TB/s")
