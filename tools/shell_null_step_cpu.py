import sys, time, cProfile, pstats
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import torch
import shell_host_cpu as S
import beyond_deep_ensembles_amd as bde
ops=S.stub_ops()
torch.manual_seed(0)
shapes=S.resnet20_shapes()
def run(kind, prof=False):
    params=[torch.nn.Parameter(torch.randn(sh)*0.05) for sh in shapes]
    base = torch.optim.SGD(params, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=3e-4) if kind=="sgd" else torch.optim.Adam(params, lr=1e-3)
    opt=bde.SVGDOptimizer(params, lambda: None, base, particle_count=8, dataset_size=50000, _ops=ops)
    loss=torch.zeros(())
    fwd=lambda: loss
    bwd=lambda l: None
    for _ in range(50): opt.step(fwd,bwd)
    best=1e9
    for _ in range(7):
        t0=time.perf_counter()
        for _ in range(500): opt.step(fwd,bwd)
        best=min(best,(time.perf_counter()-t0)/500*1e6)
    print(kind, "null-closure step host: %.1f us"%best)
    if prof:
        pr=cProfile.Profile(); pr.enable()
        for _ in range(500): opt.step(fwd,bwd)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
run("sgd", prof=True); run("adam")
