import cProfile, pstats, sys, os, torch, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from mmmot_amd import TrackingLoss, TrackingNet
from mmmot_amd.synth import make_pair
from mmmot_amd.weights import init_module
dev = torch.device('cuda', 0)
model = TrackingNet(**dict(bench.BASE_KW, score_fusion_arch='C', affinity_op='multiply', softmax_mode='none'))
init_module(model, seed=0); model.to(dev).train(); model.freeze_appearance = ('--whole' not in sys.argv)
crit = TrackingLoss(detloss_type='bce', linkloss_type='l2', det_ratio=1.5, trans_ratio=0.001)
opt = torch.optim.SGD(model.parameters(), lr=1e-4)
N, M = 10, 12
dets, info, ds = make_pair(N, M, 224, 300, seed=4000, ragged=True)
dets, info = dets.to(dev), {k: v.to(dev) for k, v in info.items()}
g = torch.Generator().manual_seed(1); L = N + M
gt = [(torch.rand(L, generator=g) > 0.3).float().to(dev), [(torch.rand(1, N, M, generator=g) > 0.9).float().to(dev)],
      (torch.rand(L, generator=g) > 0.6).float().to(dev), (torch.rand(L, generator=g) > 0.6).float().to(dev)]
def step():
    det, links, new, end, trans = model(dets, info, ds)
    loss = crit(ds, gt[0], gt[1], gt[2], gt[3], det, links, new, end, trans)
    opt.zero_grad(); loss.backward(); opt.step(); torch.cuda.synchronize()
step(); step()
t0 = time.perf_counter(); step(); step(); print('wall per step: %.1f ms' % ((time.perf_counter() - t0) / 2 * 1e3))
pr = cProfile.Profile(); pr.enable(); step(); step(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
