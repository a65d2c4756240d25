import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eav_amd import synth, transformer as T
from eav_amd.optim import CrossEntropyLoss
from tests.golden_util import tf_weights
crit = CrossEntropyLoss()
for kind in ("ast", "vit"):
    cfg = T.make_config(kind, hidden=128, layers=2, heads=2, ff=256)
    W = tf_weights(31, T.param_shapes(cfg), std=0.08)
    B = 4
    x, y = (synth.mel_batch(40, B, cfg.W, cfg.H) if kind == "ast" else synth.frame_batch(40, B, cfg.H))
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    first = None
    for rep in range(6):
        m = T.Encoder(cfg, W).cuda().train()
        m.precision = "split"
        if os.environ.get("NO_OVERLAP"):
            m.overlap_wgrad = False
        crit(m(xd).logits, yd).backward()
        torch.cuda.synchronize()
        g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        if first is None:
            first = g
        else:
            bad = [k for k in g if not torch.equal(g[k], first[k])]
            print(kind, "rep", rep, "differing tensors:", bad[:6], flush=True)
        bs = m._ws.bslots
        if rep == 0:
            print(kind, "max boost exponents per backward slot:", [int(bs[i, 3104:4128].view(torch.int32).max()) for i in range(bs.shape[0])])
        if rep == 0:
            print(kind, "any non-finite gradient:", [k for k in g if not torch.isfinite(g[k]).all()])
