import sys, time, torch
sys.path.insert(0, ".")
from eav_amd import synth
from eav_amd.eegnet import EEGNet_tor
m = EEGNet_tor(5, Chans=30, Samples=10000).cuda().eval()
x = torch.from_numpy(synth.eeg_batch(1, 64, 30, 10000)[0]).cuda()
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def nog():
    m.fused_eval = True
    with torch.no_grad(): m(x)
print("eval forward, grad enabled (unfused): %.3f ms; no_grad (fused block 1): %.3f ms" % (t(lambda: m(x)), t(nog)))
