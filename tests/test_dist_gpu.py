"""Data parallelism with REAL backwards: two ranks share the one MI355X of the test box (gloo carries the collective;
on a multi-GPU node the same code runs over RCCL).  Checked: (1) shard gradients combined by GradSync == the gradient of
the undivided batch for the AST / ViT encoders, with the all-reduce overlapped through grad_ready_hook, and in the frozen
phase only the head's slices cross the wire; (2) EEGNet keeps per-replica BatchNorm statistics (nn.DataParallel's
behaviour, EEGNet_tor.py:86-88) while its gradients are the global-batch mean; (3) Trainer_uni under data parallelism
still replays hipGraphs (compute graph -> eager all-reduce -> update graph), bit-equal to the eager schedule, and the
replicas stay identical."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
from eav_amd import dist as ed, synth, transformer as T
from eav_amd.optim import CrossEntropyLoss, FusedAdam
from tests.golden_util import tf_weights, eegnet_weights

rank, world, _ = ed.init_from_env("gloo")
assert world == 2
torch.cuda.set_device(0)
crit = CrossEntropyLoss()


def grads_of(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


# ---------------------------------------------------------------- (1) encoders: shard gradients -> global-batch gradient
for kind in ("vit", "ast"):
    cfg = T.make_config(kind, hidden=64, layers=2, heads=4, ff=128)
    W = tf_weights(31, T.param_shapes(cfg), std=0.08)
    B = 6
    x, y = (synth.mel_batch(40, B, cfg.W, cfg.H) if kind == "ast" else synth.frame_batch(40, B, cfg.H))
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    # uneven shards on purpose: rank 0 takes 4 items, rank 1 takes 2
    lo, hi = (0, 4) if rank == 0 else (4, 6)
    for freeze in (False, True):
        ref = T.Encoder(cfg, W).cuda().train()
        shard = T.Encoder(cfg, W).cuda().train()
        for m in (ref, shard):
            for k, p in m.named_parameters():
                p.requires_grad = (not freeze) or k.startswith("classifier.")
        crit(ref(xd).logits, yd).backward()
        want = grads_of(ref)
        shard._ensure_flat()
        sync = ed.GradSync([shard._flat[1]])
        sync.set_batch(hi - lo, B)
        sync.set_active(shard.head_grad_ranges() if freeze else None)
        shard.grad_ready_hook = sync.bucket            # overlapped: per-layer buckets from inside the backward
        crit(shard(xd[lo:hi]).logits, yd[lo:hi]).backward()
        sync()
        got = grads_of(shard)
        assert sorted(got) == sorted(want)
        for k in want:
            err = (got[k] - want[k]).abs().max().item()
            # (k_proj.bias has an analytically zero gradient - softmax is shift invariant - so it is pure rounding noise)
            assert err <= 1e-5 * want[k].abs().max().item() + 1e-7, (kind, freeze, k, err)
        nhead = sum(b - a for a, b in shard.head_grad_ranges())
        if freeze:
            assert sync.bytes_reduced == 4 * nhead, (sync.bytes_reduced, nhead)     # only the head crossed the wire
        else:
            assert sync.bytes_reduced == 4 * shard._flat[1].numel()

# ---------------------------------------------------------------- (2) EEGNet: per-replica BN, global-mean gradients
from eav_amd.eegnet import EEGNet_tor, Trainer_uni
S = 500
sd = eegnet_weights(61, S)
x, y = synth.eeg_batch(610, 8, 30, S)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()


def load(m):
    full = m.state_dict()
    full.update({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.load_state_dict(full)
    return m


def fresh():
    return load(EEGNet_tor(5, Chans=30, Samples=S, dropoutRate=0.0)).cuda().train()


per_shard = []
for r in range(2):                      # both ranks compute both shards' gradients and BN statistics on their own
    m = fresh()
    crit(m(xd[4 * r:4 * r + 4]), yd[4 * r:4 * r + 4]).backward()
    per_shard.append((grads_of(m), {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
m = fresh()
m._ensure_flat()
sync = ed.GradSync([m._flat[1]])
crit(m(xd[4 * rank:4 * rank + 4]), yd[4 * rank:4 * rank + 4]).backward()
sync()
for k, g in grads_of(m).items():
    want = 0.5 * (per_shard[0][0][k] + per_shard[1][0][k])
    assert (g - want).abs().max().item() <= 1e-6 * want.abs().max().item() + 1e-10, k
for k, v in per_shard[rank][1].items():          # this replica's running statistics come from ITS shard only
    assert torch.equal(m.state_dict()[k], v), k
assert not torch.equal(per_shard[0][1]["firstBN.running_mean"], per_shard[1][1]["firstBN.running_mean"])

# ---------------------------------------------------------------- (3) Trainer_uni: graph replay under data parallelism
xt, yt = synth.eeg_batch(620, 40, 30, S)
finals = []
for use_graph in (False, True):
    torch.manual_seed(99)                                    # same index order on both ranks and in both modes
    mm = load(EEGNet_tor(5, Chans=30, Samples=S, dropoutRate=0.5))
    lo, hi = ed.shard_batch(40, rank, world)                 # each rank trains on its half of the subject's trials
    tr = Trainer_uni(mm, [xt[lo:hi], yt[lo:hi], xt[:4], yt[:4]], lr=1e-3, batch_size=4, num_epochs=2, device="cuda")
    tr.use_graph = use_graph
    ed.attach(tr)
    assert tr.grad_sync is not None and tr.use_graph == use_graph
    tr.train()
    torch.cuda.synchronize()
    if use_graph:
        gs = [g for g in tr._graphs.values() if g.graph is not None]
        assert gs and all(g.graph_update is not None for g in gs)    # two graphs with the all-reduce between them
    finals.append({k: v.clone() for k, v in mm.state_dict().items()})
for k in finals[0]:
    assert torch.equal(finals[0][k], finals[1][k]), ("graph vs eager", k)
# replicas hold identical parameters (BatchNorm buffers are per replica by design)
for k, p in mm.named_parameters():
    both = [torch.empty_like(p) for _ in range(2)]
    dist.all_gather(both, p.detach())
    assert torch.equal(both[0], both[1]), ("replicas diverged", k)
dist.barrier()
dist.destroy_process_group()
open(os.path.join(OUT, f"ok_{rank}"), "w").write("ok")
"""


def test_two_ranks_on_one_gpu(tmp_path):
    script = tmp_path / "worker.py"
    body = textwrap.indent(textwrap.dedent(WORKER), "    ")
    script.write_text(f"ROOT = {ROOT!r}\nOUT = {str(tmp_path)!r}\nimport os, traceback\ntry:\n{body}\nexcept BaseException:\n"
                      "    open(os.path.join(OUT, 'err_' + os.environ.get('RANK', '0')), 'w').write(traceback.format_exc())\n"
                      "    raise\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = None
    for attempt in range(2):      # the worker takes seconds; a rendezvous that never completes (seen once on a pool box) gets one retry
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        try:
            r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                               capture_output=True, text=True, env=env, timeout=400, cwd=str(tmp_path))
            break
        except subprocess.TimeoutExpired:
            if attempt == 1:
                raise
    errs = "".join(open(tmp_path / f).read() for f in sorted(os.listdir(tmp_path)) if f.startswith("err_"))
    assert r.returncode == 0, (errs or (r.stdout[-3000:] + r.stderr[-6000:]))
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists(), r.stdout[-3000:] + r.stderr[-3000:]
