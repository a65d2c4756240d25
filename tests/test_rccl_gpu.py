"""The RCCL path itself, on the one MI355X of the test box: a ONE-rank `nccl` process group (backend "nccl" IS RCCL on
ROCm) created in a spawned child before anything touches the GPU in it.  Two ranks cannot share a GPU under RCCL
("duplicate GPU"), so tests/test_dist_gpu.py covers the 2-rank arithmetic over gloo and this file covers what gloo
cannot: `init_process_group("nccl", device_id=...)`, the asynchronous bucket all-reduces on the communicator's stream
issued from inside the encoder backward, the two-graph `GraphStep` with an RCCL call between the graphs, and
`Trainer_uni` / the fine-tune trainers under `dist.attach(force=True)`.  A sum over one rank is the identity, so every
result must be BIT-equal to the unsynchronised run.  The child also records what `NCCL_DEBUG=INFO` prints (RCCL version,
transport / algorithm lines) under gpurun_out/ for DESIGN.md section 6.
Replaces: nn.DataParallel wrap, EEGNet_tor.py:86-88, Transformer_Audio.py:59-60, Transformer_Vision.py:82-83."""
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

rank, world, local = ed.init_from_env("nccl", force=True)
assert (rank, world) == (0, 1) and dist.is_initialized()
assert ed.backend_name() == "nccl", ed.backend_name()
dev = torch.device("cuda", local)
crit = CrossEntropyLoss()

# ---------------------------------------------------------------- (0) the collective itself
t = torch.arange(1 << 20, dtype=torch.float32, device=dev)
want = t.clone()
w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
w.wait()
assert torch.equal(t, want)
# a process SUB-group over RCCL (the tail groups of eav_amd.dist.SubjectSchedule are torch.distributed.new_group communicators):
# created, reduced in, and a GradSync bound to it - identity on one rank, bit for bit
sub = dist.new_group(ranks=[0])
t2 = torch.arange(4096, dtype=torch.float32, device=dev) * 0.5
dist.all_reduce(t2, op=dist.ReduceOp.SUM, group=sub)
torch.cuda.synchronize()
assert torch.equal(t2, torch.arange(4096, dtype=torch.float32, device=dev) * 0.5)
gs = ed.GradSync([t2], group=sub, force=True)
gs.bucket(100, 2000)
gs()
torch.cuda.synchronize()
assert gs.world == 1 and gs.collectives == 3 and torch.equal(t2, torch.arange(4096, dtype=torch.float32, device=dev) * 0.5)
sched = ed.subject_schedule(1, 3)
assert sched.make_groups() == {} and sched.solo == [[1, 2, 3]] and sched.ideal_speedup() == 1.0
big = torch.randn(86_192_645, device=dev)              # the AST gradient buffer: 345 MB in one call
ref = big.clone()
dist.all_reduce(big)
torch.cuda.synchronize()
assert torch.equal(big, ref)
del big, ref


def grads_of(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


# ---------------------------------------------------------------- (1) encoders: bucket hook under RCCL, both precisions
for kind in ("vit", "ast"):
    cfg = T.make_config(kind, hidden=128, layers=2, heads=2, ff=256)        # head_dim 64: the fused split attention
    W = tf_weights(31, T.param_shapes(cfg), std=0.08)
    B = 4
    x, y = (synth.mel_batch(40, B, cfg.W, cfg.H) if kind == "ast" else synth.frame_batch(40, B, cfg.H))
    xd, yd = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    for prec in ("split", "fp32"):
        for freeze in (False, True):
            plain = T.Encoder(cfg, W).to(dev).train()
            synced = T.Encoder(cfg, W).to(dev).train()
            for m in (plain, synced):
                m.precision = prec
                for k, p in m.named_parameters():
                    p.requires_grad = (not freeze) or k.startswith("classifier.")
            crit(plain(xd).logits, yd).backward()
            want = grads_of(plain)
            synced._ensure_flat()
            sync = ed.GradSync([synced._flat[1]], force=True)
            assert sync.enabled and sync.world == 1
            sync.set_active(synced.head_grad_ranges() if freeze else None)
            synced.grad_ready_hook = sync.bucket        # async all-reduces on the RCCL stream, from inside the backward
            crit(synced(xd).logits, yd).backward()
            sync()
            torch.cuda.synchronize()
            got = grads_of(synced)
            assert sorted(got) == sorted(want)
            for k in want:
                assert torch.equal(got[k], want[k]), (kind, prec, freeze, k)
            nhead = sum(b - a for a, b in synced.head_grad_ranges())
            assert sync.bytes_reduced == 4 * (nhead if freeze else synced._flat[1].numel()), sync.bytes_reduced
            assert sync.collectives >= (1 if freeze else cfg.layers)

# ---------------------------------------------------------------- (2) Trainer_uni: compute graph -> RCCL -> update graph
from eav_amd.eegnet import EEGNet_tor, Trainer_uni
S = 500
sd = eegnet_weights(61, S)
xt, yt = synth.eeg_batch(620, 40, 30, S)


def load(m):
    full = m.state_dict()
    full.update({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.load_state_dict(full)
    return m


finals = []
for mode in ("plain", "eager+rccl", "graph+rccl"):
    torch.manual_seed(99)
    mm = load(EEGNet_tor(5, Chans=30, Samples=S, dropoutRate=0.5))
    tr = Trainer_uni(mm, [xt[:32], yt[:32], xt[32:], yt[32:]], lr=1e-3, batch_size=8, num_epochs=2, device=dev)
    tr.use_graph = mode != "eager+rccl"
    if mode != "plain":
        ed.attach(tr, force=True)
        assert tr.grad_sync is not None and tr.grad_sync.enabled
    tr.train()
    torch.cuda.synchronize()
    if mode == "graph+rccl":
        gs = [g for g in tr._graphs.values() if g.graph is not None]
        assert gs and all(g.graph_update is not None for g in gs)      # two graphs with the all-reduce between them
        assert tr.grad_sync.collectives > 0
    finals.append({k: v.clone() for k, v in mm.state_dict().items()})
for k in finals[0]:
    assert torch.equal(finals[0][k], finals[1][k]), ("eager+rccl vs plain", k)
    assert torch.equal(finals[0][k], finals[2][k]), ("graph+rccl vs plain", k)

# ---------------------------------------------------------------- (3) the fine-tune trainer (AudioModelTrainer's loop)
from eav_amd.finetune import FineTuneBase


class _Tuner(FineTuneBase):
    # the shared two-phase loop of AudioModelTrainer / ImageClassifierTrainer on an in-memory model

    def __init__(self, model, data, batch_size, device):
        self.device, self.batch_size = device, batch_size
        self.model = model.to(device)
        self.initial_lr = 1e-3
        self.optimizer = FusedAdam(self.model.parameters(), lr=1e-3, weight_decay=0.01, decoupled=True)
        self.loss_fn = CrossEntropyLoss()
        self.grad_sync = None
        self.train_dataloader = self._loader(data[0], data[1], True)
        self.test_dataloader = self._loader(data[2], data[3], False)

    def train(self, epochs, lr, freeze):
        self._enter_phase(lr, freeze)
        for e in range(epochs):
            self._train_one_epoch()
            self._keep_outputs(self._evaluate(), e == epochs - 1, freeze)


outs = []
for synced in (False, True):
    cfg = T.make_config("vit", hidden=128, layers=2, heads=2, ff=256)
    W = tf_weights(32, T.param_shapes(cfg), std=0.08)
    x, y = synth.frame_batch(41, 12, cfg.H)
    torch.manual_seed(5)
    tr = _Tuner(T.Encoder(cfg, W), [x[:8], y[:8], x[8:], y[8:]], 4, dev)
    if synced:
        ed.attach(tr, force=True)
        assert tr.grad_sync is not None and tr.model.grad_ready_hook is not None
    tr.train(1, 5e-4, True)
    tr.train(1, 5e-6, False)
    torch.cuda.synchronize()
    if synced:
        assert tr.grad_sync.collectives > 0
    outs.append(np.array(tr.outputs_test))
assert np.array_equal(outs[0], outs[1])

dist.barrier()
dist.destroy_process_group()
open(os.path.join(OUT, "ok_0"), "w").write("ok")
"""


def test_one_rank_rccl_group(tmp_path):
    script = tmp_path / "worker.py"
    body = textwrap.indent(textwrap.dedent(WORKER), "    ")
    script.write_text(f"ROOT = {ROOT!r}\nOUT = {str(tmp_path)!r}\nimport os, traceback\ntry:\n{body}\nexcept BaseException:\n"
                      "    open(os.path.join(OUT, 'err_0'), 'w').write(traceback.format_exc())\n    raise\n")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL,TUNING")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=1500,
                       cwd=str(tmp_path))
    # keep what RCCL says about itself (version, transports, algorithm / protocol choices) for DESIGN.md section 6
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        lines = [ln for ln in (r.stdout + r.stderr).splitlines() if "NCCL" in ln or "RCCL" in ln]
        keep = [ln for ln in lines if "AllReduce" not in ln][:200] + [ln for ln in lines if "AllReduce" in ln][:40]
        open(os.path.join(out_dir, "rccl_world1_info.txt"), "w").write("\n".join(keep) + "\n")
    except OSError:
        pass
    errs = open(tmp_path / "err_0").read() if (tmp_path / "err_0").exists() else ""
    assert r.returncode == 0, (errs or (r.stdout[-3000:] + r.stderr[-6000:]))
    assert (tmp_path / "ok_0").exists(), r.stdout[-3000:] + r.stderr[-3000:]
