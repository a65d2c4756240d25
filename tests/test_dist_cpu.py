"""Multi-process coverage of the N>1 path on CPU (gloo, world size 2): the gradient all-reduce
averages flat buffers, subject sharding covers 42 subjects exactly once, batch shards tile the batch."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_subject_and_batch_sharding():
    from eav_amd.dist import shard_batch, subjects_for_rank
    for world in (1, 2, 4, 8):
        seen = sorted(s for r in range(world) for s in subjects_for_rank(r, world))
        assert seen == list(range(1, 43))
        sizes = [len(subjects_for_rank(r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
        cover = []
        for r in range(world):
            lo, hi = shard_batch(70, r, world)
            cover += list(range(lo, hi))
        assert cover == list(range(70))


def test_subject_schedule_composes_rounds_and_groups():
    """SubjectSchedule: whole rounds one subject per rank + the remainder on groups of ranks (42 = 5 x 8 + 2 -> two 4-rank
    groups); every subject exactly once, ideal speed-ups as DESIGN.md section 7 states them."""
    from eav_amd.dist import subject_schedule
    for world in (1, 2, 3, 4, 5, 7, 8, 16, 42, 64):
        for n in (42, 1, 7):
            s = subject_schedule(world, n)
            seen = sorted([x for r in range(world) for x in s.solo[r]] + [g[0] for g in s.groups])
            assert seen == list(range(1, n + 1)), (world, n)
            members = [r for _, ranks in s.groups for r in ranks]
            assert len(members) == len(set(members)) and all(0 <= r < world for r in members)
            assert all(len(ranks) == s.group_size for _, ranks in s.groups)
            for r in range(world):
                g = s.group_of(r)
                assert s.subjects_of(r) == s.solo[r] + ([g[0]] if g else [])
    s8 = subject_schedule(8)
    assert s8.rounds == 5 and s8.groups == [(41, [0, 1, 2, 3]), (42, [4, 5, 6, 7])]
    assert abs(s8.ideal_speedup() - 8.0) < 1e-12                       # 42 / (5 + 1/4)
    assert abs(s8.ideal_speedup(1 / 2.7) - 42 / (5 + 1 / 2.7)) < 1e-12 and s8.ideal_speedup(1 / 2.7) > 7.6
    assert abs(subject_schedule(8, hybrid=False).ideal_speedup() - 7.0) < 1e-12     # plain round-robin: 42 / 6
    assert subject_schedule(2).groups == [] and subject_schedule(2).ideal_speedup() == 2.0
    s4 = subject_schedule(4)
    assert s4.rounds == 10 and s4.groups == [(41, [0, 1]), (42, [2, 3])] and abs(s4.ideal_speedup() - 4.0) < 1e-12
    s5 = subject_schedule(5)                                           # 42 = 8 x 5 + 2: two pairs, rank 4 idles in the tail
    assert s5.groups == [(41, [0, 1]), (42, [2, 3])] and s5.group_of(4) is None


def test_subject_groups_gloo_world4(tmp_path):
    """The tail of the subject schedule on four gloo ranks: two 2-rank groups, each all-reducing ONLY inside itself; the
    gradient of a group whose members hold the two halves of a batch is BIT-equal to the undivided batch's (the shard
    means are pre-scaled by n_r / n and summed); whole-round subjects need no collective; one all_gather at the end."""
    script = tmp_path / "w4.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from eav_amd import dist as ed
        rank, world, local = ed.init_from_env("gloo")
        assert world == 4
        sched = ed.subject_schedule(world, 6)                 # 6 subjects: one whole round + subjects 5, 6 on two pairs
        assert sched.rounds == 1 and sched.groups == [(5, [0, 1]), (6, [2, 3])]
        groups = sched.make_groups()
        sub, ranks = sched.group_of(rank)
        # a linear model y = w . x with a quadratic loss: the gradient of the mean loss over a batch of 8 rows
        torch.manual_seed(100 + sub)                          # the members of a group see the same data and weights
        X, w, t = torch.randn(8, 16), torch.randn(16), torch.randn(8)
        full = (2.0 / 8) * ((X @ w - t)[:, None] * X).sum(0)
        i = ranks.index(rank)
        Xs, ts = X[4 * i:4 * i + 4], t[4 * i:4 * i + 4]
        g = (2.0 / 4) * ((Xs @ w - ts)[:, None] * Xs).sum(0)  # gradient of the mean over this rank's shard
        sync = ed.GradSync([g], group=groups[sub])
        assert sync.world == 2 and sync.weight == 0.5
        sync()
        # the other group ran a different subject at the same time: nothing of it may have leaked in
        assert torch.allclose(g, full, rtol=1e-6, atol=1e-6), (g - full).abs().max()
        halves = [(2.0 / 4) * ((X[4 * j:4 * j + 4] @ w - t[4 * j:4 * j + 4])[:, None] * X[4 * j:4 * j + 4]).sum(0) for j in (0, 1)]
        assert torch.equal(g, 0.5 * halves[0] + 0.5 * halves[1])      # bit-equal: scale by n_r / n, then SUM in rank order
        # results: every subject reported exactly once after one all_gather
        have = torch.zeros(6)
        for s_ in sched.solo[rank]:
            have[s_ - 1] = 1.0
        have[sub - 1] = 1.0
        out = [torch.empty_like(have) for _ in range(world)]
        dist.all_gather(out, have)
        assert bool((torch.stack(out).max(0).values == 1).all())
        dist.barrier()
        dist.destroy_process_group()
        open({str(tmp_path)!r} + f"/ok_{{rank}}", "w").write("ok")
    """))
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert all((tmp_path / f"ok_{{k}}".format(k=k)).exists() for k in range(4)), r.stdout + r.stderr


def test_grad_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from eav_amd import dist as ed
        rank, world, local = ed.init_from_env("gloo")
        assert world == 2
        g = torch.full((1000,), float(rank + 1))
        h = torch.arange(10, dtype=torch.float32) * (rank + 1)
        ed.GradSync([g, h])()
        assert torch.allclose(g, torch.full((1000,), 1.5)), g[:4]
        assert torch.allclose(h, torch.arange(10, dtype=torch.float32) * 1.5)
        # overlapped form: buckets submitted out of order during the "backward", the rest at sync time
        g2 = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        s = ed.GradSync([g2])
        s.bucket(600, 900)
        s.bucket(100, 350)
        s()
        assert torch.allclose(g2, torch.arange(1000, dtype=torch.float32) * 1.5), g2[:5]
        s.bucket(0, 1000)
        s()
        g3 = torch.full((100,), float(rank + 1))
        s3 = ed.GradSync([g3])
        s3.set_active([(10, 20), (50, 60)])
        s3()                                        # frozen phase: only the head slices are synchronised
        exp = torch.full((100,), float(rank + 1)); exp[10:20] = 1.5; exp[50:60] = 1.5
        assert torch.equal(g3, exp), g3
        assert torch.allclose(g2, torch.arange(1000, dtype=torch.float32) * 1.5)   # mean of identical replicas
        # a bucket submitted from inside the backward that lies ABOVE the active range must not be reduced a second
        # time, and nothing outside the active range may change at sync time
        g4 = torch.full((100,), float(rank + 1))
        s4 = ed.GradSync([g4])
        s4.set_active([(10, 20)])
        s4.bucket(60, 80)
        s4()
        exp = torch.full((100,), float(rank + 1)); exp[10:20] = 1.5; exp[60:80] = 1.5
        assert torch.equal(g4, exp), g4
        # uneven shards: rank 0 holds 3 of the 4 items of the global batch -> gradient = 3/4 g0 + 1/4 g1
        g5 = torch.full((8,), float(rank + 1))
        s5 = ed.GradSync([g5])
        s5.set_batch(3 if rank == 0 else 1, 4)
        s5()
        assert torch.allclose(g5, torch.full((8,), 0.75 * 1 + 0.25 * 2)), g5
        assert s5.bytes_reduced == 32
        assert ed.subjects_for_rank(rank, world)[0] == 1 + rank
        # attach(): any trainer whose model keeps a flat gradient buffer - here the alternative EEG encoders, whose flat
        # layouts contain declared zero padding (ShallowConvNet) - gets the all-reduce and leaves hipGraph replay
        from types import SimpleNamespace
        from eav_amd.cnn_eeg import EEGNet
        from eav_amd.transformer_eeg import ShallowConvNet
        torch.manual_seed(0)
        for model in (ShallowConvNet(5, num_layers=1), EEGNet(4)):
            tr = ed.attach(SimpleNamespace(model=model, grad_sync=None, use_graph=True))
            assert tr.use_graph is True and tr.grad_sync is not None      # graph replay survives data parallelism
            gflat = model._flat[1]
            gflat.fill_(float(rank + 1))
            tr.grad_sync()
            assert torch.allclose(gflat, torch.full_like(gflat, 1.5))
        dist.barrier()
        dist.destroy_process_group()
        open({str(tmp_path)!r} + f"/ok_{{rank}}", "w").write("ok")     # (stdout of the two ranks may interleave)
    """))
    import socket
    with socket.socket() as sock:          # a free rendezvous port (a fixed one can linger in TIME_WAIT)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists(), r.stdout + r.stderr


def test_forced_one_rank_group_gloo(tmp_path):
    """The world-1 forced mode tests/test_rccl_gpu.py relies on (there with backend nccl): init_from_env(force=True)
    creates a one-rank group, GradSync(force=True) really issues its collectives, results are unchanged."""
    script = tmp_path / "w1.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from eav_amd import dist as ed
        assert ed.backend_name() == "none"
        rank, world, local = ed.init_from_env("gloo", force=True)
        assert (rank, world) == (0, 1) and dist.is_initialized() and ed.backend_name() == "gloo"
        g = torch.arange(100, dtype=torch.float32)
        off = ed.GradSync([g])                       # default: a one-rank group needs no exchange
        off.bucket(0, 50); off()
        assert not off.enabled and off.collectives == 0 and off.bytes_reduced == 0
        s = ed.GradSync([g], force=True)
        s.bucket(10, 30)
        s()
        assert s.enabled and s.collectives == 3 and s.bytes_reduced == 400       # [10,30) + [0,10) + [30,100)
        assert torch.equal(g, torch.arange(100, dtype=torch.float32))
        dist.destroy_process_group()
        open({str(tmp_path)!r} + "/ok", "w").write("ok")
    """))
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok").exists()
