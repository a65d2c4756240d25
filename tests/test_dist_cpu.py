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


def test_replica_shard_gives_every_group_member_the_same_step_count():
    """eav_amd.dist.replica_shard (tools/run_*_subjects.py): equal shard lengths and batch sizes on every member - members
    that disagree on the number of optimiser steps dead-lock in the gradient all-reduce (200 trials on 41 ranks gave 5
    against 4 steps with plain i::n slicing) - and a group that does not divide the batch is refused, not silently run
    with a different global batch."""
    import numpy as np
    import pytest
    from eav_amd.dist import replica_shard
    items = np.arange(200)
    for n, bs in ((1, 32), (2, 32), (4, 32), (8, 32), (16, 32), (32, 32), (4, 8), (4, 128)):
        shards = [items[replica_shard(200, bs, i, n)[0]] for i in range(n)]
        sizes = {len(s_) for s_ in shards}
        assert len(sizes) == 1 and sizes.pop() == 200 // n
        assert len(np.unique(np.concatenate(shards))) == (200 // n) * n          # disjoint
        per = {replica_shard(200, bs, i, n)[1] for i in range(n)}
        assert per == {bs // n}
        steps = {-(-len(s_) // (bs // n)) for s_ in shards}
        assert len(steps) == 1
    assert replica_shard(200, 32, 0, 1) == (slice(0, 200), 32)
    for n, bs in ((41, 32), (3, 32), (64, 32)):
        with pytest.raises(ValueError):
            replica_shard(200, bs, 0, n)
    with pytest.raises(ValueError):
        replica_shard(200, 32, 4, 4)


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


def test_subject_schedule_gloo_world8_two_groups_of_four(tmp_path):
    """The EXACT topology of the 8-GPU job (BASELINE configs[4]): subject_schedule(8, 42) = five whole rounds + subjects 41
    and 42 on two 4-RANK groups.  Eight gloo ranks: inside a group GradSync.weight = 1/4 and shard_batch(64, i, 4) = 16 rows
    per member; every element of the group gradient is BIT-equal to a sum of the four pre-scaled shard gradients in one of the
    ring's orders, identical on all four members, and equal (fp32
    rounding) to the undivided batch's; the two groups reduce at the same time and nothing of group B reaches group A (their
    data differ by seed, and a poisoned buffer reduced inside B must not show up in A); every one of the 42 subjects is
    reported exactly once after one all_gather; an uneven LAST batch (37 rows: 10/10/10/7) is weighted n_r / n.  A rank that fails exits non-zero and torchrun tears the job down (second launch below)."""
    script = tmp_path / "w8.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from eav_amd import dist as ed
        rank, world, local = ed.init_from_env("gloo")
        assert world == 8
        if os.environ.get("EAV_TEST_FAIL_RANK") == str(rank):
            raise SystemExit(3)                                # (d): a failing rank must end the whole job non-zero
        sched = ed.subject_schedule(world, 42)
        assert sched.rounds == 5 and sched.group_size == 4
        assert sched.groups == [(41, [0, 1, 2, 3]), (42, [4, 5, 6, 7])]
        assert sched.solo[rank] == [1 + rank + 8 * k for k in range(5)]
        groups = sched.make_groups()
        assert sorted(groups) == [41, 42]
        sub, ranks = sched.group_of(rank)
        assert sub == (41 if rank < 4 else 42) and rank in ranks
        i = ranks.index(rank)
        B = 64
        torch.manual_seed(1000 + sub)                          # members of a group: same data, same weights
        X, w, t = torch.randn(B, 24), torch.randn(24), torch.randn(B)
        grad = lambda Xs, ts: (2.0 / len(ts)) * ((Xs @ w - ts)[:, None] * Xs).sum(0)   # gradient of the MEAN loss over the rows
        lo, hi = ed.shard_batch(B, i, 4)
        assert (lo, hi) == (16 * i, 16 * i + 16)
        g = grad(X[lo:hi], t[lo:hi])
        sync = ed.GradSync([g], group=groups[sub])
        assert sync.world == 4 and sync.weight == 0.25 and sync.enabled
        sync()
        parts = [grad(X[16 * j:16 * j + 16], t[16 * j:16 * j + 16]) for j in range(4)]
        ref = ((0.25 * parts[0] + 0.25 * parts[1]) + 0.25 * parts[2]) + 0.25 * parts[3]
        full = grad(X, t)
        assert torch.allclose(g, full, rtol=2e-6, atol=2e-6), (g - full).abs().max()
        # all four members hold the same bits (an all-reduce result is identical on every member)
        same = [torch.empty_like(g) for _ in range(4)]
        dist.all_gather(same, g, group=groups[sub])
        assert all(torch.equal(same[0], s_) for s_ in same)
        # ... and those bits are a fixed-order sum of the pre-scaled shard gradients (gloo: ring order may differ from the
        # left-to-right one, so compare against every association of the four terms)
        import itertools
        sc = [0.25 * p_ for p_ in parts]
        cands = []
        for perm in itertools.permutations(range(4)):
            a, b, c, d = (sc[k] for k in perm)
            cands += [((a + b) + c) + d, (a + b) + (c + d)]
        # (a ring all-reduce sums each CHUNK of the buffer in its own rotation of the ring: element-wise membership)
        assert bool(torch.stack([g == c_ for c_ in cands]).any(0).all()), (g - ref).abs().max()
        # isolation: group B reduces a poisoned buffer while group A reduces zeros - A must still see zeros
        probe = torch.full((8,), 1e30 if sub == 42 else 0.0)
        ed.GradSync([probe], group=groups[sub])()
        assert float(probe.max()) == (float(torch.tensor(1e30)) if sub == 42 else 0.0) and float(probe.min()) == float(probe.max())
        # ragged last batch of 37 rows over 4 members: 10 / 10 / 10 / 7 rows, weights n_r / n
        lo, hi = ed.shard_batch(37, i, 4)
        n_r = hi - lo
        assert n_r == (10, 10, 10, 7)[i]
        g2 = grad(X[lo:hi], t[lo:hi])
        s2 = ed.GradSync([g2], group=groups[sub])
        s2.set_batch(n_r, 37)
        s2()
        assert torch.allclose(g2, grad(X[:37], t[:37]), rtol=2e-6, atol=2e-6)
        # results of the whole job: 42 subjects, each reported by exactly one rank (a group reports through its first member)
        have = torch.zeros(42)
        for s_ in sched.solo[rank]:
            have[s_ - 1] += 1.0
        if ranks[0] == rank:
            have[sub - 1] += 1.0
        out = [torch.empty_like(have) for _ in range(world)]
        dist.all_gather(out, have)
        assert torch.equal(torch.stack(out).sum(0), torch.ones(42))
        assert abs(sched.ideal_speedup() - 8.0) < 1e-12
        dist.barrier()
        dist.destroy_process_group()
        open({str(tmp_path)!r} + f"/ok_{{rank}}", "w").write("ok")
    """))
    import socket

    def run(extra_env):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", **extra_env)
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                              capture_output=True, text=True, env=env, timeout=600)

    r = run({})
    assert r.returncode == 0, r.stdout + r.stderr
    assert all((tmp_path / f"ok_{k}").exists() for k in range(8)), r.stdout + r.stderr
    for k in range(8):
        (tmp_path / f"ok_{k}").unlink()
    r = run({"EAV_TEST_FAIL_RANK": "5"})          # one rank dies before the first collective: the job must not hang or pass
    assert r.returncode != 0
    assert not any((tmp_path / f"ok_{k}").exists() for k in range(8))


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
