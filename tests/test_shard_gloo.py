"""CPU, world_size 2 over gloo: the multi-GPU shard layer (livelyspeaker_amd/shard.py).  The engine itself
needs a GPU, so the per-rank sampler here is the CPU oracle (tests may use it); what is under test is the
partitioning, the weight/conditioning broadcasts, the ragged gather, and shard-invariance of the result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_sample_fn(total, tape):
    """A p_sample_loop-shaped callable backed by the oracle; noise comes from a GLOBAL tape indexed by the
    global sample index (what Philox's sample_offset gives the real engine)."""
    from livelyspeaker_amd import synth
    from oracle import rag_oracle as orc
    cfg = synth.TED
    oracle = orc.RagOracle(synth.make_state_dict(cfg), cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens)
    sch = orc.Schedule(3, "")

    def fn(model, shape, model_kwargs=None, **kw):
        first = model["diffusion"].sample_offset
        n = shape[0]
        y = {k: v.numpy() for k, v in model_kwargs["y"].items() if torch.is_tensor(v)}
        sl = slice(first, first + n)
        out = orc.sample_loop(oracle, sch, y, tape.x_init[sl], tape.eps[:, :, sl], tape.noise[:, sl])
        return torch.from_numpy(out)
    return fn


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from livelyspeaker_amd import shard, synth
        cfg = synth.TED
        # 1. weights: rank 0's copy wins
        sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=7 + rank).items()}
        sd = shard.broadcast_state_dict(sd, torch.device("cpu"))
        ref = synth.make_state_dict(cfg, seed=7)
        assert all(np.array_equal(sd[k].numpy(), ref[k]) for k in ref)
        # 2. conditioning produced on rank 0 only (e.g. frozen CLIP text features) reaches everyone
        feat = torch.arange(total * 512, dtype=torch.float32).reshape(total, 512) if rank == 0 else torch.zeros(total, 512)
        feat = shard.broadcast_tensor(feat, torch.device("cpu"))
        assert float(feat[-1, -1]) == total * 512 - 1
        # 3. partition: contiguous, ragged, complete
        first, count = shard.shard_range(total, world, rank)
        spans = [shard.shard_range(total, world, r) for r in range(world)]
        assert sum(c for _, c in spans) == total and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        # 4. sharded sampling + ragged all_gather == unsharded result
        y = {k: torch.from_numpy(v) for k, v in synth.make_cond(cfg, total).items()}
        y["text"] = ["ignored"] * total               # non-tensor entries pass through
        tape = synth.NoiseTape(cfg, total, 3)
        diffusion = SimpleNamespace(sample_offset=0)
        fn = _oracle_sample_fn(total, tape)
        got = shard.sample_sharded(fn, {"diffusion": diffusion}, (total, 9, 3, 34), y, diffusion=diffusion)
        assert diffusion.sample_offset == first and got.shape == (total, 9, 3, 34)
        q.put((rank, got.numpy()))
    except Exception as e:                      # never leave the parent blocked on the queue
        import traceback
        q.put((rank, RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [5])
def test_world2_sharded_equals_unsharded(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for v in results.values():
        if isinstance(v, Exception):
            raise v
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(results[0], results[1])
    # single-process, unsharded reference
    from types import SimpleNamespace
    from livelyspeaker_amd import synth
    cfg = synth.TED
    tape = synth.NoiseTape(cfg, total, 3)
    y = {k: torch.from_numpy(v) for k, v in synth.make_cond(cfg, total).items()}
    diffusion = SimpleNamespace(sample_offset=0)
    whole = _oracle_sample_fn(total, tape)({"diffusion": diffusion}, (total, 9, 3, 34), model_kwargs={"y": y})
    assert np.abs(results[0] - whole.numpy()).max() < 2e-5      # BLAS blocking differs with the batch size


def test_shard_range_properties():
    from livelyspeaker_amd import shard
    for total in (1, 7, 512, 4096):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


# ---- data-parallel training step: the only real collective of the build ------------------------
def _train_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from livelyspeaker_amd import synth
        from livelyspeaker_amd.train_loop import allreduce_mean_
        from oracle import train_oracle as tro
        cfg = synth.TED
        total = 4
        per = total // world
        sd = synth.make_state_dict(cfg)
        t_all = np.array([0, 999, 321, 77])
        x, y, noise, drop, eps = synth.make_train_batch(cfg, per, 0, first_sample=rank * per, total=total)
        oracle = tro.TrainOracle(sd, cfg.n_prefix_tokens)
        _, loss, grads, _ = oracle.forward_backward(x, t_all[rank * per:(rank + 1) * per], noise, y, drop, eps)
        keys = sorted(grads)
        flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in keys]))
        allreduce_mean_(flat)                                   # what TrainLoop.forward_backward does with trainer.grad
        if rank == 0:
            xf, yf, nf, df, ef = synth.make_train_batch(cfg, total, 0)
            _, _, gfull, _ = tro.TrainOracle(sd, cfg.n_prefix_tokens).forward_backward(xf, t_all, nf, yf, df, ef)
            want = np.concatenate([gfull[k].ravel() for k in keys])
            err = float(np.abs(flat.numpy() - want).max() / np.abs(want).max())
            q.put(("ok", err))
        dist.barrier()
    except Exception as e:                                       # surface the failure instead of hanging the parent
        q.put(("error", repr(e)))
    finally:
        dist.destroy_process_group()


def test_world2_data_parallel_gradient_equals_full_batch_gradient():
    """Mean of the per-shard gradients (all-reduce / world) == gradient of the full batch: every loss term is a mean
    over the batch, shards are equal-sized."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, val = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
    assert status == "ok", val
    assert val < 1e-5, val


# ---- bench.py's shard cross-check (shard.cross_check), world 2 over gloo with a stand-in generator ------------------------------
def _xcheck_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from livelyspeaker_amd import shard
        B = 3

        def gen(first, count, shard_index):          # what a Philox-keyed sampler is: a pure function of the global sample index
            idx = torch.arange(first, first + count, dtype=torch.float32)
            return (torch.sin(idx * 12.9898)[:, None, None, None] * torch.ones(count, 2, 3, 4) + shard_index * 0.0).contiguous()

        mine = gen(rank * B, B, rank)
        ok = shard.cross_check(gen, mine, world * B, equal_shards_of=B)
        bad_mine = mine + (1e-3 if rank == 1 else 0.0)                  # rank 1 "computes something else"
        bad = shard.cross_check(gen, bad_mine, world * B, equal_shards_of=B)
        ragged = shard.cross_check(lambda f, c, s: gen(f, c, s), gen(*shard.shard_range(5, world, rank), rank), 5)      # 3 + 2 samples
        # total < world (shard_range allows it): rank 1 owns nothing; the check must not trip over the empty shard
        tiny = shard.cross_check(gen, gen(*shard.shard_range(1, world, rank), rank), 1)
        try:
            shard.cross_check(gen, mine, world * B + 1, equal_shards_of=B)
            mismatch = "accepted"
        except ValueError as e:
            mismatch = str(e)
        q.put((rank, (ok, bad, ragged, tiny, mismatch)))
    except Exception as e:
        import traceback
        q.put((rank, RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")))
    finally:
        dist.destroy_process_group()


def test_world2_shard_cross_check():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xcheck_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for v in results.values():
        if isinstance(v, Exception):
            raise v
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        ok, bad, ragged, tiny, mismatch = results[rank]
        assert ok["ranks"] == 2 and ok["collective_backend"] == "gloo" and ok["rccl_ranks"] == 0     # not RCCL: says so
        assert ok["bitwise_equal"] and ok["max_abs_diff"] == 0.0
        assert tiny["bitwise_equal"] and "equal_shards_of" in mismatch
        assert ok["checksum_recomputed"] == ok["checksum_sharded"]
        assert not bad["bitwise_equal"] and abs(bad["max_abs_diff"] - 1e-3) < 1e-6          # seen by EVERY rank (all-reduced)
        assert ragged["bitwise_equal"]
    assert results[0][0] == results[1][0]


# ---- shard.init_groups: RCCL is tried, probed, and abandoned for gloo when it does not come up (here: no GPU at all) ---------------
def _groups_worker(rank, world, port, want_rccl, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        from livelyspeaker_amd import shard
        os.environ.pop("TORCH_NCCL_ASYNC_ERROR_HANDLING", None)
        os.environ["NCCL_ASYNC_ERROR_HANDLING"] = "1"                    # the caller's own setting: must survive the probe
        info = shard.init_groups(torch.device("cpu"), rank, world, want_rccl=want_rccl, probe_timeout_s=60.0, timeout_min=2.0)
        # the probe's "no watchdog" environment is scoped to the throw-away probe group: afterwards the process env is the caller's again
        assert "TORCH_NCCL_ASYNC_ERROR_HANDLING" not in os.environ and os.environ["NCCL_ASYNC_ERROR_HANDLING"] == "1", dict(os.environ)
        t = torch.full((4,), float(rank + 1))
        shard.all_reduce_(t)                                             # the data collectives work on whatever group was chosen
        whole = shard.gather_samples(torch.full((2, 1, 1, 1), float(rank)), 2 * world)
        q.put((rank, (info, t.tolist(), whole.flatten().tolist(), shard.backend())))
    except Exception as e:
        import traceback
        q.put((rank, RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("want_rccl", [False, True])
def test_world2_process_groups_fall_back_to_gloo_when_rccl_does_not_come_up(want_rccl):
    """bench.py's bring-up on a box without GPUs: the RCCL group cannot be created (or its probe collectives fail), every rank agrees
    on that over the gloo default group, and the run carries on over gloo, reporting collective_backend / rccl_ranks / rccl_error."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_groups_worker, args=(r, 2, port, want_rccl, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(60)
    for r in range(2):
        assert not isinstance(res[r], Exception), res[r]
        info, red, whole, be = res[r]
        assert info["collective_backend"] == "gloo" and info["rccl_ranks"] == 0 and info["rccl_error"] and be == "gloo", info
        assert red == [3.0] * 4 and whole == [0.0, 0.0, 1.0, 1.0]


def test_world4_bring_up_and_gather_order():
    """The same bring-up with four ranks (the driver's N = 4 / 8 runs are the first with more than two): every rank reaches the same
    verdict, the reduction and the rank-ordered gather see all of them."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_groups_worker, args=(r, world, port, True, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    for r in range(world):
        assert not isinstance(res[r], Exception), res[r]
        info, red, whole, be = res[r]
        assert info["collective_backend"] == "gloo" and info["rccl_ranks"] == 0 and be == "gloo", info
        assert red == [10.0] * 4 and whole == [float(k) for k in range(world) for _ in range(2)]


# ---- eight ranks (the driver's N = 8 run is the first time eight launchers meet): bring-up verdict, reduction, rank-ordered gather
# with a RAGGED total, and the shard cross-check, all over gloo ---------------------------------------------------------------------
def _world8_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        from livelyspeaker_amd import shard
        info = shard.init_groups(torch.device("cpu"), rank, world, want_rccl=True, probe_timeout_s=60.0, timeout_min=3.0)
        t = torch.full((4,), float(rank + 1))
        shard.all_reduce_(t)
        total = 3 * world + 5                                             # ragged: the first five ranks own one sample more
        first, count = shard.shard_range(total, world, rank)

        def gen(f, c, shard_index):
            idx = torch.arange(f, f + c, dtype=torch.float32)
            return (torch.cos(idx * 78.233)[:, None, None, None] * torch.ones(c, 2, 1, 3)).contiguous()

        mine = gen(first, count, rank)
        whole = shard.gather_samples(mine, total)
        chk = shard.cross_check(gen, mine, total)
        # what bench.py --gpus 8 adds to its line so that the first run on eight real GPUs is diagnosable from it alone
        reps = shard.rank_reports({"device": f"cpu:{rank}", "n_cus": 256 - rank, "collective_backend": shard.backend(), "elapsed_s": 0.5 + rank})
        tg = shard.timed_gather(mine, total, reps=2)
        q.put((rank, (info, t.tolist(), whole[:, 0, 0, 0].tolist(), (first, count), chk, reps, tg)))
        shard.finish(0)
    except Exception as e:
        import traceback
        q.put((rank, RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")))


def test_world8_bring_up_ragged_gather_and_cross_check():
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=400) for _ in ps)
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    total = 3 * world + 5
    want = torch.cos(torch.arange(total, dtype=torch.float32) * 78.233).tolist()
    covered = []
    for r in range(world):
        assert not isinstance(res[r], Exception), res[r]
        info, red, whole, (first, count), chk, reps, tg = res[r]
        assert info["collective_backend"] == "gloo" and info["rccl_ranks"] == 0 and not info["rccl_probe_abandoned"], info
        # the N = 8 line's diagnostics: one record per rank, in rank order, each with its device's CU count and the backend it ended up on;
        # the gather's own time, byte count and backend
        assert [x["rank"] for x in reps] == list(range(8)) and [x["n_cus"] for x in reps] == [256 - k for k in range(8)], reps
        assert all(x["collective_backend"] == "gloo" and x["device"] == f"cpu:{x['rank']}" for x in reps)
        assert tg["ms"] > 0 and tg["collective_backend"] == "gloo" and tg["total_samples"] == total and tg["bytes_per_rank"] == count * 6 * 4, tg
        assert red == [36.0] * 4 and whole == want                         # every rank holds the whole batch in global order
        assert chk["ranks"] == 8 and chk["bitwise_equal"] and chk["checksum_recomputed"] == chk["checksum_sharded"], chk
        covered += list(range(first, first + count))
    assert covered == list(range(total))


# ---- an RCCL probe that HANGS: the run goes on over gloo, says so, and the process leaves through os._exit (exit code kept) --------
def _hung_probe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    real_new_group = dist.new_group

    def hanging_new_group(*a, **k):
        if k.get("backend") == "nccl":
            time.sleep(3600)                                               # a communicator that never comes up
        return real_new_group(*a, **k)

    dist.new_group = hanging_new_group
    try:
        from livelyspeaker_amd import shard
        info = shard.init_groups(torch.device("cpu"), rank, world, want_rccl=True, probe_timeout_s=3.0, timeout_min=2.0)
        t = torch.full((2,), float(rank + 1))
        shard.all_reduce_(t)
        q.put((rank, (info, t.tolist(), os.environ.get("TORCH_NCCL_ASYNC_ERROR_HANDLING"))))
        dist.barrier()
        shard.finish(7)                                                    # must not wait for the sleeping probe thread
        q.put((rank, "finish returned"))
    except Exception as e:
        import traceback
        q.put((rank, RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")))


def test_a_hung_rccl_probe_is_abandoned_and_the_process_exits_deterministically():
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_hung_probe_worker, args=(r, 2, port, q)) for r in range(2)]
    t0 = time.time()
    for p in ps:
        p.start()
    res = dict(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(120)
        assert p.exitcode == 7                                             # os._exit(7): not killed by a watchdog, not stuck in teardown
    assert time.time() - t0 < 200
    for r in range(2):
        assert not isinstance(res[r], Exception), res[r]
        info, red, env = res[r]
        assert info["collective_backend"] == "gloo" and info["rccl_probe_abandoned"] and "abandoned" in info["rccl_error"], info
        assert red == [3.0, 3.0] and env == "0"
    assert q.empty()                                                       # nobody got past finish()
