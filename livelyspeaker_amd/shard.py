"""Multi-GPU layer: batch-sharded replicas, one process per GPU.

The path shards perfectly -- no op couples two samples (InstanceNorm is per (sample, channel), LN per
token, token mixing within a sample, CFG and the sampler update elementwise) -- so there is NO per-step
collective.  The only communication is:
  * init:     broadcast of the weights from rank 0 (16.4 MB TED / 18 MB BEAT fp32) over RCCL/xGMI;
  * per call: optional broadcast/scatter of conditioning produced on one rank (e.g. the frozen CLIP text
              features of the LivelySpeaker config) and an all_gather of the [B/N, J, F, T] result.
The reference has no live distributed code to mirror (``mdm_utils/dist_util.py`` is stubbed, :18-41).
Noise in Philox mode is keyed by the GLOBAL sample index (``sample_offset``), so results do not depend
on how many GPUs the batch is split over.  Works with backend "nccl" (= RCCL) on GPUs and "gloo" on CPU.
"""
from __future__ import annotations

import sys

import torch
import torch.distributed as dist


#: process group the data collectives run on: None = the default group.  ``init_groups`` sets it to an RCCL group when one comes up
#: healthy and leaves the gloo default group (host copies) otherwise.
_GROUP = None


#: True once an RCCL probe thread has been abandoned (it may sit in a hung collective for good): ``finish`` then ends the process with
#: ``os._exit`` instead of waiting for interpreter shutdown to get past that thread and RCCL's own teardown.
_ABANDONED_PROBE = False


def _on() -> bool:
    return dist.is_available() and dist.is_initialized()


def backend() -> str:
    """Backend of the group the data collectives use ("nccl" = RCCL, "gloo", or "none" without a process group)."""
    return dist.get_backend(_GROUP) if _on() else "none"


def use_group(group) -> None:
    global _GROUP
    _GROUP = group


_ENV_NO_WATCHDOG = ("TORCH_NCCL_ASYNC_ERROR_HANDLING", "NCCL_ASYNC_ERROR_HANDLING")


def init_groups(device, rank: int, world: int, *, want_rccl: bool = True, probe_timeout_s: float = 120.0, timeout_min: float = 15.0) -> dict:
    """Bring up the process groups so that a broken RCCL cannot take the job down (the reference's ``dist_util.setup_dist`` is a
    stub, scripts/mdm_utils/dist_util.py:18-41, so there is nothing to mirror).  The DEFAULT group is gloo over TCP on the
    rendezvous the launcher gave us: it carries the control plane (barriers, agreement) and, if need be, the data on host copies.
    A THROW-AWAY RCCL group over the same ranks is then created and PROBED (one all_reduce + one all_gather on ``device``, waited for
    with a timeout in a helper thread).  Only if every rank's probe succeeded is the probe group destroyed and the DATA group created:
    a second RCCL group with the caller's ordinary timeout (``timeout_min``) and torch's default error handling, so a rank that dies
    mid-collective later is torn down by the watchdog as usual instead of leaving the others waiting.  Returns a record for the bench
    line: {collective_backend, rccl_ranks, rccl_error}.

    Process-wide side effect, limited to the probe: ``TORCH_NCCL_ASYNC_ERROR_HANDLING`` / ``NCCL_ASYNC_ERROR_HANDLING`` are set to "0"
    while the probe group is constructed (ProcessGroupNCCL reads them in its constructor) and put back to what they were -- unset
    included -- before the data group is created and before this function returns.  If the probe thread had to be ABANDONED inside a
    hung collective they stay at "0" (the abandoned group's watchdog must never end the process) and the run stays on gloo."""
    import os
    import threading
    from datetime import timedelta
    global _ABANDONED_PROBE
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(minutes=timeout_min))
    use_group(None)
    info = {"collective_backend": "gloo", "rccl_ranks": 0, "rccl_error": None}
    if not want_rccl:
        info["rccl_error"] = "not requested"
        return info
    ok, err = 1, None
    box = {}
    # A collective of the probe that hangs stays registered with ProcessGroupNCCL's watchdog; with torch's default error handling the
    # watchdog aborts the WHOLE process when it times out -- minutes after the run has moved on over gloo.  The probe's verdict is
    # taken here, by the join below, so the PROBE group's watchdog must not be able to end the process: no async error handling and a
    # group timeout far beyond the join's -- for that group only (the abandoned thread is dealt with by `finish`).
    saved_env = {k: os.environ.get(k) for k in _ENV_NO_WATCHDOG}
    for k in _ENV_NO_WATCHDOG:
        os.environ[k] = "0"

    def restore_env():
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def check_collectives(g):
        t = torch.full((1024,), float(rank + 1), device=device)
        dist.all_reduce(t, group=g)
        out = torch.empty(world * 8, device=device)
        dist.all_gather_into_tensor(out, torch.full((8,), float(rank), device=device), group=g)
        if getattr(device, "type", "cpu") == "cuda":
            torch.cuda.synchronize(device)
        want = world * (world + 1) / 2
        if float(t[0].item()) != want or [float(v) for v in out[::8].tolist()] != [float(r) for r in range(world)]:
            raise RuntimeError(f"RCCL probe returned wrong values ({float(t[0].item())} != {want})")

    def probe():
        try:
            if getattr(device, "type", "cpu") == "cuda":
                torch.cuda.set_device(device)       # the current device is per host thread: this helper thread starts on device 0
            g = dist.new_group(ranks=list(range(world)), backend="nccl", timeout=timedelta(seconds=max(3600.0, 20.0 * probe_timeout_s)))
            box["group"] = g
            check_collectives(g)
            box["ok"] = True
        except Exception as e:              # noqa: BLE001  (any failure means: stay on gloo)
            box["err"] = repr(e)[:300]

    th = threading.Thread(target=probe, daemon=True)
    th.start()
    th.join(probe_timeout_s)
    if th.is_alive():
        ok, err = 0, f"RCCL probe did not finish within {probe_timeout_s:.0f} s (probe thread abandoned; the process will leave through os._exit)"
        _ABANDONED_PROBE = True
        print(f"[shard] rank {rank}: {err}", file=sys.stderr, flush=True)
    elif not box.get("ok"):
        ok, err = 0, box.get("err", "RCCL probe failed")
    if not _ABANDONED_PROBE:
        restore_env()                       # (an abandoned probe group is still alive: its watchdog must stay harmless)
    agree = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)            # over gloo: every rank takes the same decision
    probe_up = bool(box.get("ok")) and not th.is_alive()
    if probe_up:                                            # the probe group has served its purpose either way: give the communicator back
        try:
            dist.destroy_process_group(box["group"])
        except Exception as e:                              # noqa: BLE001  (teardown of RCCL must not take the run down either)
            err = (err or "") + f"; destroy_process_group(probe): {e!r}"[:120]
    if int(agree.item()) == 1:
        # the data group: ordinary timeout, default error handling (the env is back to the caller's), checked once before use
        data_ok, data_err, group = 1, None, None
        try:
            group = dist.new_group(ranks=list(range(world)), backend="nccl", timeout=timedelta(minutes=timeout_min))
            check_collectives(group)
        except Exception as e:                              # noqa: BLE001
            data_ok, data_err = 0, f"RCCL data group failed after a good probe: {e!r}"[:300]
        agree2 = torch.tensor([data_ok], dtype=torch.int32)
        dist.all_reduce(agree2, op=dist.ReduceOp.MIN)
        if int(agree2.item()) == 1:
            use_group(group)
            info.update(collective_backend="nccl", rccl_ranks=world)
        else:
            info["rccl_error"] = data_err or "RCCL data group failed on another rank"
            if group is not None and data_ok:
                try:
                    dist.destroy_process_group(group)
                except Exception as e:                      # noqa: BLE001
                    info["rccl_error"] += f"; destroy_process_group: {e!r}"[:120]
    else:
        info["rccl_error"] = err or "RCCL probe failed on another rank"
    info["rccl_probe_abandoned"] = _ABANDONED_PROBE
    return info


def finish(code: int = 0) -> None:
    """Last call of a multi-rank program.  If an RCCL probe thread was abandoned, interpreter shutdown would wait on RCCL's teardown
    of a communicator that never finished coming up: flush what was printed and leave through ``os._exit`` -- deterministically, and
    said so on stderr (``os._exit`` runs no atexit handlers: stdout, stderr and the root logger's handlers are flushed here, files the
    caller opened are the caller's to flush and close first).  Otherwise: the ordinary ``destroy_process_group``."""
    import os
    if _ABANDONED_PROBE:
        print(f"[shard] leaving through os._exit({code}): an RCCL probe thread is still inside a collective", file=sys.stderr, flush=True)
        import logging
        for hd in list(logging.getLogger().handlers):       # os._exit skips atexit handlers and buffered writers other than these:
            try:                                            # callers flush and close their OWN files before calling finish()
                hd.flush()
            except Exception:                               # noqa: BLE001
                pass
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)
    if _on():
        dist.destroy_process_group()


def _stage(t: torch.Tensor) -> torch.Tensor:
    """The tensor a collective runs on: device memory under RCCL, a host copy under gloo (whose device-tensor support is partial:
    used by the CPU tests, by the two-ranks-on-one-GPU bench test, where RCCL refuses the duplicate device, and whenever the RCCL
    probe of ``init_groups`` failed)."""
    return t.cpu() if (t.is_cuda and backend() == "gloo") else t


def all_reduce_(t: torch.Tensor, op=None) -> torch.Tensor:
    """In-place all_reduce of ``t`` under either backend (no-op without a process group)."""
    if not _on():
        return t
    op = dist.ReduceOp.SUM if op is None else op
    s = _stage(t)
    dist.all_reduce(s, op=op, group=_GROUP)
    if s is not t:
        t.copy_(s)
    return t


def shard_range(total: int, world: int, rank: int):
    """Contiguous split of ``total`` samples: rank r owns [first, first+count)."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def shard_cond(y: dict, world: int, rank: int) -> dict:
    """Slice every batched tensor of the conditioning dict to this rank's samples."""
    total = next(v.shape[0] for v in y.values() if torch.is_tensor(v))
    first, count = shard_range(total, world, rank)
    return {k: (v[first:first + count].clone() if torch.is_tensor(v) and v.shape[:1] == (total,) else v)
            for k, v in y.items()}


def broadcast_state_dict(sd: dict, device, src: int = 0) -> dict:
    """Make every rank hold rank ``src``'s weights (one bucketed broadcast per dtype)."""
    if not (dist.is_available() and dist.is_initialized()):
        return sd
    keys = sorted(sd.keys())
    flat = _stage(torch.cat([sd[k].reshape(-1).to(torch.float32) for k in keys]).to(device))
    dist.broadcast(flat, src=src, group=_GROUP)
    flat = flat.cpu()
    out, off = {}, 0
    for k in keys:
        n = sd[k].numel()
        out[k] = flat[off:off + n].reshape(sd[k].shape).clone()
        off += n
    return out


def broadcast_tensor(t: torch.Tensor, device, src: int = 0) -> torch.Tensor:
    """Broadcast conditioning computed on one rank (e.g. CLIP text features [B, 512]) to all ranks."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    buf = t.to(device).contiguous()
    s = _stage(buf)
    dist.broadcast(s, src=src, group=_GROUP)
    if s is not buf:
        buf.copy_(s)
    return buf


def gather_samples(local: torch.Tensor, total: int) -> torch.Tensor:
    """all_gather the per-rank [count_r, J, F, T] results into the global [total, J, F, T] tensor.  Equal shards (the
    BASELINE configs: 4096 = 8 x 512) go through ONE ``all_gather_into_tensor`` straight into the output buffer; ragged
    shards are padded to the largest shard first and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    counts = [shard_range(total, world, r)[1] for r in range(world)]
    pad = max(counts)
    local = local.contiguous()
    if local.shape[0] != pad:
        buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
        buf[: local.shape[0]] = local
        local = buf
    out = local.new_empty((world * pad,) + tuple(local.shape[1:]))
    if local.is_cuda and backend() == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu(), group=_GROUP)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, local, group=_GROUP)
    if min(counts) == pad:
        return out
    return torch.cat([out[r * pad: r * pad + c] for r, c in enumerate(counts)], dim=0)


def rank_reports(mine: dict) -> list:
    """Every rank's small diagnostic record (device, compute units, kernel plan, its own wall time ...) collected on every rank, in rank
    order -- over the gloo control-plane group, so it works whatever became of RCCL.  One rank: ``[mine]``.  Collective."""
    if not _on():
        return [dict(mine, rank=0)]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, dict(mine, rank=dist.get_rank()))
    return out


def timed_gather(local: torch.Tensor, total: int, reps: int = 3) -> dict:
    """Wall time of the result gather alone (barrier -> ``gather_samples`` -> device synchronised), best of ``reps``, MAX over ranks: the
    only collective of a sampling call, so the first multi-GPU line says what it costs.  Collective."""
    import time
    best = float("inf")
    for _ in range(reps):
        if _on():
            dist.barrier()
        if local.is_cuda:
            torch.cuda.synchronize(local.device)
        t0 = time.perf_counter()
        whole = gather_samples(local, total)
        if whole.is_cuda:
            torch.cuda.synchronize(whole.device)
        best = min(best, time.perf_counter() - t0)
    t = torch.tensor([best], dtype=torch.float64, device=local.device)
    if _on():
        all_reduce_(t, dist.ReduceOp.MAX)
    return {"ms": round(float(t.item()) * 1e3, 3), "bytes_per_rank": int(local.numel() * local.element_size()), "total_samples": int(total),
            "collective_backend": backend(), "what": "all_gather of the per-rank results into the global batch (gather_samples), max over ranks"}


def sample_sharded(sample_fn, model, global_shape, y_global: dict, diffusion=None, **kwargs) -> torch.Tensor:
    """Run ``sample_fn`` (``diffusion.p_sample_loop`` / ``ddim_sample_loop``) on this rank's contiguous
    shard of the batch and return the gathered global result.  ``diffusion.sample_offset`` is set so the
    Philox streams follow the global sample index."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    total = int(global_shape[0])
    first, count = shard_range(total, world, rank)
    y_local = shard_cond(y_global, world, rank)
    if diffusion is not None:
        diffusion.sample_offset = first
    for name in ("init_image", "noise"):
        if kwargs.get(name) is not None:
            kwargs[name] = kwargs[name][first:first + count]
    local = sample_fn(model, (count,) + tuple(global_shape[1:]), model_kwargs={"y": y_local}, **kwargs)
    return gather_samples(local, total)


def cross_check(generate, mine: torch.Tensor, total: int, *, equal_shards_of: int | None = None) -> dict:
    """The "N GPUs == one GPU" premise checked instead of assumed: every rank re-generates the NEXT rank's shard by itself --
    ``generate(first, count, shard_index)`` must produce the samples of global indices [first, first + count) from scratch on this
    rank (that shard's conditioning, the same Philox key, ``sample_offset = first``) -- and compares it with what that rank produced
    (``mine`` = this rank's own shard, gathered over the group).  Returns the max-abs difference (all-reduced MAX) and both checksums
    (all-reduced SUM); identical streams give exactly 0.  Works on one rank too (a replay check).  Collective: every rank must call
    it.  ``equal_shards_of``: per-rank shard size when shards are defined by rank (weak scaling) rather than by ``shard_range``."""
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if on else 1
    rank = dist.get_rank() if on else 0
    nb = (rank + 1) % world
    if equal_shards_of is not None:
        if total != world * equal_shards_of:      # gather_samples lays the shards out by shard_range(total, ...): the two must agree
            raise ValueError(f"cross_check: total {total} != world {world} x equal_shards_of {equal_shards_of}")
        spans = [(r * equal_shards_of, equal_shards_of) for r in range(world)]
    else:
        spans = [shard_range(total, world, r) for r in range(world)]
    whole = gather_samples(mine, total)
    first, count = spans[nb]
    if count > 0:
        redo = generate(first, count, nb)
        theirs = whole[first:first + count]
        stats = torch.stack([(redo - theirs).abs().max().double(), redo.double().abs().sum(), theirs.double().abs().sum()])
    else:                                         # total < world: the neighbour's shard is empty and contributes nothing
        stats = torch.zeros(3, dtype=torch.float64, device=mine.device)
    mx, sums = stats[:1].clone(), stats[1:].clone()
    all_reduce_(mx, dist.ReduceOp.MAX)
    all_reduce_(sums, dist.ReduceOp.SUM)
    be = backend()
    return {"rccl_ranks": world if be in ("nccl", "none") else 0, "ranks": world, "collective_backend": be,
            "what": "every rank re-generated the NEXT rank's shard on its own GPU via sample_offset (same Philox key, that shard's "
                    "conditioning) and compared it with what that rank produced",
            "max_abs_diff": float(mx.item()), "bitwise_equal": bool(mx.item() == 0.0),
            "checksum_recomputed": float(sums[0].item()), "checksum_sharded": float(sums[1].item())}
