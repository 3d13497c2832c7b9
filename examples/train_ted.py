#!/usr/bin/env python3
"""A few optimisation steps of the RAG denoiser with the TrainLoop drop-in (scripts/train_RAG.py:37-43 of the reference) on
SYNTHETIC batches, then sampling with the trained weights.  Single GPU:

    python examples/train_ted.py [steps] [batch]
Data parallel over the GPUs of one node (one process per GPU, RCCL gradient all-reduce):
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ted.py 20 512
"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livelyspeaker_amd import synth                                                   # noqa: E402
from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel                   # noqa: E402
from livelyspeaker_amd.model_util import create_model_and_diffusion                   # noqa: E402
from livelyspeaker_amd.train_loop import TrainLoop                                    # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rank, world, local = 0, 1, 0
    if "RANK" in os.environ:
        rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dev = f"cuda:{local}"
    cfg = synth.TED
    margs = SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                            emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu", diffusion_steps=1000,
                            noise_schedule="cosine", sigma_small=True, lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0, njoints=9)
    model, diffusion = create_model_and_diffusion(margs, '')                                 # train_RAG.py:37
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}, strict=False)
    model.to(dev)
    model.train()

    def batches():                              # stands in for the LMDB loader: each rank sees its own shard of every batch
        for s in range(steps):
            x_start, y, _, _, _ = synth.make_train_batch(cfg, B, s % 4, first_sample=rank * B, total=world * B)
            # already on the device, as a pinned-memory DataLoader with non_blocking copies would deliver them
            yield torch.from_numpy(x_start).to(dev), {"y": {k: torch.from_numpy(v).to(dev) for k, v in y.items()}}

    targs = SimpleNamespace(batch_size=B, lr=1e-4, weight_decay=0.0, lr_anneal_steps=0, log_interval=1, save_interval=10 ** 9,
                            resume_checkpoint="", epochs=1, save_dir="/tmp/ls_train_example", overwrite=True, dataset="ted")
    np.random.seed(1 + rank)
    torch.manual_seed(1 + rank)
    data = list(batches())
    loop = TrainLoop(targs, None, model, diffusion, data)
    loop.noise_device = "cuda"                   # draw noise / dropout / eps on the GPU (the default "cpu" replays the reference's CPU stream)
    loop.run_step(*data[0])                       # untimed: the first step allocates the activation buffers (~4.5 GB at B=512)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.run_loop()                                                                          # train_RAG.py:43
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"{steps} steps x {world} x {B} samples in {dt:.3f} s = {dt / steps * 1e3:.1f} ms/step (batches resident on the device); "
              f"last loss {loop.last_losses['total']:.4f}")
        # the trained weights are back in `model` (run_loop ends with sync_model()): sample with them
        model.eval()
        y = synth.make_cond(cfg, 8)
        cond = {"y": {k: torch.from_numpy(v).to(dev) for k, v in y.items()}}
        _, diff_s = create_model_and_diffusion(margs, 'ddim100')
        out = diff_s.ddim_sample_loop(ClassifierFreeSampleModel(model), (8, 9, 3, 34), clip_denoised=False, model_kwargs=cond,
                                      skip_timesteps=0, init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
        print("sampled with the trained weights:", tuple(out.shape), "finite:", bool(torch.isfinite(out).all()))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
