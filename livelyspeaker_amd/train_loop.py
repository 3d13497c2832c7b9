"""``TrainLoop`` drop-in (``scripts/train_utils/train_loop.py:22-233``): same constructor arguments, ``run_loop`` /
``run_step`` / ``forward_backward`` / ``_anneal_lr`` / ``save`` and the same checkpoint file names, with the whole
optimisation step (q_sample, forward, Huber + velocity + KLD losses, backward, AdamW) executed by the gfx950 engine
through the C-ABI (``ls_train_*``).  There is no autograd graph and no CPU path.

Differences a maintainer should know about (all forced by the fused step):
  * ``losses`` come back as Python floats, not tensors; ``self.last_losses`` holds the last step's terms.
  * random draws are made on torch's CPU generator in the reference's order -- ``schedule_sampler.sample`` (numpy),
    ``randn_like(x_start)`` (gaussian_diffusion.py:1281), the ``mask_cond`` bernoulli (RAG.py:88), ``randn_like(z_mu)``
    (RAG.py:12) -- so seeding reproduces the reference's CPU-path stream.
  * data parallelism (absent from the reference: ``use_ddp=False``, train_loop.py:79): when ``torch.distributed`` is
    initialised the flat gradient array is all-reduced (RCCL over xGMI, one 16 MB bucket) and averaged between backward
    and AdamW; every rank holds the full parameters and applies the same update.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _lib
from .resample import create_named_schedule_sampler


def batch_from_reference_tuple(batch, speaker_model, device):
    """The reference's data-loader tuple -> (motion, cond), as train_loop.py:113-131 does inline (TED)."""
    _, _, _, text_padded, _, vec_seq, audio, _, aux_info = batch
    motion = vec_seq.reshape(vec_seq.shape[0], vec_seq.shape[1], 9, 3).permute(0, 2, 3, 1)
    n_frames = vec_seq.shape[1]
    vid_indices = torch.LongTensor([speaker_model.word2index[v] for v in aux_info['vid']]).to(device)
    cond = {'y': {'mask': torch.ones(vec_seq.shape[0], n_frames).to(device).bool(),
                  'lengths': torch.ones(vec_seq.shape[0], n_frames).to(device) * n_frames, 'text': aux_info["sentence"],
                  'audio_input': audio.to(device).float(), 'vid_indices': vid_indices, 'text_padded': text_padded.to(device),
                  'origin_x': motion.clone().to(device)}}
    return motion.to(device), cond


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """Average a flat gradient array over the data-parallel group in place (one bucket; RCCL on GPUs, gloo in CPU tests)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        from .shard import all_reduce_
        all_reduce_(flat, dist.ReduceOp.SUM)      # device memory under RCCL; a host copy under gloo
        flat.div_(dist.get_world_size())
    return flat


class TrainLoop:
    def __init__(self, args, train_platform, model, diffusion, data):
        self.args = args
        self.dataset = getattr(args, "dataset", None)
        self.train_platform = train_platform
        self.model = model
        self.diffusion = diffusion
        self.cond_mode = getattr(model, "cond_mode", "no_cond")
        self.data = data
        self.batch_size = args.batch_size
        self.microbatch = args.batch_size
        self.lr = args.lr
        self.log_interval = getattr(args, "log_interval", 1000)
        self.save_interval = getattr(args, "save_interval", 50000)
        self.resume_checkpoint = getattr(args, "resume_checkpoint", "")
        self.weight_decay = getattr(args, "weight_decay", 0.0)
        self.lr_anneal_steps = getattr(args, "lr_anneal_steps", 0)
        self.step = 0
        self.resume_step = 0
        self.global_batch = self.batch_size * (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1)
        self.num_epochs = getattr(args, "epochs", 1)
        self.save_dir = getattr(args, "save_dir", ".")
        self.overwrite = getattr(args, "overwrite", False)
        self.schedule_sampler_type = 'uniform'
        self.schedule_sampler = create_named_schedule_sampler(self.schedule_sampler_type, diffusion)
        self.use_ddp = False
        self.speaker_model = getattr(getattr(data, "dataset", None), "speaker_model", None)
        self.cur_lr = self.lr
        self.last_losses = {}
        #: where the step's random draws are made: "cpu" = torch's CPU generator in the reference's order (reproduces the
        #: reference's CPU-path stream; ~3 ms of host RNG + H2D per step at B=512), "cuda" = torch's generator of the
        #: training device, same order, no host work (what the reference effectively does when it trains on a GPU).
        self.noise_device = "cpu"

        di = model._device_index()
        self.device = torch.device("cuda", di)
        self.trainer = _lib.Trainer(model.njoints, model.nfeats, model.n_prefix_tokens, model.AUDIO_LEN[model.n_prefix_tokens],
                                    n_emotions=model.n_emotions, nframes=model.nframes, n_pre_seq=model.n_pre_seq,
                                    layers=model.num_layers, device=di, diffusion_steps=diffusion.num_timesteps,
                                    lambda_vel=float(diffusion.lambda_vel), kld_weight=0.01)
        self.trainer.set_schedule(diffusion)
        self._load_and_sync_parameters()
        if self.resume_step:
            self._load_optimizer_state()

    # ---------------------------------------------------------------- parameters in / out
    def _model_sd(self):
        return {k: v.detach().cpu().numpy() for k, v in self.model.state_dict().items() if not k.endswith(".pe")}

    def _load_and_sync_parameters(self):
        if self.resume_checkpoint:
            self.resume_step = parse_resume_step_from_filename(self.resume_checkpoint)
            self.model.load_state_dict(torch.load(self.resume_checkpoint, map_location="cpu"), strict=False)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from . import shard
            self.model.load_state_dict(shard.broadcast_state_dict(self.model.state_dict(), self.device), strict=False)
        self.trainer.load_state_dict(self._model_sd())

    def _load_optimizer_state(self):
        path = os.path.join(os.path.dirname(self.resume_checkpoint), f"opt{self.resume_step:09}.pt")
        lr = None
        if os.path.exists(path):
            st, lr = unpack_optimizer_state(self.model, torch.load(path, map_location="cpu"))
            if st["exp_avg"]:
                self.trainer.load_optimizer_state(st)
        # opt.load_state_dict restores the annealed lr of the last executed step through param_groups (train_loop.py:96-104).
        # Without an optimizer file the reference keeps the AdamW it has just built at args.lr (train_loop.py:60-66): the first
        # resumed step runs at the un-annealed lr and _anneal_lr takes over after it -- same here.
        self.cur_lr = lr if lr is not None else self.lr

    def sync_model(self):
        """Copy the trained master parameters back into ``self.model`` (so sampling / state_dict() see them)."""
        sd = {k: torch.from_numpy(v) for k, v in self.trainer.state_dict().items()}
        self.model.load_state_dict(sd, strict=False)
        return self.model

    # ---------------------------------------------------------------- the loop
    def run_loop(self):
        for epoch in range(self.num_epochs):
            for batch in self.data:
                if isinstance(batch, (tuple, list)) and len(batch) == 2 and isinstance(batch[1], dict):
                    motion, cond = batch
                else:
                    motion, cond = batch_from_reference_tuple(batch, self.speaker_model, self.device)
                if not (not self.lr_anneal_steps or self.step + self.resume_step < self.lr_anneal_steps):
                    break
                self.run_step(motion, cond)
                if self.step % self.log_interval == 0 and "loss" in self.last_losses:
                    print('step[{}]: loss[{:0.5f}]'.format(self.step + self.resume_step, self.last_losses["loss"]))
                    if self.train_platform is not None:
                        for k, v in self.last_losses.items():
                            self.train_platform.report_scalar(name=k, value=v, iteration=self.step, group_name='Loss')
                self.step += 1
            if epoch % 100 == 0 and epoch > 600:
                self.save()
        self.sync_model()

    def run_step(self, batch, cond):
        self.forward_backward(batch, cond)
        self.trainer.adamw(lr=self.cur_lr, weight_decay=self.weight_decay)       # mp_trainer.optimize(self.opt)
        self._anneal_lr()

    def forward_backward(self, batch, cond):
        assert self.microbatch == self.batch_size
        y = cond['y']
        B = batch.shape[0]
        t, weights = self.schedule_sampler.sample(B, "cpu")
        nd = self.device if self.noise_device == "cuda" else torch.device("cpu")
        noise = torch.randn(tuple(batch.shape), device=nd)                       # th.randn_like(x_start)
        p = float(getattr(self.model, "cond_mask_prob", 0.0))
        drop = torch.bernoulli(torch.ones(B, device=nd) * p) if p > 0. else torch.zeros(B, device=nd)   # mask_cond (training mode)
        eps = torch.randn(B, 1, 512, device=nd)                                  # reparameterize
        yy = {k: y[k] for k in ('audio_input', 'origin_x', 'vid_indices') + (('emo',) if self.model.n_prefix_tokens == 2 else ())}
        yy['origin_x'][..., self.model.n_pre_seq:] = 0                           # RAG.py:110, in place like the reference
        dev = self.device
        terms = self.trainer.forward_backward(batch.to(dev), t, noise.to(dev), {k: v.to(dev) for k, v in yy.items()}, drop.to(dev),
                                              eps.reshape(B, 512).to(dev))
        allreduce_mean_(self.trainer.grad)
        self.last_losses = terms
        return terms

    def _anneal_lr(self):
        if not self.lr_anneal_steps:
            return
        frac_done = (self.step + self.resume_step) / self.lr_anneal_steps
        self.cur_lr = self.lr * (1 - frac_done)

    def ckpt_file_name(self):
        return f"model{(self.step + self.resume_step):09d}.pt"

    def save(self):
        """model%09d.pt / opt%09d.pt in the REFERENCE's formats (train_loop.py:203-227), so either side can resume the other's run:
        the model file is ``master_params_to_state_dict`` (every state-dict entry incl. the ``*.pe`` buffers, trained values under
        the parameter keys); the optimizer file is ``torch.optim.AdamW.state_dict()`` (state[i] in parameter order + param_groups
        with the current, possibly annealed, lr)."""
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
        os.makedirs(self.save_dir, exist_ok=True)
        torch.save(pack_model_checkpoint(self.model, self.trainer.state_dict()), os.path.join(self.save_dir, self.ckpt_file_name()))
        torch.save(pack_optimizer_state(self.model, self.trainer.optimizer_state(), self.cur_lr, self.weight_decay),
                   os.path.join(self.save_dir, f"opt{(self.step + self.resume_step):09d}.pt"))


def _param_keys(model):
    """Parameter keys in optimizer order = ``model.parameters()`` order without CLIP (train_loop.py:57-59, fp16_util.py:140-150)."""
    return [k for k, _ in model.named_parameters() if not k.startswith("clip_model.")]


def pack_model_checkpoint(model, trained: dict) -> dict:
    """What ``MixedPrecisionTrainer.master_params_to_state_dict`` + ``save_checkpoint`` write (fp16_util.py:189-196,
    train_loop.py:204-216): ``model.state_dict()`` with the trained values substituted, CLIP weights dropped."""
    sd = {}
    for k, v in model.state_dict().items():
        if k.startswith("clip_model."):
            continue
        sd[k] = (torch.as_tensor(trained[k]).reshape(v.shape).to(v.dtype) if k in trained else v.detach().cpu()).clone()
    return sd


def pack_optimizer_state(model, opt: dict, lr: float, weight_decay: float) -> dict:
    """``torch.optim.AdamW.state_dict()`` layout from the engine's {'step', 'exp_avg', 'exp_avg_sq'} (keys = state-dict names)."""
    keys = _param_keys(model)
    shapes = dict(model.named_parameters())
    state = {}
    if opt["step"] > 0:
        for i, k in enumerate(keys):
            state[i] = {"step": torch.tensor(float(opt["step"])),
                        "exp_avg": torch.as_tensor(opt["exp_avg"][k]).reshape(shapes[k].shape).clone(),
                        "exp_avg_sq": torch.as_tensor(opt["exp_avg_sq"][k]).reshape(shapes[k].shape).clone()}
    group = {"lr": float(lr), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": float(weight_decay), "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(keys)))}
    return {"state": state, "param_groups": [group]}


def unpack_optimizer_state(model, st: dict):
    """Either layout -> ({'step', 'exp_avg', 'exp_avg_sq'}, lr or None): torch's AdamW state_dict (what the reference writes) or
    this package's round-1 files ({'step', 'exp_avg': {key: tensor}, 'exp_avg_sq': {...}})."""
    if "param_groups" not in st:
        return {"step": int(st["step"]), "exp_avg": dict(st["exp_avg"]), "exp_avg_sq": dict(st["exp_avg_sq"])}, None
    keys = _param_keys(model)
    group = st["param_groups"][0]
    order = list(group["params"])
    if len(order) != len(keys):
        raise ValueError(f"optimizer file holds {len(order)} parameters, the model has {len(keys)}")
    out = {"step": 0, "exp_avg": {}, "exp_avg_sq": {}}
    for pos, pid in enumerate(order):
        ent = st["state"].get(pid)
        if ent is None:
            continue
        out["step"] = int(float(ent["step"]))
        out["exp_avg"][keys[pos]] = ent["exp_avg"]
        out["exp_avg_sq"][keys[pos]] = ent["exp_avg_sq"]
    return out, float(group["lr"])


def parse_resume_step_from_filename(filename):
    """path/to/modelNNNNNN.pt -> NNNNNN (train_loop.py:230-243)."""
    split = filename.split("model")
    if len(split) < 2:
        return 0
    try:
        return int(split[-1].split(".")[0])
    except ValueError:
        return 0
