"""Factories mirroring ``scripts/mdm_utils/model_util.py`` (TED) and ``scripts_beat/mdm_utils/model_util.py``:
``create_model_and_diffusion`` (:13-16), ``get_model_args`` (:20-37), ``create_gaussian_diffusion`` (:40-74),
``load_model_wo_clip`` (:5-10)."""
from __future__ import annotations

from . import gaussian_diffusion as gd
from .rag import RAG
from .respace import SpacedDiffusion, space_timesteps


def load_model_wo_clip(model, state_dict):
    missing_keys, unexpected_keys = model.load_state_dict(state_dict, strict=False)
    print("missing_keys", missing_keys)
    print("unexpected_keys", unexpected_keys)
    assert len(unexpected_keys) == 0
    assert all([k.startswith('clip_model.') or k.endswith('.pe') for k in missing_keys])


def get_model_args(args, dataset="ted"):
    """dataset: "ted" | "beat" (the reference's two factories) | "beat150" (synthetic long-sequence BEAT variant, perf-only)."""
    beat = dataset in ("beat", "beat150")
    extra = {'nframes': 150, 'audio_len': 160745} if dataset == "beat150" else {}
    return {**extra, 'modeltype': '', 'njoints': args.njoints if beat else 9, 'nfeats': 6 if beat else 3, 'num_actions': 1370,
            'translation': True, 'pose_rep': 'rot6d', 'glob': True, 'glob_rot': True,
            'latent_dim': args.latent_dim, 'ff_size': 1024 if beat else args.ff_size, 'num_layers': args.layers,
            'num_heads': 4, 'dropout': 0.1, 'activation': "gelu", 'data_rep': 'vec_dir', 'cond_mode': args.mdm_condm,
            'cond_mask_prob': args.cond_mask_prob, 'action_emb': 'tensor', 'arch': args.arch,
            'emb_trans_dec': args.emb_trans_dec, 'clip_version': 'ViT-B/32', 'dataset': args.dataset,
            'lang_model': args.lang_model, 'mlpact': 'silu' if beat else args.mlpact,
            'n_prefix_tokens': 2 if beat else 1}


def create_gaussian_diffusion(args, timestep_respacing=''):
    steps = args.diffusion_steps
    betas = gd.get_named_beta_schedule(args.noise_schedule, steps, 1.)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return SpacedDiffusion(
        use_timesteps=sorted(space_timesteps(steps, timestep_respacing)),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X,           # "we always predict x_start" (model_util.py:42)
        model_var_type=gd.ModelVarType.FIXED_LARGE if not args.sigma_small else gd.ModelVarType.FIXED_SMALL,
        loss_type=gd.LossType.HUBER,
        rescale_timesteps=False,
        lambda_vel=args.lambda_vel, lambda_rcxyz=args.lambda_rcxyz, lambda_fc=args.lambda_fc)


def create_model_and_diffusion(args, timestep_respacing='', dataset="ted"):
    model = RAG(**get_model_args(args, dataset))
    diffusion = create_gaussian_diffusion(args, timestep_respacing)
    return model, diffusion
