"""SAG decoder: oracle vs the golden fixture from the imported reference (CPU), HIP vs
golden/oracle and the Decoder_TRANSFORMER drop-in (GPU)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, max_abs
from livelyspeaker_amd import synth


@pytest.fixture(scope="module")
def g6():
    return np.load(os.path.join(GOLDEN, "sag_golden.npz"))


def _inputs(B=6):
    return synth.make_cond(synth.TED, B)["origin_x"], synth.make_text_features(B)


def test_oracle_matches_reference_fixture(g6):
    from oracle import rag_oracle as orc
    x, z = _inputs()
    oracle = orc.SagDecoderOracle(synth.make_sag_state_dict())
    assert max_abs(oracle.decode(x, z, None), g6["G6_sag_all"]) < 2e-5
    out = oracle.decode(x, z, g6["G6_mask_ragged"])
    assert max_abs(out, g6["G6_sag_ragged"]) < 2e-5
    assert float(np.abs(out[1, :, :, 30:]).max()) == 0.0 and float(np.abs(out[4, :, :, 20:]).max()) == 0.0


def test_dropin_state_dict_contract():
    import torch
    from livelyspeaker_amd.motionclip_module import Decoder_TRANSFORMER
    dec = Decoder_TRANSFORMER(latent_dim=512, n_pre_poses=4, use_style=False)
    want = synth.make_sag_state_dict()
    have = {k: tuple(v.shape) for k, v in dec.state_dict().items() if not k.endswith(".pe")}
    assert set(have) == set(want) and all(have[k] == want[k].shape for k in want)
    missing, unexpected = dec.load_state_dict({k: torch.from_numpy(v) for k, v in want.items()}, strict=False)
    assert not unexpected and missing == ["sequence_pos_encoder.pe"]


@pytest.mark.gpu
def test_hip_decoder_vs_golden_and_oracle(g6):
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    sd = synth.make_sag_state_dict()
    eng = _lib.SagEngine()
    try:
        eng.load_state_dict(sd)
        x, z = _inputs()
        d_all = max_abs(eng.decode(x, z), g6["G6_sag_all"])
        out = eng.decode(x, z, g6["G6_mask_ragged"])
        d_rag = max_abs(out, g6["G6_sag_ragged"])
        print(f"SAG decoder vs reference: all {d_all:.3e} ragged {d_rag:.3e}")
        assert d_all < 1e-4 and d_rag < 1e-4
        assert float(np.abs(out[4, :, :, 20:]).max()) == 0.0
        # caller batch (512): spot-check against the oracle, and batch-composition independence
        B = 512
        xb, zb = synth.make_cond(synth.TED, B)["origin_x"], synth.make_text_features(B)
        big = eng.decode(xb, zb)
        pick = [0, 255, 511]
        assert max_abs(big[pick], orc.SagDecoderOracle(sd).decode(xb[pick], zb[pick])) < 1e-4
        assert np.array_equal(eng.decode(xb[pick], zb[pick]), big[pick])
    finally:
        eng.close()


# ---- BEAT twin (scripts_beat/model/motionclip_module.py:98-183: 47 joints x 6 features, mapping = Linear(283, 512)) ----
@pytest.fixture(scope="module")
def g6_beat():
    return np.load(os.path.join(GOLDEN, "sag_beat_golden.npz"))


def _inputs_beat(B=6):
    return synth.make_cond(synth.BEAT, B)["origin_x"], synth.make_text_features(B)


def test_oracle_matches_reference_fixture_beat(g6_beat):
    from oracle import rag_oracle as orc
    x, z = _inputs_beat()
    oracle = orc.SagDecoderOracle(synth.make_sag_state_dict(synth.BEAT), njoints=synth.BEAT.njoints, nfeats=synth.BEAT.nfeats)
    assert max_abs(oracle.decode(x, z, None), g6_beat["G6_sag_all"]) < 2e-5
    assert max_abs(oracle.decode(x, z, g6_beat["G6_mask_ragged"]), g6_beat["G6_sag_ragged"]) < 2e-5


@pytest.mark.gpu
def test_hip_decoder_vs_golden_beat(g6_beat):
    import torch
    from livelyspeaker_amd import _lib
    from livelyspeaker_amd.motionclip_module import Decoder_TRANSFORMER
    cfg = synth.BEAT
    sd = synth.make_sag_state_dict(cfg)
    x, z = _inputs_beat()
    eng = _lib.SagEngine(cfg.njoints, cfg.nfeats)
    try:
        eng.load_state_dict(sd)
        d_all = max_abs(eng.decode(x, z), g6_beat["G6_sag_all"])
        out = eng.decode(x, z, g6_beat["G6_mask_ragged"])
        d_rag = max_abs(out, g6_beat["G6_sag_ragged"])
        print(f"BEAT SAG decoder vs reference: all {d_all:.3e} ragged {d_rag:.3e}")
        assert d_all < 1e-4 and d_rag < 1e-4
        assert float(np.abs(out[4, :, :, 20:]).max()) == 0.0
    finally:
        eng.close()
    # the drop-in module with the variant's constructor arguments (mapping must come out as Linear(283, 512))
    dec = Decoder_TRANSFORMER(njoints=cfg.njoints, nfeats=cfg.nfeats, latent_dim=512, n_pre_poses=4, use_style=False)
    assert tuple(dec.mapping.weight.shape) == (512, 283)
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    dec = dec.to("cuda:0").eval()
    batch = {"x": torch.from_numpy(x).to("cuda:0"), "mask": torch.from_numpy(g6_beat["G6_mask_ragged"]).to("cuda:0"),
             "z": torch.from_numpy(z).to("cuda:0")}
    got = dec(batch)["output"].cpu().numpy()
    assert max_abs(got, g6_beat["G6_sag_ragged"]) < 1e-4


@pytest.mark.gpu
def test_livelyspeaker_pipeline_dropin(g6):
    """test_LivelySpeaker_ted.py:77-113 end to end: SAG.decoder(batch) -> init_image -> ddim100, skip 80 refine."""
    import torch
    from types import SimpleNamespace
    from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
    from livelyspeaker_amd.model_util import create_model_and_diffusion
    from livelyspeaker_amd.motionclip_module import Decoder_TRANSFORMER
    from oracle import rag_oracle as orc
    cfg = synth.TED
    B = 4
    dev = "cuda:0"
    dec = Decoder_TRANSFORMER(latent_dim=512, n_pre_poses=4, use_style=False)
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_sag_state_dict().items()}, strict=False)
    dec.to(dev).eval()
    y_np = synth.make_cond(cfg, B)
    vec = torch.from_numpy(y_np["origin_x"]).to(dev)
    batch = {"x": vec, "mask": torch.ones(B, 34, device=dev).bool(), "z": torch.from_numpy(synth.make_text_features(B)).to(dev)}
    decoded = dec(batch)["output"]
    assert decoded.is_cuda and tuple(decoded.shape) == (B, 9, 3, 34)
    args = SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                           emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu", diffusion_steps=1000,
                           noise_schedule="cosine", sigma_small=True, lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0)
    model, diffusion = create_model_and_diffusion(args, "ddim100")
    sd = synth.make_state_dict(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = ClassifierFreeSampleModel(model)
    model.to(dev)
    model.eval()
    y = {k: torch.from_numpy(v).to(dev) for k, v in y_np.items()}
    torch.manual_seed(7)
    sample = diffusion.ddim_sample_loop(model, (B, 9, 3, 34), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=80,
                                        init_image=decoded, progress=True, dump_steps=None, noise=None, const_noise=False)
    # oracle replay of the same pipeline with the same torch draws
    torch.manual_seed(7)
    x_init = torch.randn(B, 9, 3, 34).numpy()
    later = torch.empty(34, B, 9, 3).permute(1, 2, 3, 0)
    eps, nz = np.empty((20, 2, B, 512), np.float32), np.empty((20, B, 9, 3, 34), np.float32)
    for k in range(20):
        eps[k, 0], eps[k, 1] = torch.randn(B, 1, 512)[:, 0].numpy(), torch.randn(B, 1, 512)[:, 0].numpy()
        nz[k] = (torch.randn(B, 9, 3, 34) if k == 0 else torch.randn_like(later)).contiguous().numpy()
    init = orc.SagDecoderOracle(synth.make_sag_state_dict()).decode(y_np["origin_x"], synth.make_text_features(B))
    want = orc.sample_loop(orc.RagOracle(sd, 9, 3, 1), orc.Schedule(1000, "ddim100"), y_np, x_init, eps, nz, ddim=True,
                           skip_timesteps=80, init_image=init)
    d = max_abs(sample.cpu().numpy(), want)
    print(f"LivelySpeaker pipeline (SAG decode + 20 DDIM refine steps) vs oracle: max|d| = {d:.3e}")
    assert d < 3e-4
