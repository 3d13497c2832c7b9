"""livelyspeaker_amd -- MI355X-native RAG denoising / diffusion sampling (drop-in for the reference's
``model(x, t, y=...)`` + ``diffusion.p_sample_loop / ddim_sample_loop`` pair). See DESIGN.md."""
from .synth import BEAT, CONFIGS, TED, PathConfig  # noqa: F401


def __getattr__(name):          # lazy: importing the package must not require torch or the .so
    if name in ("RAG",):
        from .rag import RAG
        return RAG
    if name == "ClassifierFreeSampleModel":
        from .cfg_sampler import ClassifierFreeSampleModel
        return ClassifierFreeSampleModel
    if name in ("create_model_and_diffusion", "load_model_wo_clip", "create_gaussian_diffusion"):
        from . import model_util
        return getattr(model_util, name)
    if name in ("GaussianDiffusion", "SpacedDiffusion", "space_timesteps"):
        from . import gaussian_diffusion, respace
        return getattr(respace, name, None) or getattr(gaussian_diffusion, name)
    if name in ("Engine", "EngineError"):
        from . import _lib
        return getattr(_lib, name)
    raise AttributeError(name)
