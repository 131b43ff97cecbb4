"""MI355X-native Stable Diffusion hot path: host-side mirror of
``python_coreml_stable_diffusion`` (the reference package) over ``libsdmi355.so``."""
from ._lib import ATTENTION_IMPLEMENTATIONS, LIB_PATH, LibraryNotBuilt  # noqa: F401
from .hip_model import HipModel, HipVaeDecoder, HipVaeEncoder, Weights, normalize_unet_config, UNET_CONFIGS, VAE_CONFIGS  # noqa: F401
from .text_encoder import HipTextEncoder, load_tokenizer  # noqa: F401,E402
