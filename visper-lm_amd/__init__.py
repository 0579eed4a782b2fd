"""visper-lm_amd: MI355X-native (gfx950) implementation of the VisPer-LM pre-training step —
hand-written HIP kernels behind a C ABI (csrc/ -> libvisper_hip.so, include/visper_hip.h) and a
host-side mirror of the reference's `ola_vlm.model` API that drives them."""
__version__ = "0.1.0"
