"""Model classes resolved by name from YAML, as `seq2seq_vc.models` is (reference models/__init__.py:6-8,
bin/vc_train.py:348-352)."""
from .vtn import VTN  # noqa: F401
from .transformer_tts import TransformerTTS  # noqa: F401

try:  # AAS-VC needs the Conformer / alignment kernels
    from .aas_vc import AASVC  # noqa: F401
except ImportError:  # pragma: no cover
    AASVC = None

AR_VC_MODELS = ["VTN"]
NAR_VC_MODELS = ["AASVC"]
