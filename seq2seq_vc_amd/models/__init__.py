"""Model classes resolved by name from YAML, as `seq2seq_vc.models` is (reference models/__init__.py:1-8,
bin/vc_train.py:348-352)."""
from .aas_vc import AASVC  # noqa: F401
from .fastspeech_vc import FastSpeechVC  # noqa: F401
from .transformer_tts import TransformerTTS  # noqa: F401
from .vtn import VTN  # noqa: F401

AR_VC_MODELS = [VTN]
NAR_VC_MODELS = [FastSpeechVC, AASVC]
