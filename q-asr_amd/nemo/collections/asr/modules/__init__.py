from nemo.collections.asr.modules.audio_preprocessing import AudioToMelSpectrogramPreprocessor  # noqa: F401
from nemo.collections.asr.modules.conv_asr import ConvASRDecoder, ConvASREncoder  # noqa: F401
