"""MI355X-native drop-in for the hot path of kssteven418/Q-ASR (a NeMo 1.0 fork).

Only the import paths the quantised-inference entry point needs are provided
(SURVEY.md §8b): nemo.quantization.utils.{quant_modules,quant_utils,quantize_model},
nemo.collections.asr.models.EncDecCTCModel, nemo.collections.asr.metrics.wer.
Everything here is written from scratch; the integer forward runs in hand-written HIP
kernels (qasr.engine), host PyTorch is used for calibration, packing and decoding.
"""
__version__ = '1.0.0+qasr.mi355x'
