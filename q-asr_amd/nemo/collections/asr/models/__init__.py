from nemo.collections.asr.models.ctc_models import EncDecCTCModel  # noqa: F401
