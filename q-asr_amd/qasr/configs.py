"""Model configs as plain dicts in the shape of the reference's YAML (`model:` section of
examples/asr/conf/quartznet_15x5.yaml / jasper_10x5dr.yaml), generated from qasr.topology."""
from .topology import MODELS, ModelCfg, BlockCfg


def model_config(name_or_cfg, dropout=0.0):
    cfg = MODELS[name_or_cfg]() if isinstance(name_or_cfg, str) else name_or_cfg
    jasper = []
    for b in cfg.blocks:
        d = dict(filters=b.filters, repeat=b.repeat, kernel=[b.kernel], stride=[b.stride], dilation=[b.dilation],
                 dropout=dropout, residual=b.residual)
        if b.separable:
            d['separable'] = True
        if b.residual_dense:
            d['residual_dense'] = True
        jasper.append(d)
    return dict(
        sample_rate=16000, labels=list(cfg.vocabulary),
        preprocessor=dict(_target_='nemo.collections.asr.modules.AudioToMelSpectrogramPreprocessor',
                          normalize='per_feature', window_size=0.02, sample_rate=16000, window_stride=0.01,
                          window='hann', features=cfg.feat_in, n_fft=512, frame_splicing=1, dither=1e-5,
                          stft_conv=False),
        encoder=dict(_target_='nemo.collections.asr.modules.ConvASREncoder', feat_in=cfg.feat_in, activation='relu',
                     conv_mask=True, jasper=jasper),
        decoder=dict(_target_='nemo.collections.asr.modules.ConvASRDecoder', feat_in=cfg.blocks[-1].filters,
                     num_classes=cfg.num_classes, vocabulary=list(cfg.vocabulary)),
        name=cfg.name)


def topology_from_config(model_cfg) -> ModelCfg:
    enc = model_cfg['encoder']
    blocks = []
    for l in enc['jasper']:
        one = lambda v: v[0] if isinstance(v, (list, tuple)) else v
        blocks.append(BlockCfg(filters=l['filters'], kernel=one(l['kernel']), repeat=l['repeat'], stride=one(l['stride']),
                               dilation=one(l['dilation']), residual=l['residual'],
                               separable=l.get('separable', False), residual_dense=l.get('residual_dense', False)))
    dec = model_cfg['decoder']
    return ModelCfg(model_cfg.get('name', 'custom'), enc['feat_in'], blocks, dec['num_classes'],
                    list(dec.get('vocabulary') or model_cfg.get('labels')))
