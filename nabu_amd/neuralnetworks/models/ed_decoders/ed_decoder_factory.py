"""EDDecoder classes by recipe name (the role of models/ed_decoders/ed_decoder_factory.py:5-24)."""
from nabu_amd.tools.registry import Registry

_PKG = 'nabu_amd.neuralnetworks.models.ed_decoders.'
factory = Registry('decoder', {
    'speller': _PKG + 'speller:Speller',
    'dnn_decoder': _PKG + 'dnn_decoder:DNNDecoder',
}, outside=('hotstart_decoder',))
