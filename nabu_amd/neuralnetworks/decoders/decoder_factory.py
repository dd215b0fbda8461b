"""Decoder classes by recipe name (the role of nabu/neuralnetworks/decoders/decoder_factory.py:4-37)."""
from nabu_amd.tools.registry import Registry

factory = Registry('decoder', {
    'ctc_decoder': 'nabu_amd.neuralnetworks.decoders.ctc_decoder:CTCDecoder',
    'beam_search_decoder': 'nabu_amd.neuralnetworks.decoders.beam_search_decoder:BeamSearchDecoder',
}, outside=('max_decoder', 'threshold_decoder', 'feature_decoder', 'alignment_decoder', 'random_decoder'),
    undefined='Undefined %s type: %s')
