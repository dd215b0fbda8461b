"""EDEncoder classes by recipe name (the role of models/ed_encoders/ed_encoder_factory.py:4-29)."""
from nabu_amd.tools.registry import Registry

_PKG = 'nabu_amd.neuralnetworks.models.ed_encoders.'
factory = Registry('encoder', {
    'listener': _PKG + 'listener:Listener',
    'dblstm': _PKG + 'dblstm:DBLSTM',
}, outside=('dummy_encoder', 'dnn', 'hotstart_encoder'))
