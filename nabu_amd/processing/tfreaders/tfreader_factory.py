"""Readers by the `type` field of a database.conf section (the role of
nabu/processing/tfreaders/tfreader_factory.py:4-36)."""
from nabu_amd.tools.registry import Registry

_PKG = 'nabu_amd.processing.tfreaders.'
factory = Registry('data', {
    'audio_feature': _PKG + 'audio_feature_reader:AudioFeatureReader',
    'string': _PKG + 'string_reader:StringReader',
    'string_eos': _PKG + 'string_reader_eos:StringReaderEOS',
}, outside=('binary', 'alignment'), undefined='unknown %s type: %s')
