"""Writers by name (the role of nabu/processing/tfwriters/tfwriter_factory.py)."""
from nabu_amd.tools.registry import Registry

_PKG = 'nabu_amd.processing.tfwriters.'
factory = Registry('writer', {
    'array': _PKG + 'array_writer:ArrayWriter',
    'string': _PKG + 'string_writer:StringWriter',
}, outside=('binary', 'alignment'), undefined='unknown %s type: %s')
