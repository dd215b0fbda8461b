"""TfWriter factory (reference: nabu/processing/tfwriters/tfwriter_factory.py)."""


def factory(writer_type):
    '''Args: writer_type: 'array' or 'string' (the types on the hot path's data contract)'''
    if writer_type == 'array':
        from nabu_amd.processing.tfwriters import array_writer
        return array_writer.ArrayWriter
    elif writer_type == 'string':
        from nabu_amd.processing.tfwriters import string_writer
        return string_writer.StringWriter
    elif writer_type in ('binary', 'alignment'):
        raise Exception('%s writers belong to recipes outside the hot path' % writer_type)
    else:
        raise Exception('unknown writer type: %s' % writer_type)
