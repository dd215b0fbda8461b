"""TfReader factory (reference: nabu/processing/tfreaders/tfreader_factory.py:4-36)."""


def factory(datatype):
    '''Args: datatype: the `type` field of a database.conf section'''
    if datatype == 'audio_feature':
        from nabu_amd.processing.tfreaders import audio_feature_reader
        return audio_feature_reader.AudioFeatureReader
    elif datatype == 'string':
        from nabu_amd.processing.tfreaders import string_reader
        return string_reader.StringReader
    elif datatype == 'string_eos':
        from nabu_amd.processing.tfreaders import string_reader_eos
        return string_reader_eos.StringReaderEOS
    elif datatype in ('binary', 'alignment'):
        raise Exception('%s readers belong to recipes outside the hot path' % datatype)
    else:
        raise Exception('unknown data type: %s' % datatype)
