"""StringWriter (reference: nabu/processing/tfwriters/string_writer.py:7-27): features 'length' (int64)
and 'data' (the space separated symbol string)."""
from nabu_amd.processing import tfrecord
from nabu_amd.processing.tfwriters import tfwriter


class StringWriter(tfwriter.TfWriter):
    '''a TfWriter to write strings'''

    def _get_example(self, data):
        raw = data.encode() if isinstance(data, str) else bytes(data)
        return tfrecord.encode_example({'length': [len(raw)], 'data': raw})
