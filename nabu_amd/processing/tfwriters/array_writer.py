"""ArrayWriter (reference: nabu/processing/tfwriters/array_writer.py:8-27): features 'shape' (raw int32
bytes) and 'data' (raw float32 bytes of the flattened array)."""
import numpy as np

from nabu_amd.processing import tfrecord
from nabu_amd.processing.tfwriters import tfwriter


class ArrayWriter(tfwriter.TfWriter):
    '''a TfWriter to write numpy arrays'''

    def _get_example(self, data):
        data = np.asarray(data)
        return tfrecord.encode_example({
            'shape': np.array(data.shape, np.int32).tobytes(),      # reference: int32 bytes of the shape
            'data': data.reshape([-1]).astype(np.float32).tobytes()})
