"""AudioFeatureReader (reference: nabu/processing/tfreaders/audio_feature_reader.py:9-78)."""
import os

import numpy as np

from nabu_amd.processing.tfreaders import tfreader


class AudioFeatureReader(tfreader.TfReader):
    '''reader for audio features: 'data' = raw float32 [T, dim]'''

    def _read_metadata(self, datadirs):
        metadata = dict()
        self._lengths(datadirs, metadata)
        with open(os.path.join(datadirs[0], 'dim')) as fid:
            metadata['dim'] = int(fid.read())
        for datadir in datadirs:
            with open(os.path.join(datadir, 'dim')) as fid:
                if metadata['dim'] != int(fid.read()):
                    raise Exception('all audio feature reader dimensions must be the same')
        return metadata

    def sequence_length(self, filename):
        '''frames = bytes of the 'data' feature / (4 * dim), read from the record's length prefixes'''
        from nabu_amd.processing import tfrecord
        n = tfrecord.peek_single_bytes_feature(filename, 'data')
        if n is None or n % (4 * self.metadata['dim']):
            return super(AudioFeatureReader, self).sequence_length(filename)
        return n // (4 * self.metadata['dim'])

    def _process_features(self, features):
        data = np.frombuffer(features['data'][0], np.float32).reshape([-1, self.metadata['dim']])
        return data, data.shape[0]
