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

    def _process_features(self, features):
        data = np.frombuffer(features['data'][0], np.float32).reshape([-1, self.metadata['dim']])
        return data, data.shape[0]
