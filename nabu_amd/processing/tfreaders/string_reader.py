"""StringReader (reference: nabu/processing/tfreaders/string_reader.py:9-105): the space separated
symbols of 'data' are mapped to their index in the alphabet file (0-based; the nonesymbol sits at -1)."""
import os

import numpy as np

from nabu_amd.processing.tfreaders import tfreader


class StringReader(tfreader.TfReader):
    '''a reader for reading and encoding text data'''

    def _read_metadata(self, datadirs):
        metadata = dict()
        self._lengths(datadirs, metadata)
        with open(os.path.join(datadirs[0], 'nonesymbol')) as fid:
            nonesymbol = fid.read()
        with open(os.path.join(datadirs[0], 'alphabet')) as fid:
            alphabet = fid.read().split()
        for datadir in datadirs:
            with open(os.path.join(datadir, 'alphabet')) as fid:
                if alphabet != fid.read().split():
                    raise Exception('all string reader alphabets must be the same')
        metadata['alphabet'] = [nonesymbol] + alphabet
        metadata['index'] = {s: i for i, s in reversed(list(enumerate(metadata['alphabet'])))}
        return metadata

    def _encode(self, features):
        symbols = features['data'][0].decode().split(' ')
        symbols = [s for s in symbols if s != '']
        try:
            return np.array([self.metadata['index'][s] - 1 for s in symbols], np.int32)
        except KeyError:
            raise Exception('not all string elements found in alphabet: %s' % features['data'][0])

    def _process_features(self, features):
        data = self._encode(features)
        return data, data.shape[0]
