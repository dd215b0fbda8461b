"""StringReader (reference: nabu/processing/tfreaders/string_reader.py:9-105): the space separated
symbols of 'data' are mapped to their index in the alphabet file (0-based; the nonesymbol sits at -1)."""
import os

import numpy as np

from nabu_amd.processing.tfreaders import tfreader


class StringReader(tfreader.TfReader):
    '''a reader for reading and encoding text data'''

    @staticmethod
    def _symbols_of(datadir):
        with open(os.path.join(datadir, 'alphabet')) as fid:
            return fid.read().split()

    def _read_metadata(self, datadirs):
        metadata = dict()
        self._lengths(datadirs, metadata)
        alphabets = [self._symbols_of(d) for d in datadirs]
        if any(a != alphabets[0] for a in alphabets[1:]):
            raise Exception('string data sets with different alphabets cannot be read together: %s' % (datadirs,))
        with open(os.path.join(datadirs[0], 'nonesymbol')) as fid:
            padding_symbol = fid.read()
        metadata['alphabet'] = [padding_symbol] + alphabets[0]
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
