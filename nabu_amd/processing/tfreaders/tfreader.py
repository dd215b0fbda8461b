"""General TFRecord reader (reference: nabu/processing/tfreaders/tfreader.py:8-78).  The reference
reader is a graph op fed by a filename queue; here it is a plain callable: filename -> (data, length)."""
import os
from abc import ABCMeta, abstractmethod

import numpy as np

from nabu_amd.processing import tfrecord


class TfReader(object, metaclass=ABCMeta):
    '''class for reading tfrecord files and processing them'''

    def __init__(self, datadirs):
        '''Args: datadirs: the directories where the metadata was stored, as a list of strings'''
        self.metadata = self._read_metadata(datadirs)

    def __call__(self, filename):
        '''read one utterance; returns (numpy array, sequence length)'''
        records = tfrecord.read_records(filename)
        if len(records) != 1:
            raise Exception('%s: expected one example per file, found %d' % (filename, len(records)))
        return self._process_features(tfrecord.decode_example(records[0]))

    def sequence_length(self, filename):
        '''the sequence length of one utterance; readers that can tell it without decoding the
        record override this (the bucketing input pipeline asks for the length of EVERY utterance
        before the first step)'''
        return self(filename)[1]

    @staticmethod
    def _lengths(datadirs, metadata):
        '''max_length and the summed sequence_length_histogram of the directories
        (common part of the reference readers' _read_metadata)'''
        max_lengths = []
        for datadir in datadirs:
            with open(os.path.join(datadir, 'max_length')) as fid:
                max_lengths.append(int(fid.read()))
        metadata['max_length'] = max(max_lengths)
        metadata['sequence_length_histogram'] = np.zeros([metadata['max_length'] + 1])
        for datadir in datadirs:
            histogram = np.load(os.path.join(datadir, 'sequence_length_histogram.npy'))
            metadata['sequence_length_histogram'][:histogram.shape[0]] += histogram

    @abstractmethod
    def _read_metadata(self, datadirs):
        '''read the metadata written by the processor; returns a dict'''

    @abstractmethod
    def _process_features(self, features):
        '''features (dict name -> list) -> (data, sequence_length)'''
