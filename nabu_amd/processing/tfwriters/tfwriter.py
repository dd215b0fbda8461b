"""General TFRecord writer (reference: nabu/processing/tfwriters/tfwriter.py:9-55): one file per
utterance under <datadir>/data/file<N>, and a line `name<TAB>filename` in <datadir>/pointers.scp."""
import os
from abc import ABCMeta, abstractmethod

from nabu_amd.processing import tfrecord


class TfWriter(object, metaclass=ABCMeta):
    '''a general class for writing utterances as TFRecord files'''

    def __init__(self, datadir):
        if not os.path.exists(datadir):
            os.makedirs(datadir)
        self.scp_file = os.path.join(datadir, 'pointers.scp')
        self.write_dir = os.path.join(datadir, 'data')
        os.makedirs(self.write_dir)
        self.filenum = 0

    def write(self, data, name):
        example = self._get_example(data)
        filename = os.path.join(self.write_dir, 'file%d' % self.filenum)
        self.filenum += 1
        tfrecord.write_records(filename, [example])
        with open(self.scp_file, 'a') as fid:
            fid.write('%s\t%s\n' % (name, filename))

    @abstractmethod
    def _get_example(self, data):
        '''the serialized tf.train.Example of one utterance'''
